#!/bin/bash
# Regenerates what profiles/ holds for the current build (run via gpurun; copies into profiles/ under the prefix $TAG):
#   pmc_traffic.json                 FETCH_SIZE / WRITE_SIZE passes (tools/gpu_pmc_traffic.sh), first: bench.py looks its kernel up there
#   ${TAG}_bench_*.json              the bench lines (default run incl. cpu_baseline and the smoother table; atoms8; stress201)
#   ${TAG}_kernel_stats_*.csv        rocprofv3 --kernel-trace --stats of the default bench command / of the stress201 one
#   ${TAG}_bench_torchrun_n1.json    the driver's launch line for N = 1
#   ${TAG}_pmc_sgs_sweep.txt         LDS / issue counters of the SSOR sweep kernel
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${TAG:-r02}
O=$R/gpurun_out/refresh
rm -rf $O; mkdir -p $O
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
cd $R
for w in atoms64000 atoms8 ${STRESS:+stress201}; do
  bash tools/gpu_pmc_traffic.sh $w > $O/pmc_traffic_$w.log 2>&1 || { tail -5 $O/pmc_traffic_$w.log; exit 4; }
  echo "pmc $w done"
done
rm -rf $R/gpurun_out/pmc_traffic
python3 - <<PY
import json, os
d = {}
for w in ("atoms64000", "atoms8", "stress201"):
    f = "$R/gpurun_out/pmc_traffic_%s.json" % w
    if os.path.exists(f):
        d.update(json.load(open(f)))
old = json.load(open("$R/profiles/pmc_traffic.json"))
for k, v in old.items():
    if k not in d and isinstance(v, dict) and "kernels" in v:
        d[k] = v
json.dump(d, open("$R/profiles/pmc_traffic.json", "w"), indent=1)
json.dump(d, open("$O/pmc_traffic.json", "w"), indent=1)
PY
timeout -k 10 300 python bench.py --workload atoms8 --steps 20 --warmup 3 --no-cpu-baseline > $O/${TAG}_bench_atoms8.json 2> $O/bench_atoms8.err || { tail -5 $O/bench_atoms8.err; exit 1; }
echo "atoms8 done"
timeout -k 10 600 python bench.py > $O/${TAG}_bench_default.json 2> $O/bench_default.err || { tail -5 $O/bench_default.err; exit 2; }
echo "default done"
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-smoother-table > $O/prof.log 2>&1 || { tail -5 $O/prof.log; exit 3; }
cp $O/prof/*/*_kernel_stats.csv $O/${TAG}_kernel_stats_default.csv && rm -rf $O/prof
echo "rocprof default done"
if [ -n "$STRESS" ]; then
  cd $R
  timeout -k 10 600 python bench.py --workload stress201 --cycles 2 --steps 3 --warmup 1 --no-cpu-baseline --no-smoother-table --smoother Jacobi > $O/${TAG}_bench_stress201.json 2> $O/bench_stress201.err || { tail -5 $O/bench_stress201.err; exit 6; }
  cd /tmp
  timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/bench.py --workload stress201 --cycles 2 --steps 3 --warmup 1 --no-cpu-baseline --no-smoother-table --smoother Jacobi > $O/prof.log 2>&1 || { tail -5 $O/prof.log; exit 7; }
  cp $O/prof/*/*_kernel_stats.csv $O/${TAG}_kernel_stats_stress201.csv && rm -rf $O/prof
  echo "stress201 done"
fi
cd $R
timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --no-smoother-table > $O/${TAG}_bench_torchrun_n1.json 2> $O/bench_torchrun1.err || { tail -5 $O/bench_torchrun1.err; exit 5; }
bash tools/gpu_pmc_sgs.sh > $O/${TAG}_pmc_sgs_sweep.txt 2>&1 || { tail -5 $O/${TAG}_pmc_sgs_sweep.txt; exit 8; }
for f in $O/${TAG}_bench_*.json; do python tools/print_bench.py $f; done
timeout -k 10 120 python __graft_entry__.py smoke 2>&1 | tail -1
