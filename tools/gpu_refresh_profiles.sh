#!/bin/bash
# Regenerates what profiles/ holds for the current build (run via gpurun, then copy gpurun_out/refresh/* into profiles/):
#   pmc_traffic.json                          FETCH_SIZE / WRITE_SIZE passes (tools/gpu_pmc_traffic.sh), first: bench.py reads it
#   bench_default.json / bench_atoms8.json    the bench lines (default run incl. cpu_baseline)
#   kernel_stats.csv                          rocprofv3 --kernel-trace --stats of the default bench command
#   bench_torchrun1.json                      the driver's launch line for N = 1
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/refresh
rm -rf $O; mkdir -p $O
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
cd $R
for w in atoms64000 atoms8; do
  bash tools/gpu_pmc_traffic.sh $w > $O/pmc_traffic_$w.log 2>&1 || { tail -5 $O/pmc_traffic_$w.log; exit 4; }
done
rm -rf $R/gpurun_out/pmc_traffic
python3 - <<PY
import json
d = json.load(open("$R/profiles/pmc_traffic.json"))
for w in ("atoms64000", "atoms8"):
    d.update(json.load(open("$R/gpurun_out/pmc_traffic_%s.json" % w)))
json.dump(d, open("$R/profiles/pmc_traffic.json", "w"), indent=1)
json.dump(d, open("$O/pmc_traffic.json", "w"), indent=1)
PY
echo "pmc done"
timeout -k 10 300 python bench.py --workload atoms8 --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_atoms8.json 2> $O/bench_atoms8.err || { tail -5 $O/bench_atoms8.err; exit 1; }
echo "atoms8 done"
timeout -k 10 500 python bench.py > $O/bench_default.json 2> $O/bench_default.err || { tail -5 $O/bench_default.err; exit 2; }
echo "default done"
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/prof.log 2>&1 || { tail -5 $O/prof.log; exit 3; }
cp $O/prof/*/*_kernel_stats.csv $O/kernel_stats.csv && rm -rf $O/prof
echo "rocprof done"
cd $R
timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_torchrun1.json 2> $O/bench_torchrun1.err || { tail -5 $O/bench_torchrun1.err; exit 5; }
python tools/print_bench.py $O/bench_torchrun1.json
python tools/print_bench.py $O/bench_atoms8.json
python tools/print_bench.py $O/bench_default.json
timeout -k 10 120 python __graft_entry__.py smoke 2>&1 | tail -2
