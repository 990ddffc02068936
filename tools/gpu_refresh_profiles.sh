#!/bin/bash
# Regenerates what profiles/ holds for the current build (run via gpurun, in stages because one call is limited to 20
# minutes: STAGE=pmc | bench | more; results under gpurun_out/refresh, tools/collect_profiles.sh copies them into profiles/
# under the prefix $TAG).  Run tools/stamp_commit.sh first: the lines then carry the commit they were measured on.
#   pmc_traffic.json                 FETCH_SIZE / WRITE_SIZE passes (tools/gpu_pmc_traffic.sh) of every workload; bench.py looks
#                                    its level-0 kernel up there and uses the bytes when the pass was made on these sources
#   ${TAG}_bench_*.json              the bench lines: default run (incl. cpu_baseline and the smoother table), atoms8, atoms1000,
#                                    atoms8000, stress201, the driver's launch line for N = 1, --gpus 2 / 3 on the one GPU
#   ${TAG}_kernel_stats_*.csv        rocprofv3 --kernel-trace --stats of the default bench command / of the stress201 one
#   ${TAG}_sgs_phase_cycles.txt      cycles per step and per phase of the shipped SSOR sweep
#   ${TAG}_pmc_sgs_sweep.txt         LDS / issue counters of the SSOR sweep kernel
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${TAG:-r03}
STAGE=${STAGE:-bench}
O=$R/gpurun_out/refresh
mkdir -p $O
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
cd $R
case $STAGE in
pmc)
  for w in ${WORKLOADS:-atoms64000 atoms8 atoms1000 atoms8000 stress201}; do
    CYCLES=1 bash tools/gpu_pmc_traffic.sh $w > $O/pmc_traffic_$w.log 2>&1 || { tail -5 $O/pmc_traffic_$w.log; exit 4; }
    cp $R/gpurun_out/pmc_traffic_$w.json $O/
    echo "pmc $w done"
  done
  rm -rf $R/gpurun_out/pmc_traffic
  ;;
bench)
  # (profiles/pmc_traffic.json of the pmc stage must already be in the tree: the lines look their kernel up there)
  timeout -k 10 300 python bench.py --workload atoms8 --steps 20 --warmup 3 --no-cpu-baseline > $O/${TAG}_bench_atoms8.json 2> $O/bench_atoms8.err || { tail -5 $O/bench_atoms8.err; exit 1; }
  echo "atoms8 done"
  timeout -k 10 300 python bench.py --workload atoms1000 --steps 10 --warmup 2 --no-cpu-baseline > $O/${TAG}_bench_atoms1000.json 2> $O/bench_atoms1000.err || { tail -5 $O/bench_atoms1000.err; exit 1; }
  echo "atoms1000 done"
  timeout -k 10 300 python bench.py --workload atoms8000 --steps 10 --warmup 2 --no-cpu-baseline > $O/${TAG}_bench_atoms8000.json 2> $O/bench_atoms8000.err || { tail -5 $O/bench_atoms8000.err; exit 1; }
  echo "atoms8000 done"
  timeout -k 10 600 python bench.py > $O/${TAG}_bench_default.json 2> $O/bench_default.err || { tail -5 $O/bench_default.err; exit 2; }
  echo "default done"
  cd /tmp
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-smoother-table > $O/prof.log 2>&1 || { tail -5 $O/prof.log; exit 3; }
  cp $O/prof/*/*_kernel_stats.csv $O/${TAG}_kernel_stats_default.csv && rm -rf $O/prof
  echo "rocprof default done"
  cd $R
  timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --no-smoother-table > $O/${TAG}_bench_torchrun_n1.json 2> $O/bench_torchrun1.err || { tail -5 $O/bench_torchrun1.err; exit 5; }
  echo "torchrun n1 done"
  ;;
more)
  timeout -k 10 600 python bench.py --workload stress201 --cycles 2 --steps 3 --warmup 1 --no-cpu-baseline --no-smoother-table --smoother Jacobi > $O/${TAG}_bench_stress201.json 2> $O/bench_stress201.err || { tail -5 $O/bench_stress201.err; exit 6; }
  cd /tmp
  timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/bench.py --workload stress201 --cycles 2 --steps 3 --warmup 1 --no-cpu-baseline --no-smoother-table --smoother Jacobi > $O/prof.log 2>&1 || { tail -5 $O/prof.log; exit 7; }
  cp $O/prof/*/*_kernel_stats.csv $O/${TAG}_kernel_stats_stress201.csv && rm -rf $O/prof
  echo "stress201 done"
  cd $R
  # N ranks sharing the one GPU (functional evidence: iteration counts, transport, per-rank coarse iteration -- not speed)
  for n in 2 3; do
    timeout -k 10 500 python bench.py --gpus $n --steps 3 --warmup 1 --no-cpu-baseline --no-smoother-table > $O/${TAG}_bench_gpus${n}_shared_gpu_peer.json 2> $O/bench_gpus$n.err || { tail -5 $O/bench_gpus$n.err; exit 9; }
    echo "gpus $n done"
  done
  { echo "# GMG_OPTIONS=sgs_phase_profile=1 python tools/sgs_probe.py 20 5 1 1 (64 k atoms, cycle 4, level 1; MI355X)"
    echo "# per range: direction, steps | shader cycles per step (instrumented run) | load + write-back cycles | wave 0, cycles per turn (4 phases) by part"
    GMG_OPTIONS=sgs_phase_profile=1 timeout -k 10 300 python tools/sgs_probe.py 20 5 1 1 2>&1 | grep "steps\|ms per application"; } > $O/${TAG}_sgs_phase_cycles.txt || { tail -5 $O/${TAG}_sgs_phase_cycles.txt; exit 10; }
  timeout -k 10 300 python tools/sgs_probe.py 20 5 1 20 2>&1 | grep "ms per application" | sed 's/^/# uninstrumented: /' >> $O/${TAG}_sgs_phase_cycles.txt
  bash tools/gpu_pmc_sgs.sh > $O/${TAG}_pmc_sgs_sweep.txt 2>&1 || { tail -5 $O/${TAG}_pmc_sgs_sweep.txt; exit 8; }
  timeout -k 10 120 python __graft_entry__.py smoke 2>&1 | tail -1
  ;;
esac
for f in $O/${TAG}_bench_*.json; do [ -s $f ] && python tools/print_bench.py $f | head -4; done
exit 0
