#!/bin/bash
# Run on the GPU box (via gpurun): GPU tests, rocprof kernel trace of the small case, and the
# PMC passes for HBM traffic (separate runs, as the MI355X guide prescribes).
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
cd $R
timeout -k 10 300 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1 || { tail -30 gpurun_out/gpu_tests.log; exit 1; }
tail -2 gpurun_out/gpu_tests.log
cd /tmp
B="python3 $R/bench.py --no-cpu-baseline"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_atoms8 -- $B --workload atoms8 --steps 10 --warmup 2 > $R/gpurun_out/prof_atoms8.log 2>&1 || exit 2
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  d=$(echo $c | tr ' ' '_')
  timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/pmc_$d -- $B --steps 1 --warmup 0 --profile-every 0 > $R/gpurun_out/pmc_$d.log 2>&1 || exit 3
done
ls $R/gpurun_out/pmc_FETCH_SIZE/*; du -sh $R/gpurun_out
