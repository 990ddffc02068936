#!/bin/bash
# GPU tests, then benches of the small and the headline workload (+ optional env toggles)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
export PYTHONUNBUFFERED=1
cd $R
if [ -z "$SKIP_TESTS" ]; then
timeout -k 10 300 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1 || { tail -30 gpurun_out/gpu_tests.log; exit 1; }
tail -1 gpurun_out/gpu_tests.log
fi
for w in atoms8 atoms64000; do
  timeout -k 10 300 python bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_$w.log 2>&1 || { tail -5 gpurun_out/bench_$w.log; exit 2; }
  python tools/print_bench.py gpurun_out/bench_$w.log
done
