#!/bin/bash
# parity tests, then the pattern-run kernel against the per-entry gather kernel
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
export PYTHONUNBUFFERED=1
cd $R
timeout -k 10 120 python tools/spmv_probe.py 20 200 && GMG_OPTIONS=disable_sellp=1 timeout -k 10 120 python tools/spmv_probe.py 20 200 || exit 2
GMG_OPTIONS=sell_grid=1536 timeout -k 10 120 python tools/spmv_probe.py 20 200 || exit 2
timeout -k 10 400 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1 || { tail -40 gpurun_out/gpu_tests.log; exit 1; }
tail -1 gpurun_out/gpu_tests.log
timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_v0.log 2>&1 || { tail -5 gpurun_out/bench_v0.log; exit 3; }
python tools/print_bench.py gpurun_out/bench_v0.log
