#!/bin/bash
set -o pipefail
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export PYTHONUNBUFFERED=1
timeout -k 10 120 python tools/spmv_probe.py 20 200 || exit 2
GMG_OPTIONS=sell_grid=2048 timeout -k 10 120 python tools/spmv_probe.py 20 200 || exit 2
GMG_OPTIONS=sellp_cost=3 timeout -k 10 120 python tools/spmv_probe.py 20 200 || exit 2
timeout -k 10 400 python -m pytest tests -m gpu -x -q -k "layout" > gpurun_out/gpu_tests.log 2>&1 || { tail -40 gpurun_out/gpu_tests.log; exit 1; }
tail -1 gpurun_out/gpu_tests.log
