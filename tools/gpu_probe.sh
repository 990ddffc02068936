#!/bin/bash
set -o pipefail
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export PYTHONUNBUFFERED=1
for c in 1 2 3 4 6; do GMG_SELLP_COST=$c timeout -k 10 120 python tools/spmv_probe.py 20 200 || exit 2; done
GMG_SELL_GRID=4096 timeout -k 10 120 python tools/spmv_probe.py 20 200 || exit 2
GMG_SELL_GRID=1792 timeout -k 10 120 python tools/spmv_probe.py 20 200 || exit 2
timeout -k 10 400 python -m pytest tests -m gpu -x -q -k "layout or parity" > gpurun_out/gpu_tests.log 2>&1 || { tail -40 gpurun_out/gpu_tests.log; exit 1; }
tail -1 gpurun_out/gpu_tests.log
