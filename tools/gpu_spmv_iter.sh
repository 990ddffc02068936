#!/bin/bash
# level-0 SpMV tuning loop: the 121^3 probe at a few grid sizes, then the layout / parity tests
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
export PYTHONUNBUFFERED=1
cd $R
for g in ${GRIDS:-0 768 1024 1280 1536 2048}; do
  GMG_OPTIONS=sell_grid=$g timeout -k 10 120 python tools/spmv_probe.py ${NACL:-20} 300 2>&1 | grep "spmv" || exit 2
done
timeout -k 10 600 python -m pytest tests/test_gpu_layouts.py tests/test_gpu_parity.py -x -q > gpurun_out/spmv_iter_tests.log 2>&1 || { tail -40 gpurun_out/spmv_iter_tests.log; exit 1; }
tail -1 gpurun_out/spmv_iter_tests.log
