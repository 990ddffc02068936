#!/bin/bash
# GPU box: SSOR parity tests first, then the config-5 solve times per smoother
set -o pipefail
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "ssor or smoother or vcycle" > gpurun_out/ssor_tests.log 2>&1 || { tail -30 gpurun_out/ssor_tests.log; exit 1; }
tail -2 gpurun_out/ssor_tests.log
timeout -k 10 300 python -m pytest tests/test_adaptive_golden.py -m gpu -x -q > gpurun_out/ssor_golden.log 2>&1 || { tail -30 gpurun_out/ssor_golden.log; exit 1; }
tail -2 gpurun_out/ssor_golden.log
CONFIGS="${CONFIGS:-SSOR:1 SSOR:20 SSOR:64 Jacobi:1}" bash tools/gpu_smoothers.sh
