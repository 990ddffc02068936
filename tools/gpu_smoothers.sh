#!/bin/bash
# BASELINE config 5, cycle 4, solve time per smoother (first solve of the cycle: includes first-touch effects)
set -o pipefail
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export PYTHONUNBUFFERED=1
for cfg in "Jacobi 1" "Chebyshev 1" "SSOR 64" "SSOR 20" "SSOR 1"; do
  set -- $cfg
  timeout -k 10 400 python tools/run_config.py 20 5 $1 $2 > gpurun_out/cfg5_$1_$2.log 2>&1 || { tail -5 gpurun_out/cfg5_$1_$2.log; exit 2; }
  echo "$1 blocks=$2: $(grep -A1 '^cycle 4' gpurun_out/cfg5_$1_$2.log | tail -1)"
done
