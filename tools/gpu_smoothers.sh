#!/bin/bash
# BASELINE config 5, cycle 4, solve time per smoother (first solve of the cycle: includes first-touch effects)
set -o pipefail
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export PYTHONUNBUFFERED=1
for cfg in ${CONFIGS:-"Jacobi:1" "Chebyshev:1" "SSOR:64" "SSOR:20" "SSOR:1"}; do
  s=${cfg%%:*}; b=${cfg##*:}
  timeout -k 10 400 python tools/run_config.py 20 5 $s $b > gpurun_out/cfg5_${s}_$b.log 2>&1 || { tail -5 gpurun_out/cfg5_${s}_$b.log; exit 2; }
  echo "$s blocks=$b: $(grep -A1 '^cycle 4' gpurun_out/cfg5_${s}_$b.log | tail -1)"
done
