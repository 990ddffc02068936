#!/bin/bash
# four-wave SSOR sweep: parity first, then the sweep timings (one-wave sweep beside it) and a short bench
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
export PYTHONUNBUFFERED=1
cd $R
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -k "ssor or SSOR" > gpurun_out/phase_tests.log 2>&1 || { tail -30 gpurun_out/phase_tests.log; exit 1; }
tail -1 gpurun_out/phase_tests.log
[ -n "$ONLY_TESTS" ] && exit 0
timeout -k 10 300 python tools/sgs_probe.py 20 5 1 20 2>&1 | grep -v "^\[gmg\]" | tail -3 || exit 2
GMG_OPTIONS=sgs_disable_phase=1 timeout -k 10 300 python tools/sgs_probe.py 20 5 1 20 2>&1 | grep -v "^\[gmg\]" | tail -2 || exit 3
timeout -k 10 300 python tools/sgs_probe.py 20 5 20 20 2>&1 | grep -v "^\[gmg\]" | tail -2 || exit 4
