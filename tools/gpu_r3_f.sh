#!/bin/bash
# SSOR sweep with one dependent wave: parity first (every spin is bounded), then timings against the barrier version
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; mkdir -p $R/gpurun_out/r3f
O=$R/gpurun_out/r3f
export PYTHONUNBUFFERED=1
cd $R
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_layouts.py -x -q -k "ssor or SSOR" > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -1 $O/tests.log
[ -n "$ONLY_TESTS" ] && exit 0
GMG_OPTIONS=sgs_dep=1 timeout -k 10 300 python tools/sgs_probe.py 20 5 1 20 2>&1 | grep -v "^\[gmg\]" | tail -3 || exit 2
timeout -k 10 300 python tools/sgs_probe.py 20 5 1 20 2>&1 | grep -v "^\[gmg\]" | tail -2 || exit 3
GMG_OPTIONS=sgs_dep=1 timeout -k 10 300 python tools/sgs_probe.py 20 5 20 20 2>&1 | grep -v "^\[gmg\]" | tail -2 || exit 4
GMG_OPTIONS=sgs_dep=1,sgs_phase_profile=1 timeout -k 10 300 python tools/sgs_probe.py 20 5 1 1 > $O/dep_profile.txt 2>&1 || exit 5
grep -v "^\[gmg\] level\|^\[gmg\] upload" $O/dep_profile.txt | tail -12
