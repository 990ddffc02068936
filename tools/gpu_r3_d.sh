#!/bin/bash
# SSOR sweep with LDS hand-over: parity first (a hang cannot happen: every spin is bounded), then timings against the barrier version
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; mkdir -p $R/gpurun_out/r3d
O=$R/gpurun_out/r3d
export PYTHONUNBUFFERED=1
cd $R
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_layouts.py -x -q -k "ssor or SSOR or smoother" > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -1 $O/tests.log
[ -n "$ONLY_TESTS" ] && exit 0
timeout -k 10 300 python tools/sgs_probe.py 20 5 1 20 2>&1 | grep -v "^\[gmg\]" | tail -3 || exit 2
GMG_OPTIONS=sgs_disable_chain=1 timeout -k 10 300 python tools/sgs_probe.py 20 5 1 20 2>&1 | grep -v "^\[gmg\]" | tail -2 || exit 3
timeout -k 10 300 python tools/sgs_probe.py 20 5 20 20 2>&1 | grep -v "^\[gmg\]" | tail -2 || exit 4
GMG_OPTIONS=sgs_phase_profile=1 timeout -k 10 300 python tools/sgs_probe.py 20 5 1 1 > $O/phase_cycles.txt 2>&1 || { tail -5 $O/phase_cycles.txt; exit 5; }
grep "four-wave\|steps" $O/phase_cycles.txt | head -24
timeout -k 10 300 python -m pytest tests/test_adaptive_golden.py -m gpu -x -q 2>&1 | tail -2
