#!/bin/bash
# SSOR sweep with the records loaded straight into registers: parity first, then timings against the LDS-staged version
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; mkdir -p $R/gpurun_out/r3h
O=$R/gpurun_out/r3h
export PYTHONUNBUFFERED=1
cd $R
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_layouts.py -x -q -k "ssor or SSOR" > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -1 $O/tests.log
[ -n "$ONLY_TESTS" ] && exit 0
for o in "sgs_reg=1" "sgs_reg=1,sgs_pf_lead_kb=-1" "sgs_reg=1,sgs_pf_lead_kb=512" "sgs_reg=0" "sgs_reg=0,sgs_pf_lead_kb=-1" $MORE_OPTS; do
  echo "== $o"
  GMG_OPTIONS=$o timeout -k 10 300 python tools/sgs_probe.py 20 5 1 20 2>&1 | grep -v "^\[gmg\]" | tail -2 || exit 2
done
timeout -k 10 300 python tools/sgs_probe.py 20 5 20 20 2>&1 | grep -v "^\[gmg\]" | tail -2 || exit 4
GMG_OPTIONS=sgs_phase_profile=1,debug_upload=1 timeout -k 10 300 python tools/sgs_probe.py 20 5 1 1 > $O/reg_profile.txt 2>&1 || exit 5
grep "SGS\|steps" $O/reg_profile.txt | tail -13 | cut -c1-200
