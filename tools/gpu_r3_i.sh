#!/bin/bash
# four-wave SSOR sweep with the records stored at the step's own width: parity, then timings against the chunk-wide records
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; mkdir -p $R/gpurun_out/r3i
O=$R/gpurun_out/r3i
export PYTHONUNBUFFERED=1
cd $R
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_layouts.py -x -q -k "ssor or SSOR" > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -1 $O/tests.log
for o in "sgs_reg=0" $MORE_OPTS; do
  echo "== $o"
  GMG_OPTIONS=$o timeout -k 10 300 python tools/sgs_probe.py 20 5 1 20 2>&1 | grep -v "^\[gmg\]" | tail -2 || exit 2
done
timeout -k 10 300 python tools/sgs_probe.py 20 5 20 20 2>&1 | grep -v "^\[gmg\]" | tail -2 || exit 4
GMG_OPTIONS=sgs_phase_profile=1,debug_upload=1 timeout -k 10 300 python tools/sgs_probe.py 20 5 1 1 > $O/profile.txt 2>&1 || exit 5
grep "SGS\|steps" $O/profile.txt | tail -21 | cut -c1-200
