#!/bin/bash
# timing experiments of the SSOR sweep (instrumented variant; modes > 0 give wrong results by design)
cd ${GRAFT_REPO_ROOT:-$(pwd)}; mkdir -p gpurun_out
timeout -k 10 200 python tools/sgs_probe.py 20 5 ${BLOCKS:-1} 20 2>&1 | grep -v "^\[gmg\]" | tail -3
for m in ${MODES:-1 2 3 4 5 6}; do
  GMG_OPTIONS=sgs_profile=$m timeout -k 10 200 python tools/sgs_probe.py 20 5 ${BLOCKS:-1} 2 > gpurun_out/ssor_mode$m.log 2>&1
  grep "rows, 8 ranges\|rows, 1[0-9] ranges" gpurun_out/ssor_mode$m.log | tail -1 | python3 -c "
import sys,re
l=sys.stdin.read()
v=re.findall(r'\[(\d+) (\d+) (\d+) (\d+)\]', l)
print('mode', $m-1, 'sweep cycles fwd', sum(int(x[0]) for x in v[:len(v)//2]), 'bwd', sum(int(x[0]) for x in v[len(v)//2:]), 'wait', sum(int(x[1]) for x in v))
"
done
