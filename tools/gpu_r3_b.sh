#!/bin/bash
# lattice kernel: parity tests, then timings at 121^3 and 201^3 against the pattern-run kernel
set -o pipefail
cd ${GRAFT_REPO_ROOT:-$(pwd)}; mkdir -p gpurun_out/r3b
O=gpurun_out/r3b
export PYTHONUNBUFFERED=1
timeout -k 10 600 python -m pytest tests/test_gpu_lattice.py -x -q > $O/lattice_tests.log 2>&1 || { tail -40 $O/lattice_tests.log; exit 1; }
tail -2 $O/lattice_tests.log
PROBE_OPTIONS="disable_lattice=1;lattice_segments=0" timeout -k 10 300 python tools/lattice_probe.py 121 300 2>&1 | grep "spmv" || exit 2
for o in disable_lattice=1 lattice_segments=0 lattice_segments=2 lattice_segments=4; do GMG_OPTIONS=$o,debug_upload=1 timeout -k 10 200 python tools/spmv_probe.py 20 300 2>&1 | grep "spmv\|lattice" || exit 5; done
PROBE_OPTIONS="lattice_segments=0" timeout -k 10 500 python tools/lattice_probe.py 201 100 2>&1 | grep "spmv" || exit 3
timeout -k 10 900 python -m pytest tests/test_gpu_layouts.py tests/test_gpu_parity.py tests/test_gpu_two_ranks.py -x -q > $O/tests2.log 2>&1 || { tail -40 $O/tests2.log; exit 4; }
tail -2 $O/tests2.log
