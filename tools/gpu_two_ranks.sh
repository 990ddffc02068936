#!/bin/bash
set -o pipefail
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout -k 10 500 python -m pytest tests/test_gpu_two_ranks.py -m gpu -x -q > gpurun_out/two_ranks.log 2>&1
rc=$?
tail -30 gpurun_out/two_ranks.log
ls /dev/shm | head
exit $rc
