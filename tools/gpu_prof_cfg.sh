#!/bin/bash
# GPU box: rocprofv3 kernel stats of one BASELINE config run (tools/run_config.py <nacl> <cycles> <smoother> <blocks>)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
cd /tmp
tag=${TAG:-cfg}
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$tag -- python3 $R/tools/run_config.py "$@" > $R/gpurun_out/prof_$tag.log 2>&1 || { tail -20 $R/gpurun_out/prof_$tag.log; exit 2; }
f=$(ls $R/gpurun_out/prof_$tag/*/*kernel_stats.csv | head -1)
cp $f $R/gpurun_out/prof_${tag}_kernel_stats.csv
head -25 $f | cut -c1-200
