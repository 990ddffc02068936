#!/bin/bash
# GPU tests, then benches of the small and the headline workload (+ optional env toggles)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
export PYTHONUNBUFFERED=1
cd $R
timeout -k 10 300 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1 || { tail -30 gpurun_out/gpu_tests.log; exit 1; }
tail -1 gpurun_out/gpu_tests.log
for w in atoms8 atoms64000; do
  timeout -k 10 300 python bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_$w.log 2>&1 || { tail -5 gpurun_out/bench_$w.log; exit 2; }
  python - <<PY
import json
d=json.loads(open("gpurun_out/bench_$w.log").read().strip().splitlines()[-1])
r=d["roofline"]
print("$w", "value %.3e"%d["value"], "ms/step %.3f"%d["ms_per_step"], "spmv_us", r["avg_launch_us"], "GB/s", r["achieved"], "upd_us", r["cg_update_kernel"]["avg_launch_us"])
PY
done
