#!/bin/bash
# GPU box: the whole -m gpu suite, smoke, then the default bench line
set -o pipefail
cd ${GRAFT_REPO_ROOT:-$(pwd)}; mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1 || { tail -40 gpurun_out/gpu_tests.log; exit 1; }
tail -2 gpurun_out/gpu_tests.log
timeout -k 10 120 python __graft_entry__.py smoke 2>&1 | tail -1 || exit 2
timeout -k 10 600 python bench.py "$@" > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err || { tail -20 gpurun_out/bench_default.err; exit 3; }
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_default.json").read().strip().splitlines()[-1])
print("value %.4g  ms/step %.2f  its %d" % (d["value"], d["ms_per_step"], d["config"]["outer_cg_iterations"]))
for k, v in d["config"]["smoothers"].items():
    print("  ", k, {a: b for a, b in v.items() if a not in ("sweep",)})
r = d["roofline"]
print("roofline frac", r["frac"], "achieved", r["achieved"], "launch us", r["avg_launch_us"], "coarse_iteration", r.get("coarse_iteration"))
print("kernel_time_per_step", r.get("kernel_time_per_step"))
print("cpu", d["cpu_baseline"])
PY
