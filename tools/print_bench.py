"""one-screen summary of a bench.py line: python tools/print_bench.py file.json"""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
c, r = d["config"], d["roofline"] or {}
print(sys.argv[1], "n_gpus", d["n_gpus"], "value %.4g" % d["value"], "ms/step %.2f" % d["ms_per_step"], "its", c["outer_cg_iterations"], "/", c["coarse_cg_iterations_per_step"],
      "commit", c.get("commit"), c.get("source_sha16"))
print("  transport", c.get("transport"), c.get("communicator"), c.get("transport_note"))
for k, v in c["smoothers"].items():
    print("  ", k, {a: (round(b, 1) if isinstance(b, float) else b) for a, b in v.items() if a not in ("sweep", "smoother")})
if c.get("single_gpu_same_smoother"):
    print("  single GPU, same smoother:", {a: b for a, b in c["single_gpu_same_smoother"].items() if a not in ("sweep", "note")})
print("  roofline: frac", r.get("frac"), r.get("frac_basis", "")[:6], "achieved", r.get("achieved"), "layout GB/s", r.get("layout_GBps"), "launch us", r.get("avg_launch_us"), "noop us", r.get("avg_noop_launch_us"),
      "traffic", r.get("traffic"))
print("  coarse_iteration", {k: v for k, v in (r.get("coarse_iteration") or {}).items() if k != "note"})
print("  kernel_time_per_step", {k: v for k, v in (r.get("kernel_time_per_step") or {}).items() if "note" not in k})
print("  dominant", {k: v for k, v in (r.get("dominant_kernel") or {}).items() if "note" not in k})
print("  cpu", d.get("cpu_baseline"), "setup_s", c.get("setup_seconds"), [x["solve_ms"] for x in c["cycles"]])
