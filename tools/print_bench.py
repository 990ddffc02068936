import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], "value %.4g"%d["value"], "ms/step %.2f"%d["ms_per_step"], "spmv_us", d["roofline"]["avg_launch_us"], [c["solve_ms"] for c in d["config"]["cycles"]])
