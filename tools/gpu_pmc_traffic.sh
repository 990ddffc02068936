#!/bin/bash
# HBM traffic of the level-0 kernels from PMC counters: separate rocprofv3 passes for FETCH_SIZE
# and WRITE_SIZE (they do not fit one pass), cycle-0 problem of the given workload.  Writes
# gpurun_out/pmc_traffic_<workload>.json = {workload: {commit, cycles, kernels: {name: {traffic_bytes, ...}}}};
# tools/gpu_refresh_profiles.sh merges it into profiles/pmc_traffic.json (bench.py looks the level-0 kernel up there).
R=${GRAFT_REPO_ROOT:-$(pwd)}
W=${1:-atoms64000}
mkdir -p $R/gpurun_out/pmc_traffic
export TMPDIR=/tmp PYTHONUNBUFFERED=1
cd /tmp
B="python3 $R/bench.py --no-cpu-baseline --no-smoother-table --steps 1 --warmup 0 --profile-every 0 --cycles ${CYCLES:-1} --workload $W"
COMMIT=$(cut -d" " -f1 $R/.bench_commit 2>/dev/null || echo unknown)
SRC=$(python3 $R/bench.py --source-hash)
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/pmc_traffic/$c  # a pass of another workload must not leak into this one
  timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/pmc_traffic/$c -- $B > $R/gpurun_out/pmc_traffic/$c.log 2>&1 || { echo "$c failed"; exit 1; }
done
python3 - <<PY
import csv, glob, collections, json
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob("$R/gpurun_out/pmc_traffic/%s/*/*_counter_collection.csv" % c):
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, v in agg.items():
    if "gmg::" not in k or "FETCH_SIZE" not in v or "WRITE_SIZE" not in v:
        continue
    med = {}
    for c, vals in v.items():
        big = sorted(x for x in vals if x >= 0.5 * max(vals))
        med[c] = big[len(big) // 2]
    name = k.split("gmg::")[1].split("(")[0]
    # counters are in KB; gfx950 FETCH_SIZE counts 64 B per 128-B request of wide streaming reads -> x2
    out[name] = {"fetch_kb": med["FETCH_SIZE"], "write_kb": med["WRITE_SIZE"], "launches": len(v["FETCH_SIZE"]),
                 "traffic_bytes": int((2 * med["FETCH_SIZE"] + med["WRITE_SIZE"]) * 1024)}
json.dump({"$W": {"commit": "$COMMIT", "source_sha16": "$SRC", "cycles": ${CYCLES:-1}, "kernels": out}}, open("$R/gpurun_out/pmc_traffic_$W.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
