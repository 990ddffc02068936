#!/bin/bash
# PMC passes (one rocprofv3 run each, counters only) on the headline workload.
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/pmc
export TMPDIR=/tmp PYTHONUNBUFFERED=1
cd /tmp
B="python3 $R/bench.py --no-cpu-baseline --steps 1 --warmup 0 --profile-every 0 ${BENCH_ARGS}"
i=0
while read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $line --output-format csv -d $R/gpurun_out/pmc/pass$i -- $B > $R/gpurun_out/pmc/pass$i.log 2>&1 || { echo "pass $i failed"; tail -3 $R/gpurun_out/pmc/pass$i.log; }
done <<LIST
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE
TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum
TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum
TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum
TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_BUSY_sum TCC_TAG_STALL_sum TCC_EA0_RDREQ_LEVEL_sum
TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_REQ_sum
LIST
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$R/gpurun_out/pmc/pass*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "spmv" in k or "cg_update" in k:
            agg[k[:48]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    print(k)
    for c, vals in sorted(v.items()):
        big = sorted(x for x in vals if x >= 0.5 * max(vals)) or [0]
        print("   %-44s n=%4d median=%.4g" % (c, len(vals), big[len(big)//2]))
PY
