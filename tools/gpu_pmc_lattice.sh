#!/bin/bash
# PMC passes (counters only) on the lattice kernel probe (LATN = lattice size)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmcl
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
cd /tmp
i=0
while read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $line --output-format csv -d $OUT/pass$i -- python3 $R/tools/lattice_probe.py ${LATN:-201} 20 > $OUT/pass$i.log 2>&1 || { echo "pass $i failed"; tail -3 $OUT/pass$i.log; }
done <<LIST
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE
SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_LDS_BANK_CONFLICT
SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_IFETCH SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_BUSY_CU_CYCLES SQ_LEVEL_WAVES
TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum
TA_TA_BUSY_sum TA_BUSY_avr
LIST
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/pass*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "spmv" in k:
            agg[k[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    print(k)
    for c, vals in sorted(v.items()):
        vals = sorted(vals)
        print("   %-40s n=%3d median=%.4g" % (c, len(vals), vals[len(vals)//2]))
PY
