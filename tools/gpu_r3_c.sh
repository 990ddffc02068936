#!/bin/bash
# lattice kernel: kernel durations (rocprofv3 --kernel-trace) and issue counters of the probe at 121^3 / 201^3
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; mkdir -p $R/gpurun_out/r3c
O=$R/gpurun_out/r3c
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
cd /tmp
for n in ${SIZES:-121 201}; do
  PROBE_OPTIONS="lattice_segments=0" timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace$n -- python3 $R/tools/lattice_probe.py $n 100 > $O/trace$n.log 2>&1 || { tail -5 $O/trace$n.log; exit 1; }
  grep "spmv" $O/trace$n.log
  python3 - <<PY
import csv, glob
for f in glob.glob("$O/trace$n/*/*_kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "lattice" in r["Name"] or "sellp" in r["Name"]:
            print(r["Name"][:60], "calls", r["Calls"], "avg ns", r["AverageNs"], "min", r["MinNs"], "max", r["MaxNs"])
PY
  for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_RD" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
    d=$(echo $c | tr ' ' '_')
    PROBE_OPTIONS="lattice_segments=0" timeout -k 10 400 rocprofv3 --pmc $c --output-format csv -d $O/pmc${n}_$d -- python3 $R/tools/lattice_probe.py $n 20 > $O/pmc${n}_$d.log 2>&1 || { tail -5 $O/pmc${n}_$d.log; echo "pmc $c failed"; continue; }
  done
  python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob("$O/pmc${n}_*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "lattice" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("n = $n, per launch (median):")
for k, v in sorted(agg.items()):
    v = sorted(v); print("  %-28s %14.0f   (launches %d)" % (k, v[len(v)//2], len(v)))
PY
done
