#!/bin/bash
# GPU box: LDS counters of the SSOR sweep kernel (tools/sgs_probe.py)
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
cd /tmp
rm -rf $R/gpurun_out/pmc_sgs
for c in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS"; do
  d=$(echo $c | tr ' ' '_')
  timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/pmc_sgs/$d -- python3 $R/tools/sgs_probe.py 20 5 ${BLOCKS:-1} 3 > $R/gpurun_out/pmc_sgs_$d.log 2>&1 || { tail -5 $R/gpurun_out/pmc_sgs_$d.log; exit 3; }
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$R/gpurun_out/pmc_sgs/*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "sgs_phase_kernel" in r["Kernel_Name"] or "sgs_wave_kernel<" in r["Kernel_Name"]:
            agg[r["Counter_Name"]][int(r["Grid_Size"]) if "Grid_Size" in r else 0].append(float(r["Counter_Value"]))
for c, v in sorted(agg.items()):
    for g, vals in v.items():
        print(f"{c:28s} grid {g}: max {max(vals):.4g} (n={len(vals)})")
PY
