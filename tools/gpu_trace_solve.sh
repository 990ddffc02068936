#!/bin/bash
# kernel trace of the default bench command (few steps): which launches surround the runtime's copy kernels
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/trace; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp PYTHONUNBUFFERED=1
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $O/prof -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-smoother-table --cycles 5 > $O/log.txt 2>&1 || { tail -5 $O/log.txt; exit 1; }
python3 - <<PY
import csv, glob, collections
ks = []
for f in glob.glob("$O/prof/*/*_kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        ks.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-60:]))
mc = []
for f in glob.glob("$O/prof/*/*_memory_copy_trace.csv"):
    for r in csv.DictReader(open(f)):
        mc.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "MEMCPY " + r.get("Direction", "") + " " + r.get("Bytes", r.get("Size", "?"))))
ev = sorted(ks + mc)
print(len(ks), "kernels", len(mc), "memory copies")
# the last solve: take the last 2500 events and print a histogram of names + a window around copy kernels
tail = ev[-4000:]
c = collections.Counter(n for _, _, n in tail)
for n, k in c.most_common(25): print(k, n)
cp = [i for i, e in enumerate(tail) if "copyBuffer" in e[2] or e[2].startswith("MEMCPY")]
print("copies in tail:", len(cp))
for i in cp[:12]:
    print("---")
    for j in range(max(0, i - 2), min(len(tail), i + 3)):
        s, e, n = tail[j]
        print("   %10.1f us  dur %7.1f us  %s" % ((s - tail[0][0]) / 1e3, (e - s) / 1e3, n))
PY
rm -rf $O/prof
