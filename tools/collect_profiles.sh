#!/bin/bash
# After the stages of tools/gpu_refresh_profiles.sh: merges gpurun_out/refresh/pmc_traffic_<workload>.json into
# profiles/pmc_traffic.json (STAGE=pmc) and copies the ${TAG}_* files into profiles/.
R=$(cd $(dirname $0)/.. && pwd)
TAG=${TAG:-r03}
O=$R/gpurun_out/refresh
python3 - <<PY
import json, os
old = json.load(open("$R/profiles/pmc_traffic.json")) if os.path.exists("$R/profiles/pmc_traffic.json") else {}
d = dict(old)
for w in ("atoms64000", "atoms8", "atoms1000", "atoms8000", "stress201"):
    f = "$O/pmc_traffic_%s.json" % w
    if os.path.exists(f):
        d.update(json.load(open(f)))
json.dump(d, open("$R/profiles/pmc_traffic.json", "w"), indent=1)
for w, v in d.items():
    if isinstance(v, dict) and "kernels" in v:
        print(w, v.get("commit"), v.get("source_sha16"), {k: x["traffic_bytes"] for k, x in v["kernels"].items() if "lattice" in k or "sellp" in k})
PY
for f in $O/${TAG}_*; do [ -s $f ] && cp $f $R/profiles/; done
ls $R/profiles | grep "^${TAG}_"
