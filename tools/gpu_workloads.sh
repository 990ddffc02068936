#!/bin/bash
# bench line of every workload (5 adaptive cycles each, last solve timed; stress201: 2 cycles)
set -o pipefail
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
for w in atoms8 atoms1000 atoms8000 atoms64000; do
  timeout -k 10 300 python bench.py --workload $w --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/wl_$w.json 2> gpurun_out/wl_$w.err || { tail -3 gpurun_out/wl_$w.err; exit 2; }
  python tools/print_bench.py gpurun_out/wl_$w.json
done
timeout -k 10 400 python bench.py --workload stress201 --cycles 2 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/wl_stress201.json 2> gpurun_out/wl_stress201.err || { tail -3 gpurun_out/wl_stress201.err; exit 2; }
python tools/print_bench.py gpurun_out/wl_stress201.json
