#!/bin/bash
# the driver's launch line for N = 1, with level 0 replicated (auto) and partitioned (always)
set -o pipefail
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
for m in auto always; do
  timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --partition-level0 $m > gpurun_out/bench_torchrun1_$m.json 2> gpurun_out/bench_torchrun1_$m.err || { tail -5 gpurun_out/bench_torchrun1_$m.err; exit 5; }
  python tools/print_bench.py gpurun_out/bench_torchrun1_$m.json
done
