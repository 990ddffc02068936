#!/bin/bash
# the driver's launch line for N = 1 (stdout must be exactly one JSON line), and the plain default run
set -o pipefail
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_torchrun1.json 2> gpurun_out/bench_torchrun1.err || { tail -5 gpurun_out/bench_torchrun1.err; exit 5; }
wc -l gpurun_out/bench_torchrun1.json
python tools/print_bench.py gpurun_out/bench_torchrun1.json
timeout -k 10 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_plain.json 2> gpurun_out/bench_plain.err || { tail -5 gpurun_out/bench_plain.err; exit 6; }
wc -l gpurun_out/bench_plain.json
python tools/print_bench.py gpurun_out/bench_plain.json
