#!/bin/bash
# round 3, first GPU call: the -m gpu suite, smoke, the default bench line, and the plainly started multi-rank bench
# lines (ranks sharing the box's one GPU) with the shared direction ring fine-grained and coarse-grained
set -o pipefail
cd ${GRAFT_REPO_ROOT:-$(pwd)}; mkdir -p gpurun_out/r3a
O=gpurun_out/r3a
export PYTHONUNBUFFERED=1
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1 || { tail -40 $O/gpu_tests.log; exit 1; }
tail -2 $O/gpu_tests.log
timeout -k 10 120 python __graft_entry__.py smoke 2>&1 | tail -1 || exit 2
timeout -k 10 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err || { tail -20 $O/bench_default.err; exit 3; }
python tools/print_bench.py $O/bench_default.json
echo "default done"
timeout -k 10 500 python bench.py --gpus 2 --steps 3 --warmup 1 > $O/bench_gpus2_shared.json 2> $O/bench_gpus2.err || { tail -20 $O/bench_gpus2.err; exit 4; }
python tools/print_bench.py $O/bench_gpus2_shared.json
echo "gpus2 done"
timeout -k 10 500 python bench.py --gpus 3 --transport peer --steps 3 --warmup 1 > $O/bench_gpus3_shared.json 2> $O/bench_gpus3.err || { tail -20 $O/bench_gpus3.err; exit 5; }
python tools/print_bench.py $O/bench_gpus3_shared.json
echo "gpus3 done"
GMG_PEER_COARSE=1 timeout -k 10 500 python bench.py --gpus 2 --steps 3 --warmup 1 --no-single-gpu-reference > $O/bench_gpus2_shared_coarse_ring.json 2> $O/bench_gpus2c.err || { tail -20 $O/bench_gpus2c.err; exit 6; }
python tools/print_bench.py $O/bench_gpus2_shared_coarse_ring.json
