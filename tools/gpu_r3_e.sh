#!/bin/bash
# partitioned level 0 on the lattice kernel: multi-rank tests, then the plainly started 2- and 3-rank bench lines (ranks share the GPU)
set -o pipefail
cd ${GRAFT_REPO_ROOT:-$(pwd)}; mkdir -p gpurun_out/r3e
O=gpurun_out/r3e
export PYTHONUNBUFFERED=1
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_two_ranks.py tests/test_bench_launch.py tests/test_gpu_layouts.py -x -q > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -2 $O/tests.log
GMG_OPTIONS=debug_upload=1 timeout -k 10 500 python bench.py --gpus 2 --steps 3 --warmup 1 > $O/bench_gpus2_shared.json 2> $O/bench_gpus2.err || { tail -20 $O/bench_gpus2.err; exit 4; }
grep "lattice interior" $O/bench_gpus2.err | tail -2
python tools/print_bench.py $O/bench_gpus2_shared.json
timeout -k 10 500 python bench.py --gpus 3 --transport peer --steps 3 --warmup 1 > $O/bench_gpus3_shared.json 2> $O/bench_gpus3.err || { tail -20 $O/bench_gpus3.err; exit 5; }
python tools/print_bench.py $O/bench_gpus3_shared.json
