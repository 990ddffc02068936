"""bench.py's launch contract: `--gpus N` started plainly must start its own N rank processes --
as CHILDREN, before this process touches torch or HIP -- and relay rank 0's one JSON line.

Here (no GPU) the children can only fail loudly; what is checked is that N of them were started with the torchrun
environment and that their failure comes back as a non-zero exit code.  On the GPU box the same command line is run
for real with the ranks sharing the one GPU (functional: iteration counts, transport, memory types)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _has_gpu():
    import torch

    return torch.cuda.is_available()


def test_source_hash_is_stable_and_cheap():
    a = subprocess.check_output([sys.executable, BENCH, "--source-hash"]).decode().strip()
    b = subprocess.check_output([sys.executable, BENCH, "--source-hash"]).decode().strip()
    assert a == b and len(a) == 16 and int(a, 16) >= 0


def test_plain_start_with_two_gpus_spawns_two_ranks_and_reports_their_failure():
    if _has_gpu():
        pytest.skip("GPU present: covered by test_two_and_three_ranks_share_the_gpu")
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env, capture_output=True, timeout=280)
    err = r.stderr.decode(errors="replace")
    assert r.returncode != 0                      # the ranks' failure is propagated ...
    assert "--gpus 2 but WORLD_SIZE=1" not in err  # ... and it is not the old refusal to start
    # the ranks ran bench.py's main() and failed loudly (no CPU fallback); the launcher ends the other rank as soon as the
    # first one has failed, so the second message may be missing -- that two ranks were started is in the launcher's report
    assert err.count("no HIP device visible") >= 1, err[-2000:]
    assert "local_rank: 0" in err or "local_rank: 1" in err, err[-2000:]
    assert r.stdout.decode().strip() == ""         # no result line without a measurement


@pytest.mark.gpu
@pytest.mark.parametrize("n_ranks", [2, 3])
def test_two_and_three_ranks_share_the_gpu(n_ranks):
    """`python bench.py --gpus N` exactly as the driver starts it, on a box with ONE GPU: the ranks share it, the peer
    transport carries halo entries, sums and SSOR pieces; the solve must take the iterations the single-GPU layout takes
    with the same N SSOR blocks (rank 0 measures that reference inside the same run)."""
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, BENCH, "--gpus", str(n_ranks), "--workload", "atoms8", "--steps", "2", "--warmup", "1", "--cycles", "3",
                        "--partition-level0", "always"], env=env, capture_output=True, timeout=560)
    assert r.returncode == 0, r.stderr.decode(errors="replace")[-3000:]
    lines = [l for l in r.stdout.decode().splitlines() if l.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    cfg = d["config"]
    assert d["n_gpus"] == n_ranks and cfg["peer_ranks"] == n_ranks and cfg["transport"] == "peer"
    assert cfg["communicator"]["ranks"] == n_ranks and cfg["communicator"]["level0_partitioned"]
    assert cfg["communicator"]["ring_memory"] != "not allocated"
    single = cfg["single_gpu_same_smoother"]
    assert single["ssor_blocks"] == n_ranks
    assert single["outer_cg_iterations"] == d["config"]["outer_cg_iterations"]
    assert abs(single["coarse_cg_iterations"] - cfg["coarse_cg_iterations_per_step"]) <= 2 * cfg["outer_cg_iterations"]  # (sums in another order)
    assert len(d["roofline"]["coarse_iteration"]["us_per_iteration_by_rank"]) == n_ranks
