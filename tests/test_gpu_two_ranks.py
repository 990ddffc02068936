"""Two ranks on ONE GPU: the rank-parallel layout end to end.

RCCL refuses two ranks on the same device, so the ranks talk through the library's peer-to-peer
transport (GMG_COMM_TRANSPORT=peer, csrc/gmg_comm.hpp: hipIpc-mapped mailboxes, the sender's kernel
stores into the receiver's memory and publishes a sequence number, the receiver's kernel polls it;
the coarse CG pushes the halo entries of d into the neighbours' shared direction vectors) -- same
partition, same halo plans, same distributed coarse CG and V-cycle all-gathers as over RCCL; only
the bytes travel differently.  On this one-GPU box the "peer" stores land in the same device; what
anchors the N-rank results are the reference's logs and the single-process layout.  Checked against the reference's printed numbers (the
reductions are summed in a different order than on one rank, hence 1e-9 instead of 11 digits
for the norms; the iteration counts must not change -- the reference's mpirun=3 / mpirun=7 logs
show the same counts as mpirun=1)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import rel_close
from gpu_util import capi
from test_adaptive_golden import check_cycle

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def run_ranks(n_ranks, golden_dir, tmp_path, monkeypatch, nacl=0, env_extra=None, part="always", blocks=1):
    """n_ranks = 0: one plain process without a communicator (the single-GPU layout)"""
    monkeypatch.setenv("GMG_COMM_TRANSPORT", "peer")
    monkeypatch.setenv("GMG_PEER_SLOT_MB", "16")
    uid = capi().Context.unique_id()
    assert uid.startswith(b"GMGPEER:")
    name = uid[len(b"GMGPEER:"):].split(b"\0")[0].decode()
    env = dict(os.environ)
    env.update(env_extra or {})
    outs = [str(tmp_path / f"n{n_ranks}_{part}_b{blocks}_rank{r}.json") for r in range(max(1, n_ranks))]
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "two_rank_worker.py"), str(r), str(n_ranks), uid.hex(), golden_dir,
                               outs[r]] + ([str(nacl), part, str(blocks)] if (nacl or blocks != 1) else []), env=env) for r in range(max(1, n_ranks))]
    try:
        for p in procs:
            assert p.wait(timeout=280) == 0
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
        try:
            os.unlink("/dev/shm" + name)  # rank 0 unlinks it on a clean exit
        except OSError:
            pass
    return [json.load(open(o)) for o in outs]


@pytest.mark.parametrize("n_ranks", [2, 3])
def test_six_adaptive_cycles_on_several_ranks(golden, golden_dir, tmp_path, monkeypatch, n_ranks):
    """Levels >= 1 are replicated and swept sequentially, so every rank count reproduces the mpirun=1 log."""
    G = golden["tests/gaussian-charges.mpirun=1"]["runs"][0]["cycles"]
    per_rank = run_ranks(n_ranks, golden_dir, tmp_path, monkeypatch)
    for reps in per_rank:
        assert [r["cg_iterations"] for r in reps] == [1, 6, 7, 6, 7, 7]
        for r, g in zip(reps, G):
            check_cycle(r, g, digits=9)
    # all ranks hold the same replicated results
    for other in per_rank[1:]:
        for a, b in zip(per_rank[0], other):
            for k in ("sol_l2", "refine_threshold", "energy_total", "starting_value"):
                assert rel_close(a[k], b[k], 13), k


def test_three_kernel_coarse_cg_on_two_and_three_ranks(golden, golden_dir, tmp_path, monkeypatch):
    """BASELINE config 2 (8 atoms, 45^3 level 0, SSOR smoother, the cluster run's marking) on two and three ranks:
    distributed three-kernel coarse CG over the peer transport (direction ring + x flush, halo entries of d stored into
    the neighbours' vectors, sums as one-workgroup kernels), two adaptive cycles.  Checked against the REFERENCE's
    cluster log (SSOR_run.o876223:14-31: cells, DoFs by level, starting values, solution norms) and against the same
    problem in the single-GPU layout (iteration counts)."""
    G = [c for run in golden["cluster/SSOR_run"]["runs"] if run.get("n_atoms") == 8 for c in run["cycles"]]
    two = run_ranks(2, golden_dir, tmp_path, monkeypatch, nacl=1)
    two += run_ranks(3, golden_dir, tmp_path, monkeypatch, nacl=1)  # unequal chunks, a middle rank with two neighbours
    one = run_ranks(0, golden_dir, tmp_path, monkeypatch, nacl=1)[0]
    # "Partition level 0 = auto": a 45^3 level 0 stays replicated (every rank runs the single-GPU coarse CG),
    # the system matrix and the outer CG vectors are still partitioned
    two += run_ranks(2, golden_dir, tmp_path, monkeypatch, nacl=1, part="auto")
    for reps in two:
        for c, (r, g) in enumerate(zip(reps, one)):
            assert r["dofs_by_level"] == g["dofs_by_level"] and r["cg_iterations"] == g["cg_iterations"]
            for k in ("sol_l1", "sol_l2", "sol_linf", "rhs_l2", "starting_value"):
                assert rel_close(r[k], g[k], 9), (k, r[k], g[k])
            ref = G[c]
            assert r["active_cells"] == ref["active_cells"] and r["dofs_by_level"] == ref["dofs_by_level"]
            assert abs(r["starting_value"] - ref["starting_value"]) <= (0.6e-6 if c == 0 else 5e-10)
            for k in ("sol_l1", "sol_l2", "sol_linf"):
                assert rel_close(r[k], ref[k], 8), (c, k, r[k], ref[k])


@pytest.mark.parametrize("n_ranks", [2, 3])
def test_block_ssor_swept_by_the_ranks(golden_dir, tmp_path, monkeypatch, n_ranks):
    """SSOR on N ranks as the reference applies it: N blocks, one per rank.  Every rank sweeps only its block and the
    pieces are all-gathered; the result must be what ONE process computes with the same N blocks (six adaptive cycles
    of the 2-atom problem: multi-level V-cycles with edge matrices)."""
    ranks = run_ranks(n_ranks, golden_dir, tmp_path, monkeypatch, blocks=n_ranks)
    one = run_ranks(0, golden_dir, tmp_path, monkeypatch, blocks=n_ranks)[0]
    for reps in ranks:
        for r, g in zip(reps, one):
            assert r["dofs_by_level"] == g["dofs_by_level"] and r["cg_iterations"] == g["cg_iterations"]
            for k in ("sol_l1", "sol_l2", "sol_linf", "starting_value", "refine_threshold", "energy_total"):
                assert rel_close(r[k], g[k], 9), (k, r[k], g[k])
