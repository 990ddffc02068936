"""N > 1 path on the CPU (gloo, world_size 2): the partition and halo plans that upload()
hands to the GPU library (csrc/host/partition.h) drive a rank-parallel CG whose local SpMV is
the oracle's; ghost import = point-to-point sends per the plan, dot products = all-reduce.
The iteration count and the solution must equal the serial solve (CG/Jacobi are
partition independent)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from gpu_util import pkg


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _halo_exchange(loc, x, rank):
    """x = [owned | ghost]: fill the ghost tail from the neighbours (gmg_comm.hpp:halo_exchange)."""
    n = loc.n_rows
    reqs, so, ro = [], 0, 0
    for nb, sc, rc in zip(loc.neighbor_rank, loc.send_count, loc.recv_count):
        if sc:
            buf = torch.from_numpy(np.ascontiguousarray(x[loc.send_idx[so:so + sc]]))
            reqs.append(dist.isend(buf, int(nb)))
        so += sc
    recv = []
    for nb, sc, rc in zip(loc.neighbor_rank, loc.send_count, loc.recv_count):
        if rc:
            t = torch.empty(int(rc), dtype=torch.float64)
            dist.recv(t, int(nb))
            recv.append((ro, t))
        ro += rc
    for r in reqs:
        r.wait()
    for off, t in recv:
        x[n + off:n + off + len(t)] = t.numpy()


def _allreduce(v):
    t = torch.tensor([v], dtype=torch.float64)
    dist.all_reduce(t)
    return float(t.item())


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ["OMP_NUM_THREADS"] = "2"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import gmg_oracle as go

    S = pkg().step50
    p = S.Problem(S.prm_text(left=0, right=1, mesh_size=0.25, vacuum=4, problem="GaussianCharges", dim=3, bc="Inhomogeneous",
                             cycles=1, r_c=0.5, cutoff=3.5, rhs_optimization=True, quad_rhs=1, global_refinement=0))
    p.set_nacl_atoms(1)
    p.run_cycle(0, on_device=False)
    loc = p.localize("level", 0, rank, world)
    b_full = p.vector("rhs")
    n, b0 = loc.n_rows, loc.row_begin
    b = b_full[b0:b0 + n]
    # SolverCG, identity preconditioner, zero start, tol 1e-10 (the coarse solver, src/step-50.cc:960-967)
    x = np.zeros(n)
    g = -b
    d = np.zeros(loc.n_cols)
    res = np.sqrt(_allreduce(float(g @ g)))
    d[:n] = -g
    gh = res * res
    it = 0
    while res > 1e-10 and it < 1000:
        it += 1
        _halo_exchange(loc, d, rank)
        h = go.spmv(loc, d)
        alpha = gh / _allreduce(float(d[:n] @ h))
        x += alpha * d[:n]
        g += alpha * h
        res = np.sqrt(_allreduce(float(g @ g)))
        if res <= 1e-10:
            break
        beta = gh
        gh = res * res
        beta = gh / beta
        d[:n] = beta * d[:n] - g
    if rank == 0:
        full = p.hierarchy()
        xs, its, _, _ = go.OracleMG(full).coarse_solve(b_full)
        out["serial_its"] = its
        out["serial_x"] = xs[b0:b0 + n]
    out[f"its{rank}"] = it
    out[f"x{rank}"] = x
    out[f"plan{rank}"] = (loc.neighbor_rank.tolist(), loc.send_count.tolist(), loc.recv_count.tolist(), int(n), int(loc.n_cols))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_cg_matches_serial():
    pkg().build.build_all()
    from oracle import gmg_oracle as go

    go.build()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    assert out["its0"] == out["its1"] == out["serial_its"]
    assert np.abs(out["x0"] - out["serial_x"]).max() <= 1e-9 * np.abs(out["serial_x"]).max()
    nb0, sc0, rc0, n0, nc0 = out["plan0"]
    nb1, sc1, rc1, n1, nc1 = out["plan1"]
    assert nb0 == [1] and nb1 == [0] and sc0 == rc1 and sc1 == rc0 and nc0 == n0 + rc0[0] and nc1 == n1 + rc1[0]


def test_partition_covers_every_row_and_ghosts_are_consistent():
    """Plans of all ranks agree pairwise (what rank a sends to b is what b expects from a) and
    the local matrices reassemble the global one, for 1..5 ranks."""
    S = pkg().step50
    pkg().build.build_all()
    p = S.Problem(S.prm_text(left=0, right=1, problem="Step16", dim=3, bc="Homogeneous", cycles=1, global_refinement=3))
    p.run_cycle(0, on_device=False)
    G = p.matrix("system")
    rng = np.random.default_rng(0)
    x = rng.standard_normal(G.n_rows)
    from oracle import gmg_oracle as go

    y_ref = go.spmv(G, x)
    for world in (1, 2, 3, 5):
        locs = [p.localize("system", 0, r, world) for r in range(world)]
        assert sum(l.n_rows for l in locs) == G.n_rows
        y = np.zeros(G.n_rows)
        for r, l in enumerate(locs):
            xl = np.concatenate([x[l.row_begin:l.row_begin + l.n_rows], x[l.ghost_global]])
            y[l.row_begin:l.row_begin + l.n_rows] = go.spmv(l, xl)
            so = 0
            for nb, sc in zip(l.neighbor_rank, l.send_count):
                peer = locs[int(nb)]
                k = list(peer.neighbor_rank).index(r)
                ro = int(sum(peer.recv_count[:k]))
                assert peer.recv_count[k] == sc
                sent_global = l.row_begin + l.send_idx[so:so + sc]
                assert np.array_equal(sent_global, peer.ghost_global[ro:ro + sc])
                so += sc
        assert np.array_equal(y, y_ref)
