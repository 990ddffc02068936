"""All six adaptive cycles of the reference's current regression test
(tests/gaussian-charges.mpirun=1.output: 2 atoms, Exact BC, QGauss(5) rhs, SSOR smoother) --
host-side C++ mesh/assembly/estimator + CPU oracle solver, CPU only.

This pins, against numbers the reference printed: the 2:1-balanced refinement, hanging-node
constraints, the level / edge (interface) matrices, the prebuilt transfers and copy indices,
the multi-level SSOR V-cycle with edge terms, the Kelly + residual estimator (float quirks
included: thresholds match to all 11 printed digits), the solution transfer (starting values)
and the electrostatic energy."""
import os

import numpy as np
import pytest

from conftest import rel_close
from gpu_util import pkg
from oracle import gmg_oracle as go

KEYS11 = ("rhs_l1", "rhs_l2", "rhs_linf", "matrix_l1", "matrix_linf")


def run_cycles(golden_dir, n_cycles, solve, communicator=False):
    S = pkg().step50
    pkg().build.build_all()
    p = S.Problem(S.prm_text(left=0, right=1, mesh_size=0.25, vacuum=10, problem="GaussianCharges", dim=3, bc="Exact",
                             cycles=n_cycles, r_c=0.5, cutoff=3.5, rhs_optimization=True, quad_rhs=4, global_refinement=0,
                             smoother="SSOR", partition_level0="always"))
    p.read_lammps(os.path.join(golden_dir, "atom_n1_2.data"))
    if communicator:  # the N > 1 layout on a 1-rank RCCL communicator, kept across cycles (gmg_reset)
        p.set_communicator(0, 1, pkg().capi.Context.unique_id())
    return p, [solve(p, c) for c in range(n_cycles)]


def check_cycle(r, g, digits=11):
    assert r["active_cells"] == g["active_cells"]
    assert r["dofs"] == g["dofs"] and r["dofs_by_level"] == g["dofs_by_level"]
    assert r["cg_iterations"] == g["cg_iterations"]
    assert abs(r["starting_value"] - g["starting_value"]) < 0.6e-10
    assert abs(r["convergence_value"] - g["convergence_value"]) <= 2e-5 * g["convergence_value"]
    for k in ("sol_l1", "sol_l2", "sol_linf", "refine_threshold", "energy_norm_error"):
        assert rel_close(r[k], g[k], digits), (k, r[k], g[k])
    for k, gk in (("energy_analytical", "energy_analytical"), ("energy_short", "energy_short"),
                  ("energy_fe_long", "energy_fe_long"), ("energy_self", "energy_self"), ("energy_total", "energy_total_split")):
        assert rel_close(r[k], g[gk], digits), (k, r[k], g[gk])


def test_six_adaptive_cycles_host_plus_oracle(golden, golden_dir):
    G = golden["tests/gaussian-charges.mpirun=1"]["runs"][0]["cycles"]

    def solve(p, cycle):
        p.run_cycle(cycle, on_device=False)
        h = p.hierarchy()
        b, A = h.system_rhs, h.system_matrix
        r = go.OracleMG(h, smoother=go.SSOR).solve(b, x0=p.vector("initial_guess"))
        assert r["status"] == go.OK
        x = r["x"]
        rep = p.finish_cycle_with(x)
        rep.update(cg_iterations=r["iterations"], starting_value=r["starting_value"], convergence_value=r["convergence_value"],
                   sol_l1=float(np.abs(x).sum()), sol_l2=float(np.sqrt(x @ x)), sol_linf=float(np.abs(x).max()))
        g = G[cycle]
        assert rel_close(float(np.abs(b).sum()), g["rhs_l1"], 11) and rel_close(float(np.sqrt(b @ b)), g["rhs_l2"], 11)
        assert rel_close(float(np.sqrt((A.val ** 2).sum())), g["matrix_frobenius"], 10)
        if cycle > 0:  # edge matrices exist and act only on refinement edges
            assert any(I.nnz > 0 for I in h.edge_matrices[1:])
            assert h.edge_matrices[0].nnz == 0
        return rep

    _, reps = run_cycles(golden_dir, 6, solve)
    assert [r["cg_iterations"] for r in reps] == [1, 6, 7, 6, 7, 7]
    for r, g in zip(reps, G):
        check_cycle(r, g)


@pytest.mark.gpu
@pytest.mark.parametrize("communicator", [False, True])
def test_six_adaptive_cycles_on_mi355x(golden, golden_dir, communicator):
    """The same six cycles with the solve on the GPU through the C-ABI (host SolverCG over
    gmg_precondition: multi-level SSOR V-cycle with edge matrices, device-resident coarse CG)."""
    G = golden["tests/gaussian-charges.mpirun=1"]["runs"][0]["cycles"]
    _, reps = run_cycles(golden_dir, 6, lambda p, c: p.run_cycle(c, on_device=True), communicator)
    assert [r["cg_iterations"] for r in reps] == [1, 6, 7, 6, 7, 7]
    for r, g in zip(reps, G):
        check_cycle(r, g)
        for k in KEYS11:
            assert rel_close(r[k], g[k], 11), k
        assert rel_close(r["matrix_frobenius"], g["matrix_frobenius"], 10)
