"""Pins the CPU oracle (oracle/) against the reference's committed golden logs.

Every number below is a line the reference itself printed (tests/golden/reference_logs.json,
extracted by tests/golden/make_golden.py from the reference's .output / cluster logs).
Tolerance: the printed precision of the golden (11 significant digits for the 3D logs,
6 for the older 2D/3D logs).
"""
import os

import numpy as np
import pytest

from conftest import rel_close
from oracle import gmg_oracle as go
from oracle import step50_oracle as so


def _norms(x):
    return float(np.abs(x).sum()), float(np.sqrt((x * x).sum())), float(np.abs(x).max())


def _check_cycle(g, r, x, digits):
    assert r["status"] == go.OK
    assert r["iterations"] == g["cg_iterations"]
    # "Starting value" is printed std::fixed (absolute digits), src/step-50.cc:1009
    assert abs(r["starting_value"] - g["starting_value"]) < 0.6 * 10.0 ** (-min(digits, 6))
    l1, l2, li = _norms(x)
    assert rel_close(l1, g["sol_l1"], digits)
    assert rel_close(l2, g["sol_l2"], digits)
    assert rel_close(li, g["sol_linf"], digits)


@pytest.fixture(scope="module")
def hier_two_atoms(golden_dir):
    q, p = so.read_lammps(os.path.join(golden_dir, "atom_n1_2.data"))
    h = so.build_gaussian_cycle0(q, p, left=0, right=1, h=0.25, vacuum=10, r_c=0.5, cutoff_param=3.5, n_q_rhs=4,
                                 bc="Exact")
    return q, p, h


def test_gaussian_charges_cycle0_inputs(golden, hier_two_atoms):
    """tests/gaussian-charges.mpirun=1.output:8-15 -- rhs and matrix norms (src/step-50.cc:946-952)."""
    g = golden["tests/gaussian-charges.mpirun=1"]["runs"][0]["cycles"][0]
    _, _, h = hier_two_atoms
    assert h.info["n_cells"] == g["active_cells"]
    assert h.info["dofs_by_level"] == g["dofs_by_level"]
    l1, l2, li = _norms(h.system_rhs)
    assert rel_close(l1, g["rhs_l1"], 11) and rel_close(l2, g["rhs_l2"], 11) and rel_close(li, g["rhs_linf"], 11)
    A = h.system_matrix
    assert A.nnz == (3 * 45 - 2) ** 3  # full 27-point pattern incl. stored zeros
    assert rel_close(A.l1_norm(), g["matrix_l1"], 11)
    assert rel_close(A.linfty_norm(), g["matrix_linf"], 11)
    assert rel_close(A.frobenius_norm(), g["matrix_frobenius"], 11)


def test_gaussian_charges_cycle0_solve_and_energy(golden, hier_two_atoms):
    """tests/gaussian-charges.mpirun=1.output:16-29 -- single level: V-cycle == coarse CG."""
    g = golden["tests/gaussian-charges.mpirun=1"]["runs"][0]["cycles"][0]
    q, p, h = hier_two_atoms
    mg = go.OracleMG(h, smoother=go.SSOR)
    r = mg.solve(h.system_rhs)
    _check_cycle(g, r, r["x"], 11)
    assert r["coarse_iterations"] == 112
    # final residual varies in the 7th digit with the rank count (line 18 of the three logs)
    assert abs(r["convergence_value"] - g["convergence_value"]) < 1e-5 * g["convergence_value"]
    xd = np.where(h.constrained, h.boundary_values, r["x"])  # constraints.distribute, :1016
    e = so.electrostatic_energy(h.lattice, xd, q, p, 0.5)
    assert rel_close(e["analytical"], g["energy_analytical"], 11)
    assert rel_close(e["short"], g["energy_short"], 11)
    assert rel_close(e["fe_long"], g["energy_fe_long"], 11)
    assert rel_close(e["self"], g["energy_self"], 11)
    assert rel_close(e["total"], g["energy_total_split"], 11)
    assert rel_close(e["abs_error"], g["energy_abs_error"], 10)


def test_cluster_8_atoms_cycle0(golden, golden_dir):
    """Cluster runs.../SSOR_run.o876223:15-22 -- BASELINE config 2, cycle 0."""
    g = golden["cluster/SSOR_run"]["runs"][0]["cycles"][0]
    q, p = so.read_lammps(os.path.join(golden_dir, "atom_n1_8.data"))
    h = so.build_gaussian_cycle0(q, p, left=0, right=1, h=0.25, vacuum=10, r_c=0.5, cutoff_param=3.5, n_q_rhs=1,
                                 bc="Inhomogeneous")
    assert h.info["dofs_by_level"] == g["dofs_by_level"]
    mg = go.OracleMG(h, smoother=go.SSOR)
    r = mg.solve(h.system_rhs)
    _check_cycle(g, r, r["x"], 11)
    assert r["coarse_iterations"] == 97


@pytest.mark.parametrize("key,dim,left,right,problem,digits", [
    ("tests_2D/step-16.mpirun=1", 2, 0.0, 1.0, "Step16", 6),
    ("tests_3D/step-16.mpirun=1", 3, 0.0, 1.0, "Step16", 6),
    ("tests_3D/gaussian-charges.mpirun=1", 3, -2.5, 2.5, "GaussianCharges", 6),
])
def test_uniform_five_level_jacobi_vcycle(golden, key, dim, left, right, problem, digits):
    """Five-level V-cycles of the older goldens (Jacobi omega=0.5 x 2)."""
    g = golden[key]["runs"][0]["cycles"][0]
    h = so.build_uniform_hierarchy(dim, left, right, 4, problem=problem)
    assert h.info["dofs_by_level"] == g["dofs_by_level"]
    mg = go.OracleMG(h, smoother=go.JACOBI)
    r = mg.solve(h.system_rhs)
    _check_cycle(g, r, r["x"], digits)
    assert rel_close(r["convergence_value"], g["convergence_value"], 4)


@pytest.mark.parametrize("run,opt", [(0, True), (1, False)])
def test_with_optimal_parameters_cycle0(golden, golden_dir, run, opt):
    """tests/test_with_optimal_parameters.mpirun=1.output:6-14 and :86-94."""
    g = golden["tests/test_with_optimal_parameters.mpirun=1"]["runs"][run]["cycles"][0]
    q, p = so.read_lammps(os.path.join(golden_dir, "atom_2.data"))
    h = so.build_uniform_hierarchy(3, -5.0, 5.0, 4, problem="GaussianCharges", charges=q, pos=p, n_q_rhs=1,
                                   cutoff_param=3.5, rhs_optimization=opt)
    mg = go.OracleMG(h, smoother=go.JACOBI)
    r = mg.solve(h.system_rhs)
    _check_cycle(g, r, r["x"], 11)
    assert rel_close(r["convergence_value"], g["convergence_value"], 6)


def test_ssor_changes_iteration_count_2d():
    """SGS(0.5) gives 5 iterations where the Jacobi golden has 7."""
    h = so.build_uniform_hierarchy(2, 0.0, 1.0, 4, problem="Step16")
    r = go.OracleMG(h, smoother=go.SSOR).solve(h.system_rhs)
    assert r["iterations"] == 5


def test_outer_nonconvergence_is_reported():
    h = so.build_uniform_hierarchy(2, 0.0, 1.0, 4, problem="Step16")
    r = go.OracleMG(h, smoother=go.JACOBI).solve(h.system_rhs, max_it=2)
    assert r["status"] == go.ERR_OUTER_NOCONV and r["iterations"] == 2


def test_transpose_spmv_matches_explicit_transpose():
    h = so.build_uniform_hierarchy(3, 0.0, 1.0, 3, problem="Step16")
    P = h.prolongations[-1]
    rng = np.random.default_rng(0)
    x = rng.standard_normal(P.n_rows)
    y = go.spmv_transpose(P, x)
    yt = go.spmv(P.transpose(), x)
    assert np.array_equal(y, yt)  # same summation order -> bit-exact
