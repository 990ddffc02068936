"""Host-side C++ (csrc/host: mirror of the reference's LaplaceProblem) against the numpy oracle
generator and the golden logs -- CPU only.  The host code produces the inputs of the hot path;
these tests show the two independent producers agree entry by entry."""
import os

import numpy as np
import pytest

from conftest import rel_close
from gpu_util import pkg
from oracle import gmg_oracle as go
from oracle import step50_oracle as so


def S():
    pkg().build.build_all()
    return pkg().step50


def _perm_compare(hm, om, perm, tol=1e-13):
    """host matrix (its DoF order) == oracle matrix (lexicographic order) under the DoF permutation."""
    assert hm.n_rows == om.n_rows and hm.nnz == om.nnz
    n = om.n_cols
    rows = np.repeat(np.arange(hm.n_rows), np.diff(hm.rowptr))
    hk = perm[rows].astype(np.int64) * n + perm[hm.col]
    o = np.argsort(hk)
    rows_o = np.repeat(np.arange(om.n_rows), np.diff(om.rowptr))
    ok = rows_o.astype(np.int64) * n + om.col
    oo = np.argsort(ok)
    assert np.array_equal(hk[o], ok[oo])
    assert np.abs(hm.val[o] - om.val[oo]).max() <= tol * np.abs(om.val).max()


def _lex_index(coords, lat):
    idx = np.rint((coords[:, :lat.dim] - lat.origin) / lat.h).astype(np.int64)
    out = idx[:, 0].copy()
    for d in range(1, lat.dim):
        out += idx[:, d] * lat.nv ** d
    return out


def test_parameter_reader_rejects_unknown_keys():
    with pytest.raises(RuntimeError):
        S().Problem("subsection Geometry\n set No such key = 1\nend\n")


def test_nacl_generator_matches_reference_atom_files(golden_dir):
    """The bench's synthetic atoms are the reference's atom/*.data files regenerated."""
    Sm = S()
    for n, name in [(1, "atom_n1_8.data"), (3, "atom_n3_216.data")]:
        p = Sm.Problem(Sm.prm_text(problem="GaussianCharges", dim=3))
        p.set_nacl_atoms(n)
        q, x = p.atoms()
        qr, xr = so.read_lammps(os.path.join(golden_dir, name))
        assert np.array_equal(q, qr) and np.array_equal(x, xr)
        p2 = Sm.Problem(Sm.prm_text(problem="GaussianCharges", dim=3))
        p2.read_lammps(os.path.join(golden_dir, name))
        q2, x2 = p2.atoms()
        assert np.array_equal(q2, qr) and np.array_equal(x2, xr)
    ref = "/root/reference/atom/atom_n5_1000.data"
    if os.path.exists(ref):
        p = Sm.Problem(Sm.prm_text(problem="GaussianCharges", dim=3))
        p.set_nacl_atoms(5)
        q, x = p.atoms()
        qr, xr = so.read_lammps(ref)
        assert np.array_equal(q, qr) and np.array_equal(x, xr)


def test_host_assembly_equals_oracle_generator_8_atoms(golden_dir):
    Sm = S()
    p = Sm.Problem(Sm.prm_text(left=0, right=1, mesh_size=0.25, vacuum=10, problem="GaussianCharges", dim=3,
                               bc="Inhomogeneous", cycles=1, r_c=0.5, cutoff=3.5, rhs_optimization=True, quad_rhs=1,
                               global_refinement=0))
    p.read_lammps(os.path.join(golden_dir, "atom_n1_8.data"))
    rep = p.run_cycle(0, on_device=False)
    q, x = p.atoms()
    oh = so.build_gaussian_cycle0(q, x, left=0, right=1, h=0.25, vacuum=10, r_c=0.5, cutoff_param=3.5, n_q_rhs=1,
                                  bc="Inhomogeneous")
    assert rep["dofs_by_level"] == [91125] and rep["active_cells"] == 85184
    perm = _lex_index(p.dof_coordinates(), oh.lattice)
    assert np.array_equal(np.sort(perm), np.arange(91125))
    hh = p.hierarchy()
    b = np.zeros(91125)
    b[perm] = hh.system_rhs
    assert np.abs(b - oh.system_rhs).max() <= 1e-13 * np.abs(oh.system_rhs).max()
    assert hh.system_matrix.nnz == (3 * 45 - 2) ** 3
    _perm_compare(hh.system_matrix, oh.system_matrix, perm)
    # level 0 is numbered lexicographically ("Level 0 numbering", default): the oracle generator's order, entry for entry
    _perm_compare(hh.level_matrices[0], oh.level_matrices[0], np.arange(91125))
    lv = hh.level_matrices[0]
    assert np.array_equal(lv.rowptr, oh.level_matrices[0].rowptr) and np.array_equal(lv.col, oh.level_matrices[0].col)
    cm = np.zeros(91125, dtype=bool)
    cm[perm] = hh.constrained
    assert np.array_equal(cm, oh.constrained)
    # "cell-wise": deal.II's first-touch order on level 0 as well -- then level 0 and the active mesh (identical at cycle 0)
    # share one numbering
    p2 = Sm.Problem(Sm.prm_text(left=0, right=1, mesh_size=0.25, vacuum=10, problem="GaussianCharges", dim=3,
                                bc="Inhomogeneous", cycles=1, r_c=0.5, cutoff=3.5, rhs_optimization=True, quad_rhs=1,
                                global_refinement=0, level0_numbering="cell-wise"))
    p2.read_lammps(os.path.join(golden_dir, "atom_n1_8.data"))
    p2.run_cycle(0, on_device=False)
    _perm_compare(p2.matrix("level", 0), oh.level_matrices[0], perm)


@pytest.mark.parametrize("dim,key", [(2, "tests_2D/step-16.mpirun=1"), (3, "tests_3D/step-16.mpirun=1")])
def test_host_five_level_hierarchy_reproduces_golden(golden, dim, key):
    """Host-produced operators + oracle solver == the reference's printed cycle 0."""
    g = golden[key]["runs"][0]["cycles"][0]
    Sm = S()
    p = Sm.Problem(Sm.prm_text(left=0, right=1, problem="Step16", dim=dim, bc="Homogeneous", cycles=1,
                               global_refinement=4, smoother="Jacobi"))
    rep = p.run_cycle(0, on_device=False)
    assert rep["dofs_by_level"] == g["dofs_by_level"] and rep["active_cells"] == g["active_cells"]
    h = p.hierarchy()
    r = go.OracleMG(h, smoother=go.JACOBI).solve(h.system_rhs)
    x = r["x"]
    assert r["iterations"] == g["cg_iterations"]
    assert rel_close(float(np.abs(x).sum()), g["sol_l1"], 6)
    assert rel_close(float(np.sqrt(x @ x)), g["sol_l2"], 6)
    assert rel_close(float(np.abs(x).max()), g["sol_linf"], 6)
    for P in h.prolongations:  # rows sum to 1 except where coarse boundary columns were zeroed
        s = np.zeros(P.n_rows)
        np.add.at(s, np.repeat(np.arange(P.n_rows), np.diff(P.rowptr)), P.val)
        assert s.max() <= 1.0 + 1e-15 and s.min() >= 0.0


def test_host_two_atom_exact_bc_golden(golden, golden_dir):
    """tests/gaussian-charges.mpirun=1.output:8-15: rhs and matrix norms from the host assembly."""
    g = golden["tests/gaussian-charges.mpirun=1"]["runs"][0]["cycles"][0]
    Sm = S()
    p = Sm.Problem(Sm.prm_text(left=0, right=1, mesh_size=0.25, vacuum=10, problem="GaussianCharges", dim=3, bc="Exact",
                               cycles=1, r_c=0.5, cutoff=3.5, rhs_optimization=True, quad_rhs=4, global_refinement=0))
    p.read_lammps(os.path.join(golden_dir, "atom_n1_2.data"))
    p.run_cycle(0, on_device=False)
    h = p.hierarchy()
    b = h.system_rhs
    assert rel_close(float(np.abs(b).sum()), g["rhs_l1"], 11)
    assert rel_close(float(np.sqrt(b @ b)), g["rhs_l2"], 11)
    assert rel_close(float(np.abs(b).max()), g["rhs_linf"], 11)
    A = h.system_matrix
    assert rel_close(float(np.sqrt((A.val ** 2).sum())), g["matrix_frobenius"], 11)


def _rc_variation_rhs(golden_dir, optimized, cutoff, on_device=False):
    """The setup of the reference's tests_rhs_rc_variation (2 atoms, [-2.5, 2.5]^3 in 16^3 cells, homogeneous
    boundary values, QGauss(2) for the rhs): returns (|b|_2, |b|_inf) of the assembled right-hand side."""
    S = pkg().step50
    p = S.Problem(S.prm_text(left=-2.5, right=2.5, mesh_size=0.3125, vacuum=0, problem="GaussianCharges", dim=3, bc="Homogeneous",
                             cycles=1, r_c=0.5, cutoff=cutoff, rhs_optimization=optimized, quad_rhs=1, global_refinement=0,
                             smoother="Jacobi", densities_on_device=on_device))
    p.read_lammps(os.path.join(golden_dir, "atom_2.data"))
    p.run_cycle(0, on_device=on_device)
    b = p.hierarchy().system_rhs
    t = p.total_charge_density()
    return float(np.sqrt(b @ b)), float(np.abs(b).max()), float(np.sqrt(t @ t))


def _cutoff_table(golden_dir, name):
    rows = {}
    with open(os.path.join(golden_dir, name)) as fh:
        for line in fh:
            parts = line.split()
            if len(parts) == 2 and parts[0][0].isdigit():
                rows[float(parts[0])] = float(parts[1])
    return rows


def check_rc_variation(golden, golden_dir, on_device):
    g = golden["tests_rhs_rc_variation/rc_variation.mpirun=1"]["runs"][0]["cycles"][0]
    l2, linf, tot = _rc_variation_rhs(golden_dir, False, 3.0, on_device)
    assert rel_close(l2, g["rhs_l2"], 11) and rel_close(linf, g["rhs_linf"], 11)
    t2 = _cutoff_table(golden_dir, "RHS_Norm_value_comparison_L2.dat")
    ti = _cutoff_table(golden_dir, "RHS_Norm_value_comparison_Linf.dat")
    tt = _cutoff_table(golden_dir, "Total_charge_density_AbsErr_L2.dat")
    assert len(t2) == len(ti) == len(tt) == 17
    for cutoff in sorted(t2):
        if cutoff > 4.5:  # both tables print 0.000000000000 from there on
            continue
        o2, oi, ot = _rc_variation_rhs(golden_dir, True, cutoff, on_device)
        assert abs(abs(ot - tot) - tt[cutoff]) <= 1e-9, (cutoff, abs(ot - tot), tt[cutoff])  # printed to 1e-9
        # the tables carry 10 decimals, of which the reference printed 7 significant digits
        assert abs(abs(o2 - l2) - t2[cutoff]) <= 1.5e-10 + 1e-6 * t2[cutoff], (cutoff, abs(o2 - l2), t2[cutoff])
        assert abs(abs(oi - linf) - ti[cutoff]) <= 1.5e-10 + 1e-6 * ti[cutoff], (cutoff, abs(oi - linf), ti[cutoff])


def test_rhs_cutoff_lists_match_reference_error_table(golden, golden_dir):
    """SURVEY 8(f) N1 goldens: the rhs norms of tests_rhs_rc_variation/*.output, and for every cutoff
    2.0 .. 4.5 the absolute difference of the rhs norms with and without the per-cell atom lists
    (src/step-50.cc:260-306) as tabulated in Plotting/RHS_Norm_value_comparison_{L2,Linf}.dat, and of the l2 norm of the per-DoF
    integrated charge density (Plotting/Total_charge_density_AbsErr_L2.dat)."""
    check_rc_variation(golden, golden_dir, on_device=False)
