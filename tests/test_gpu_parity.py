"""GPU parity tests: the HIP path behind the C-ABI against the CPU oracle on the same inputs.

Bar (DESIGN.md "Parity"): operator applications, transfers and Jacobi/Chebyshev/SGS smoother
steps are compared BIT-EXACTLY (same CSR summation order, no FMA contraction on either side);
anything that contains a global reduction (dot products, CG) is compared to a relative
tolerance of 1e-12 on scalars and 1e-9 on solution vectors, and CG iteration counts must be
identical.
"""
import os

import numpy as np
import pytest

from conftest import rel_close
from gpu_util import capi
from oracle import gmg_oracle as go
from oracle import step50_oracle as so

pytestmark = pytest.mark.gpu


def _norms(x):
    return float(np.abs(x).sum()), float(np.sqrt((x * x).sum())), float(np.abs(x).max())


@pytest.fixture(scope="module")
def hier3():
    return so.build_uniform_hierarchy(3, 0.0, 1.0, 4, problem="Step16")


@pytest.fixture(scope="module")
def ctx3(hier3):
    c = capi().Context(len(hier3.level_matrices))
    c.load_hierarchy(hier3)
    yield c
    c.close()


@pytest.fixture(scope="module")
def hier45(golden_dir):
    q, p = so.read_lammps(os.path.join(golden_dir, "atom_n1_8.data"))
    return so.build_gaussian_cycle0(q, p, left=0, right=1, h=0.25, vacuum=10, r_c=0.5, cutoff_param=3.5, n_q_rhs=1,
                                    bc="Inhomogeneous")


@pytest.fixture(scope="module")
def ctx45(hier45):
    c = capi().Context(1)
    c.load_hierarchy(hier45)
    yield c
    c.close()


def test_blas1_and_norms(ctx3, hier3):
    n = hier3.system_matrix.n_rows
    rng = np.random.default_rng(1)
    a, b = rng.standard_normal(n), rng.standard_normal(n)
    x, y = ctx3.vector(n, a), ctx3.vector(n, b)
    assert abs(ctx3.dot(x, y) - float(a @ b)) <= 1e-12 * np.sqrt((a @ a) * (b @ b))
    l1, l2, li = ctx3.norms(x)
    assert abs(l1 - np.abs(a).sum()) <= 1e-12 * l1 and abs(l2 - np.sqrt(a @ a)) <= 1e-12 * l2 and li == np.abs(a).max()
    ctx3.add(y, 0.37, x)
    assert np.array_equal(y.download(), b + 0.37 * a)
    ctx3.sadd(y, -1.25, 3.0, x)
    assert np.array_equal(y.download(), -1.25 * (b + 0.37 * a) + 3.0 * a)
    ctx3.equ(y, -1.0, x)
    assert np.array_equal(y.download(), -a)
    assert not ctx3.all_zero(x)
    ctx3.set_zero(x)
    assert ctx3.all_zero(x)


@pytest.mark.parametrize("which", ["system", 4, 2, 0])
def test_spmv_bit_exact(ctx3, hier3, which):
    m = hier3.system_matrix if which == "system" else hier3.level_matrices[which]
    rng = np.random.default_rng(2)
    a = rng.standard_normal(m.n_cols)
    x, y = ctx3.vector(m.n_cols, a), ctx3.vector(m.n_rows)
    ctx3.spmv(capi().SYSTEM if which == "system" else which, y, x)
    assert np.array_equal(y.download(), go.spmv(m, a))


def test_spmv_45_bit_exact(ctx45, hier45):
    m = hier45.system_matrix
    rng = np.random.default_rng(3)
    a = rng.standard_normal(m.n_cols)
    x, y = ctx45.vector(m.n_cols, a), ctx45.vector(m.n_rows)
    ctx45.spmv(capi().SYSTEM, y, x)
    assert np.array_equal(y.download(), go.spmv(m, a))
    ctx45.spmv(0, y, x)
    assert np.array_equal(y.download(), go.spmv(hier45.level_matrices[0], a))


@pytest.mark.parametrize("level", [0, 2, 3])
def test_transfer_bit_exact(ctx3, hier3, level):
    P = hier3.prolongations[level]
    rng = np.random.default_rng(4)
    c, f, c0 = rng.standard_normal(P.n_cols), rng.standard_normal(P.n_rows), rng.standard_normal(P.n_cols)
    vc, vf = ctx3.vector(P.n_cols, c), ctx3.vector(P.n_rows)
    ctx3.prolongate(level, vf, vc)
    assert np.array_equal(vf.download(), go.spmv(P, c))
    vf.upload(f)
    vc.upload(c0)
    ctx3.restrict_and_add(level, vc, vf)
    assert np.array_equal(vc.download(), go.spmv_transpose(P, f, c0))


@pytest.mark.parametrize("kind,name", [(0, "jacobi"), (2, "chebyshev"), (1, "ssor")])
@pytest.mark.parametrize("from_zero", [True, False])
def test_smoother_step_bit_exact(ctx3, hier3, kind, name, from_zero):
    level = 3
    n = hier3.level_matrices[level].n_rows
    rng = np.random.default_rng(5)
    u0, rhs = rng.standard_normal(n), rng.standard_normal(n)
    mg = go.OracleMG(hier3, smoother=kind, omega=0.5, steps=2, cheb_degree=3)
    ctx3.set_smoother(kind, 0.5, 2, cheb_degree=3)
    ref = mg.smooth(level, u0, rhs, from_zero)
    u, r = ctx3.vector(n, u0), ctx3.vector(n, rhs)
    ctx3.smoother_step(level, u, r, from_zero)
    got = u.download()
    assert np.array_equal(got, ref), float(np.abs(got - ref).max())


def test_coarse_solve_45(ctx45, hier45):
    """MGCoarseGridIterativeSolver on the 45^3 lattice: 97 iterations."""
    b = hier45.system_rhs
    mg = go.OracleMG(hier45)
    x_ref, it_ref, res_ref, rc_ref = mg.coarse_solve(b)
    vb, vx = ctx45.vector(len(b), b), ctx45.vector(len(b))
    it, res, rc = ctx45.coarse_solve(vx, vb)
    assert rc == 0 and rc_ref == 0
    assert it == it_ref == 97
    assert abs(res - res_ref) <= 1e-6 * res_ref
    x = vx.download()
    assert np.abs(x - x_ref).max() <= 1e-9 * np.abs(x_ref).max()


def test_coarse_solve_zero_rhs_and_nonconvergence(ctx45, hier45):
    n = hier45.system_matrix.n_rows
    vb, vx = ctx45.vector(n, np.zeros(n)), ctx45.vector(n)
    it, res, rc = ctx45.coarse_solve(vx, vb)
    assert (it, res, rc) == (0, 0.0, 0)
    ctx45.set_coarse(1e-10, 5)
    vb.upload(hier45.system_rhs)
    it, res, rc = ctx45.coarse_solve(vx, vb)
    assert rc == capi().ERR_COARSE_NOCONV and it == 5
    ctx45.set_coarse(1e-10, 1000)


@pytest.mark.parametrize("kind", [0, 2, 1])
def test_vcycle_matches_oracle(ctx3, hier3, kind):
    n = hier3.system_matrix.n_rows
    rng = np.random.default_rng(6)
    src = rng.standard_normal(n) * (~hier3.constrained)
    mg = go.OracleMG(hier3, smoother=kind, cheb_degree=2)
    ctx3.set_smoother(kind, 0.5, 2, cheb_degree=2)
    ref, rc = mg.vcycle(src)
    vs, vd = ctx3.vector(n, src), ctx3.vector(n)
    ctx3.precondition(vd, vs)
    got = vd.download()
    assert rc == 0
    assert np.abs(got - ref).max() <= 1e-10 * np.abs(ref).max()


@pytest.mark.parametrize("key,dim,left,right,problem", [
    ("tests_2D/step-16.mpirun=1", 2, 0.0, 1.0, "Step16"),
    ("tests_3D/step-16.mpirun=1", 3, 0.0, 1.0, "Step16"),
    ("tests_3D/gaussian-charges.mpirun=1", 3, -2.5, 2.5, "GaussianCharges"),
])
def test_full_solve_against_golden_logs(golden, key, dim, left, right, problem):
    """Outer CG + five-level Jacobi V-cycle on the GPU reproduces the reference's printed
    iteration counts and norms (6 printed digits)."""
    g = golden[key]["runs"][0]["cycles"][0]
    h = so.build_uniform_hierarchy(dim, left, right, 4, problem=problem)
    c = capi().Context(len(h.level_matrices))
    c.load_hierarchy(h)
    c.set_smoother(capi().JACOBI, 0.5, 2)
    n = h.system_matrix.n_rows
    vb, vx = c.vector(n, h.system_rhs), c.vector(n)
    r = c.cg_solve(vx, vb)
    x = vx.download()
    assert r["status"] == 0 and r["iterations"] == g["cg_iterations"]
    assert abs(r["starting_value"] - g["starting_value"]) < 0.6e-6
    assert rel_close(r["convergence_value"], g["convergence_value"], 4)
    for got, key_ in zip(_norms(x), ("sol_l1", "sol_l2", "sol_linf")):
        assert rel_close(got, g[key_], 6)
    c.close()


def test_full_solve_45_against_cluster_log(golden, ctx45, hier45):
    """BASELINE config 2, cycle 0: Cluster runs.../SSOR_run.o876223:15-22 (11 digits)."""
    g = golden["cluster/SSOR_run"]["runs"][0]["cycles"][0]
    n = hier45.system_matrix.n_rows
    vb, vx = ctx45.vector(n, hier45.system_rhs), ctx45.vector(n)
    ctx45.stats_reset()
    r = ctx45.cg_solve(vx, vb)
    x = vx.download()
    assert r["status"] == 0 and r["iterations"] == g["cg_iterations"] == 1
    assert abs(r["starting_value"] - g["starting_value"]) < 0.6e-6
    for got, key_ in zip(_norms(x), ("sol_l1", "sol_l2", "sol_linf")):
        assert rel_close(got, g[key_], 11)
    assert ctx45.stats().coarse_iterations == 97


def test_jacobi_preconditioned_mode(ctx3, hier3):
    """prm 'Preconditioner = Jacobi' (src/step-50.cc:996-1004)."""
    n = hier3.system_matrix.n_rows
    mg = go.OracleMG(hier3)
    ref = mg.solve(hier3.system_rhs, precond=go.PRECOND_JACOBI)
    vb, vx = ctx3.vector(n, hier3.system_rhs), ctx3.vector(n)
    r = ctx3.cg_solve(vx, vb, precond=capi().PRECOND_JACOBI)
    assert r["iterations"] == ref["iterations"]
    assert np.abs(vx.download() - ref["x"]).max() <= 1e-9 * np.abs(ref["x"]).max()


def test_outer_nonconvergence_code(ctx3, hier3):
    n = hier3.system_matrix.n_rows
    ctx3.set_smoother(capi().JACOBI, 0.5, 2)
    vb, vx = ctx3.vector(n, hier3.system_rhs), ctx3.vector(n)
    r = ctx3.cg_solve(vx, vb, max_it=2)
    assert r["status"] == capi().ERR_OUTER_NOCONV and r["iterations"] == 2


def test_single_rank_communicator_path(hier45):
    """The distributed coarse CG (RCCL all-reduces, unfused direction kernel) on a 1-rank
    communicator must give the same iteration count as the fused single-GPU path."""
    c = capi().Context(1)
    c.comm_init(0, 1, capi().Context.unique_id())
    c.set_global_sizes(hier45.system_matrix.n_rows, hier45.level_matrices[0].n_rows)
    c.load_hierarchy(hier45)
    n = hier45.system_matrix.n_rows
    vb, vx = c.vector(n, hier45.system_rhs), c.vector(n)
    it, res, rc = c.coarse_solve(vx, vb)
    assert rc == 0 and it == 97
    x_ref, *_ = go.OracleMG(hier45).coarse_solve(hier45.system_rhs)
    assert np.abs(vx.download() - x_ref).max() <= 1e-9 * np.abs(x_ref).max()
    c.close()


@pytest.mark.parametrize("blocks,variant", [(1, "phase"), (3, "phase"), (16, "phase"), (1, "reg"), (3, "reg"), (16, "reg"), (1, "reg-ranges"), (3, "reg-ranges"), (1, "reg-nosplit"), (1, "phase-ranges"), (3, "phase-ranges"), (1, "phase-nosplit"), (3, "phase-nosplit"), (1, "wave"), (3, "wave"), (16, "wave"),
                                            (1, "dep"), (3, "dep"), (16, "dep"), (1, "dep-ranges"), (3, "dep-ranges"),
                                            (1, "ranges"), (3, "ranges"), (1, "sweep"), (3, "sweep")])
def test_block_ssor_matches_oracle_bit_exact(hier3, blocks, variant):
    """B-block SSOR == the reference's rank-local SGS on B ranks (block Jacobi across ranks), in the device
    variants: the four-wave sweep with its records staged through LDS (gmg_sgs_phase.hpp, the default) or loaded into registers (gmg_sgs_reg.hpp), the one-dependent-wave sweep (gmg_sgs_dep.hpp) and the one-wave sweep (gmg_sgs.hpp), each with a
    block's rows in one LDS range and with the LDS budget cut to 300 doubles so that every block is swept in many
    ranges (working sets written back / reloaded in between), and the generic CSR sweep."""
    level = 4
    n = hier3.level_matrices[level].n_rows
    rng = np.random.default_rng(7)
    u0, rhs = rng.standard_normal(n), rng.standard_normal(n)
    mg = go.OracleMG(hier3, smoother=go.SSOR, ssor_blocks=blocks)
    c = capi().Context(len(hier3.level_matrices))
    c.set_tuning(ssor_blocks=blocks)
    if variant in ("wave", "ranges"):
        c.set_option("sgs_disable_phase", 1)
    if variant.startswith("dep"):
        c.set_option("sgs_dep", 1)
    if variant.startswith("reg"):
        c.set_option("sgs_reg", 1)
    if variant in ("phase-nosplit", "reg-nosplit"):
        c.set_option("sgs_phase_nosplit", 1)
    if variant.endswith("ranges"):
        c.set_option("sgs_y_slots", 300)
    if variant == "sweep":
        c.set_option("sgs_disable_wave", 1)
    c.load_hierarchy(hier3)
    c.set_smoother(capi().SSOR, 0.5, 2)
    for from_zero in (True, False):
        ref = mg.smooth(level, u0, rhs, from_zero)
        u, r = c.vector(n, u0), c.vector(n, rhs)
        c.smoother_step(level, u, r, from_zero)
        assert np.array_equal(u.download(), ref)
    exact = go.OracleMG(hier3, smoother=go.SSOR).smooth(level, u0, rhs, True)
    assert (blocks == 1) == np.array_equal(exact, mg.smooth(level, u0, rhs, True))  # the blocks really decouple
    c.close()


def test_rejected_launch_is_reported(hier3):
    """A kernel launch the runtime rejects (here: the SSOR sweep asked for more dynamic LDS than a CU has) must come
    back as ERR_HIP from the entry point that enqueued it, not as a silently missing result."""
    c = capi().Context(len(hier3.level_matrices))
    c.load_hierarchy(hier3)
    c.set_smoother(capi().SSOR, 0.5, 2)
    n = hier3.level_matrices[3].n_rows
    u, r = c.vector(n, np.zeros(n)), c.vector(n, np.ones(n))
    c.smoother_step(3, u, r, True)  # fine
    c.set_option("sgs_lds_bytes_override", 400 * 1024)
    with pytest.raises(capi().GMGError) as exc:
        c.smoother_step(3, u, r, True)
    assert exc.value.code == capi().ERR_HIP
    c.set_option("sgs_lds_bytes_override", 0)
    c.smoother_step(3, u, r, True)  # the context stays usable
    assert np.isfinite(u.download()).all()
    c.close()
