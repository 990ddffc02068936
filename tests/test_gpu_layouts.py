"""Edge cases of the device layouts behind gmg_spmv (all compared bit-exactly with the oracle):
SELL-64 with and without each compression, the CSR row-window kernel incl. its long-row path,
empty rows, row counts that are not multiples of 64, rectangular operators; plus
size-independent properties of the full-size BASELINE operator (121^3 level 0)."""
from types import SimpleNamespace

import numpy as np
import pytest

from gpu_util import capi, pkg
from oracle import gmg_oracle as go

pytestmark = pytest.mark.gpu


def csr(n_rows, n_cols, rows):
    """rows: list of (cols, vals) per row"""
    rp = np.zeros(n_rows + 1, dtype=np.int64)
    col, val = [], []
    for i, (c, v) in enumerate(rows):
        col += list(c)
        val += list(v)
        rp[i + 1] = len(col)
    return SimpleNamespace(n_rows=n_rows, n_cols=n_cols, rowptr=rp, col=np.array(col, dtype=np.int32),
                           val=np.array(val, dtype=np.float64), nnz=len(col))


def apply_level0(m, x, options=()):
    c = capi().Context(1)
    for key in options:
        c.set_option(key, 1)
    c.set_level_matrix(0, m)
    vx, vy = c.vector(m.n_cols, x), c.vector(m.n_rows)
    c.spmv(0, vy, vx)
    y = vy.download()
    lay = int(c.stats().spmv0_layout)
    c.close()
    return y, lay


def banded(n, width, rng, distinct=None, spread=1, n_classes=0):
    rows = []
    vectors = [rng.choice(distinct, width) for _ in range(n_classes)] if n_classes else None
    for i in range(n):
        c = sorted({min(n - 1, max(0, i + spread * (k - width // 2))) for k in range(width)} | {i})
        if vectors is not None and len(c) == width:
            v = vectors[(i * 7 + i // 64) % n_classes].copy()
        else:
            v = rng.standard_normal(len(c)) if distinct is None else rng.choice(distinct, len(c))
        v[c.index(i)] = 4.0 + abs(v[c.index(i)])
        rows.append((c, v))
    return csr(n, n, rows)


@pytest.mark.parametrize("case,expect", [
    ("val8_col16_runs", 1 + 2 + 4 + 8), ("val8_col16_lattice", 1 + 2 + 4 + 8 + 16 + 32), ("val8_col16_rowclass", 1 + 2 + 4 + 8 + 16), ("val8_col16_rowclass_rr", 1 + 2 + 4 + 8 + 16), ("val8_col16_rowclass_codes", 1 + 2 + 4 + 8), ("val8_col16", 1 + 2 + 4), ("val64_col16", 1 + 4), ("val8_col32", 1 + 2), ("val64_col32", 1),
])
def test_sell_variants_bit_exact(case, expect):
    """27 consecutive columns per row are nine runs of three: with 8-bit value codes that operator goes to
    the pattern-run kernel (layout bit 8) unless the diagnostic switch keeps it on the per-entry kernel.  When the rows
    have few distinct coefficient vectors (a lattice: here 7 of them) the kernel runs with row classes (bit 16: one
    class byte per row instead of one code per entry) unless those are switched off; random values per entry give far
    more than 96 classes and stay on the codes.  `_rr`: the round-robin slice order large lattices get (here forced; a
    bigger operator so that every wave owns several pairs of slices).  `_lattice`: nine runs of three with strides 3 and 9
    ARE a 3 x 3 x n lattice in the kernel's terms: with few classes the plane-by-plane kernel (bit 32) takes the interior rows;
    the other row-class cases switch it off to stay on the kernel they were written for."""
    rng = np.random.default_rng(11)
    n = (5000 if not case.endswith("_rr") else 700000) + 37  # not a multiple of 64
    distinct = np.array([-1 / 6, -1 / 12, 8 / 3, 0.0, 1.25]) if "val8" in case else None
    spread = 1 if "col16" in case else 3000  # 27 * 3000 > 65535 columns within a slice
    m = banded(n if spread == 1 else 200000, 27, rng, distinct, spread, n_classes=7 if ("rowclass" in case or "lattice" in case) else 0)
    x = rng.standard_normal(m.n_cols)
    opts = ("disable_sellp",) if case == "val8_col16" else ("disable_rowclass",) if case.endswith("_codes") else ("sellp_rr",) if case.endswith("_rr") else ()
    if "rowclass" in case:
        opts += ("disable_lattice",)
    y, lay = apply_level0(m, x, opts)
    assert lay == expect, (case, lay)
    assert np.array_equal(y, go.spmv(m, x))


@pytest.mark.parametrize("kind,name", [(0, "jacobi"), (2, "chebyshev")])
@pytest.mark.parametrize("from_zero", [True, False])
def test_smoother_epilogues_on_the_pattern_run_kernel(kind, name, from_zero):
    """Damped-Jacobi and Chebyshev steps are epilogues of the SpMV kernels: here on an operator that
    is served by the pattern-run kernel (27 consecutive columns, 5 distinct values), as level 1 of a
    synthetic two-level hierarchy, bit for bit against the oracle's smoother."""
    rng = np.random.default_rng(21)
    n = 64 * 40 + 17
    A1 = banded(n, 27, rng, np.array([-1 / 6, -1 / 12, 8 / 3, 0.0, 1.25]), 1)
    A0 = banded(64, 3, rng)
    P = csr(n, 64, [([i % 64], [1.0]) for i in range(n)])
    empty = SimpleNamespace(n_rows=0, n_cols=0, rowptr=np.zeros(1, dtype=np.int64), col=np.zeros(0, dtype=np.int32),
                            val=np.zeros(0), nnz=0)
    idx = np.arange(n, dtype=np.int32)
    hier = SimpleNamespace(system_matrix=A1, level_matrices=[A0, A1], edge_matrices=[None, None], prolongations=[P],
                           copy_global=[np.zeros(0, dtype=np.int32), idx], copy_level=[np.zeros(0, dtype=np.int32), idx])
    del empty
    c = capi().Context(2)
    c.load_hierarchy(hier)
    c.set_smoother(kind, 0.5, 2, cheb_degree=3)
    mg = go.OracleMG(hier, smoother=kind, omega=0.5, steps=2, cheb_degree=3)
    u0, rhs = rng.standard_normal(n), rng.standard_normal(n)
    ref = mg.smooth(1, u0, rhs, from_zero)
    u, r = c.vector(n, u0), c.vector(n, rhs)
    c.smoother_step(1, u, r, from_zero)
    got = u.download()
    c.close()
    assert np.array_equal(got, ref), float(np.abs(got - ref).max())


def _random_sparse(n, per_row, rng, reach):
    """rows with ~per_row random columns within +-reach of the diagonal (ascending, diagonal included, some stored zeros)"""
    rows = []
    for i in range(n):
        c = sorted(set(int(x) for x in np.clip(i + rng.integers(-reach, reach + 1, per_row), 0, n - 1)) | {i})
        v = rng.standard_normal(len(c))
        v[rng.random(len(c)) < 0.15] = 0.0
        v[c.index(i)] = 4.0 + abs(v[c.index(i)])
        rows.append((c, v))
    return csr(n, n, rows)


@pytest.mark.parametrize("sweep", ["phase", "reg", "dep"])
@pytest.mark.parametrize("blocks", [1, 4])
@pytest.mark.parametrize("shape", ["chain", "wide-stages", "wide-rows", "scattered"])
def test_ssor_on_synthetic_dependency_shapes(shape, blocks, sweep):
    """The SSOR sweep on operators whose dependency graphs stress the four-wave sweep (gmg_sgs_phase.hpp) in ways the mesh
    hierarchies do not, bit for bit against the oracle: a band of 27 consecutive columns (every stage is ONE row, every
    lower neighbour is late), a sparse operator with hundreds of independent rows per stage (stages cut into steps of 32
    rows), rows of 61 entries (more than a record holds: the plan falls back to the one-wave sweep), and scattered columns
    (heads, T1 and T2 of every length)."""
    rng = np.random.default_rng({"chain": 31, "wide-stages": 32, "wide-rows": 33, "scattered": 34}[shape])
    n = 3000 + 17
    if shape == "chain":
        A1 = banded(n, 27, rng)
    elif shape == "wide-stages":
        A1 = _random_sparse(n, 3, rng, 700)
    elif shape == "wide-rows":
        A1 = banded(n, 61, rng)
    else:
        A1 = _random_sparse(n, 14, rng, 60)
    A0 = banded(64, 3, rng)
    P = csr(n, 64, [([i % 64], [1.0]) for i in range(n)])
    idx = np.arange(n, dtype=np.int32)
    hier = SimpleNamespace(system_matrix=A1, level_matrices=[A0, A1], edge_matrices=[None, None], prolongations=[P],
                           copy_global=[np.zeros(0, dtype=np.int32), idx], copy_level=[np.zeros(0, dtype=np.int32), idx])
    c = capi().Context(2)
    c.set_tuning(ssor_blocks=blocks)
    if sweep == "dep":  # one dependent wave fed by three preparing waves (gmg_sgs_dep.hpp)
        c.set_option("sgs_dep", 1)
    if sweep == "reg":  # the records loaded straight into registers (gmg_sgs_reg.hpp); default: staged through LDS regions (gmg_sgs_phase.hpp)
        c.set_option("sgs_reg", 1)
    c.load_hierarchy(hier)
    c.set_smoother(capi().SSOR, 0.5, 2)
    mg = go.OracleMG(hier, smoother=go.SSOR, omega=0.5, steps=2, ssor_blocks=blocks)
    u0, rhs = rng.standard_normal(n), rng.standard_normal(n)
    for from_zero in (True, False):
        ref = mg.smooth(1, u0, rhs, from_zero)
        u, r = c.vector(n, u0), c.vector(n, rhs)
        c.smoother_step(1, u, r, from_zero)
        got = u.download()
        assert np.array_equal(got, ref), float(np.abs(got - ref).max())
    c.close()


def test_more_than_256_values_falls_back_to_fp64_values():
    rng = np.random.default_rng(12)
    m = banded(3000, 27, rng)  # random values: thousands of distinct doubles
    x = rng.standard_normal(3000)
    y, lay = apply_level0(m, x)
    assert lay == 1 + 4  # SELL-64, fp64 values, 16-bit columns
    assert np.array_equal(y, go.spmv(m, x))
    m9 = banded(3000, 9, rng)  # 9 -> 12 entries per row would be 33 % padding: stays CSR
    y, lay = apply_level0(m9, x)
    assert lay == 0 and np.array_equal(y, go.spmv(m9, x))


def test_csr_window_irregular_empty_and_long_rows():
    rng = np.random.default_rng(13)
    n = 3000
    rows = []
    for i in range(n):
        k = [0, 1, 3, 40, 9][i % 5] if i != 1234 else 6000  # empty rows, ragged rows, one row wider than the LDS window
        c = np.sort(rng.choice(n if k < 3000 else 20000, size=k, replace=False)) if k else []
        rows.append((c, rng.standard_normal(k)))
    m = csr(n, 20000, rows)
    x = rng.standard_normal(20000)
    c = capi().Context(2)
    c.set_level_matrix(0, banded(2048, 3, rng))
    c.set_level_matrix(1, banded(n, 3, rng))
    c.set_prolongation(0, csr(n, 2048, [(cc[cc < 2048][:50], vv[:len(cc[cc < 2048][:50])]) for cc, vv in
                                        ((np.asarray(r[0], dtype=np.int64), np.asarray(r[1])) for r in rows)]))
    P = csr(n, 2048, [(cc[cc < 2048][:50], vv[:len(cc[cc < 2048][:50])]) for cc, vv in
                      ((np.asarray(r[0], dtype=np.int64), np.asarray(r[1])) for r in rows)])
    xc = rng.standard_normal(2048)
    vf, vc = c.vector(n), c.vector(2048, xc)
    c.prolongate(0, vf, vc)
    assert np.array_equal(vf.download(), go.spmv(P, xc))
    f, c0 = rng.standard_normal(n), rng.standard_normal(2048)
    vf.upload(f); vc.upload(c0)
    c.restrict_and_add(0, vc, vf)
    assert np.array_equal(vc.download(), go.spmv_transpose(P, f, c0))
    c.close()
    # the long row (6000 nnz > 4096-entry window) goes through the strided path: tolerance, not bit-exact
    y, lay = apply_level0(m, x)
    ref = go.spmv(m, x)
    assert lay == 0
    mask = np.ones(n, dtype=bool); mask[1234] = False
    assert np.array_equal(y[mask], ref[mask])
    assert abs(y[1234] - ref[1234]) <= 1e-12 * np.abs(m.val[m.rowptr[1234]:m.rowptr[1235]]).sum() * np.abs(x).max()


def test_invalid_operator_is_rejected():
    c = capi().Context(1)
    bad = csr(4, 4, [([0], [1.0]), ([7], [1.0]), ([2], [1.0]), ([3], [1.0])])  # column out of range
    with pytest.raises(capi().GMGError) as e:
        c.set_level_matrix(0, bad)
    assert e.value.code == capi().ERR_INVALID
    with pytest.raises(capi().GMGError):
        c.spmv(0, c.vector(4), c.vector(4))  # operator not set
    c.close()


@pytest.mark.parametrize("nacl,log", [(3, "cluster/SSOR_run"), (5, "cluster/SSOR_run"), (7, "cluster/SSOR_run"), (10, "cluster/SSOR_run"),
                                      (5, "cluster/without_opti"), (10, "cluster/without_opti")])
def test_cluster_runs_cycle0(golden, nacl, log):
    """Cycle 0 of the reference's cluster runs between the 8-atom and the 64k-atom case (BASELINE configs 3
    and 4 among them: 1000 and 8000 atoms; 53^3 .. 81^3 level 0), with and without the rhs cutoff lists
    ('Flag for RHS evaluation optimization'): every printed digit of the solve.  The single level makes the
    smoother irrelevant; the coarse CG is the fused variant below 200 k rows, the three-kernel one above."""
    from conftest import rel_close

    run = next(r for r in golden[log]["runs"] if r.get("n_atoms") == 8 * nacl ** 3)
    g = run["cycles"][0]
    S = pkg().step50
    p = S.Problem(S.prm_text(left=0, right=nacl, mesh_size=0.25, vacuum=10, problem="GaussianCharges", dim=3, bc="Inhomogeneous",
                             cycles=1, r_c=0.5, cutoff=3.5, rhs_optimization=log.endswith("SSOR_run"), quad_rhs=1,
                             global_refinement=0, smoother="Jacobi"))
    p.set_nacl_atoms(nacl)
    r = p.run_cycle(0, on_device=True)
    assert r["active_cells"] == g["active_cells"] and r["dofs_by_level"] == g["dofs_by_level"]
    assert r["cg_iterations"] == g["cg_iterations"] == 1
    assert abs(r["starting_value"] - g["starting_value"]) < 0.6e-6
    assert abs(r["convergence_value"] - g["convergence_value"]) <= 1e-4 * g["convergence_value"]
    for k in ("sol_l1", "sol_l2", "sol_linf", "rhs_l2"):
        if k in g:
            assert rel_close(r[k], g[k], 11), (k, r[k], g[k])
    p.close()


@pytest.fixture(scope="module")
def full_size():
    """BASELINE config 5, cycle 0: 121^3 level 0 (1 771 561 rows, 47 045 881 nnz) on the host."""
    S = pkg().step50
    p = S.Problem(S.prm_text(left=0, right=20, mesh_size=0.25, vacuum=10, problem="GaussianCharges", dim=3, bc="Inhomogeneous",
                             cycles=1, r_c=0.5, cutoff=3.5, rhs_optimization=True, quad_rhs=1, global_refinement=0, smoother="Jacobi"))
    p.set_nacl_atoms(20)
    rep = p.run_cycle(0, on_device=True)
    return p, rep


def test_full_size_cycle0_matches_cluster_log(golden, full_size):
    """SSOR_64k_atoms.o876224:14-22 -- every printed digit, at the reference's largest size."""
    from conftest import rel_close

    p, r = full_size
    g = golden["cluster/SSOR_64k_atoms"]["runs"][0]["cycles"][0]
    assert r["active_cells"] == g["active_cells"] and r["dofs_by_level"] == g["dofs_by_level"]
    assert r["cg_iterations"] == g["cg_iterations"] == 1
    assert abs(r["starting_value"] - g["starting_value"]) < 0.6e-6
    assert abs(r["convergence_value"] - g["convergence_value"]) <= 1e-4 * g["convergence_value"]
    for k in ("sol_l1", "sol_l2", "sol_linf"):
        assert rel_close(r[k], g[k], 11), k


def test_full_size_operator_properties(full_size):
    """Size-independent properties of the 121^3 operator on the GPU: linearity, symmetry,
    and the residual of the solve recomputed independently."""
    p, rep = full_size
    ctx = capi().Context.view(p.gmg_context())
    n = rep["dofs"]
    rng = np.random.default_rng(5)
    a, b = rng.standard_normal(n), rng.standard_normal(n)
    va, vb, vab = ctx.vector(n, a), ctx.vector(n, b), ctx.vector(n, 2.0 * a - 3.0 * b)
    ya, yb, yab = ctx.vector(n), ctx.vector(n), ctx.vector(n)
    for which in (0, capi().SYSTEM):
        ctx.spmv(which, ya, va); ctx.spmv(which, yb, vb); ctx.spmv(which, yab, vab)
        A_a, A_b, A_ab = ya.download(), yb.download(), yab.download()
        scale = np.abs(A_a).max() + np.abs(A_b).max()
        assert np.abs(A_ab - (2.0 * A_a - 3.0 * A_b)).max() <= 1e-13 * scale * 10          # linearity
        assert abs(ctx.dot(vb, ya) - ctx.dot(va, yb)) <= 1e-11 * abs(ctx.dot(va, ya))        # symmetry
    # the 121^3 level-0 product (plane-by-plane lattice kernel, layout bit 32) against the oracle, bit for bit
    assert (int(ctx.stats().spmv0_layout) - 1) & 32
    A0 = p.matrix("level", 0)
    ctx.spmv(0, ya, va)
    assert np.array_equal(ya.download(), go.spmv(A0, a))
    # residual of the converged solve: |b - A x| <= 1e-8 |b| recomputed with numpy on the host
    h = p.hierarchy()
    x = np.where(h.constrained, 0.0, p.vector("solution"))
    vx, vr = ctx.vector(n, x), ctx.vector(n)
    ctx.spmv(capi().SYSTEM, vr, vx)
    res = np.linalg.norm(h.system_rhs - vr.download())
    assert res <= 1.05e-8 * np.linalg.norm(h.system_rhs)
