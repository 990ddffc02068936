"""The plane-by-plane lattice kernel (csrc/gmg_lattice.hpp: spmv_lattice_kernel) against the oracle, bit for bit.

The level-0 operator of every BASELINE config is a 27-point stencil on an n^3 vertex lattice (45^3 ... 121^3);
the reference applies it as a CSR vmult (src/step-50.cc:962-967, LA::MPI::SparseMatrix).  Here synthetic lattices of
awkward shapes exercise what the cubes do not: lines shorter than a unit of 124 rows, plane sizes that are / are not
multiples of 124, a last plane step that is cut by the end of the interior, one and several segments per XCD slab, rows
whose neighbours are missing (domain boundary: the CSR row is shorter, the class table holds +0.0 there)."""
from types import SimpleNamespace

import numpy as np
import pytest

from gpu_util import capi
from oracle import gmg_oracle as go

pytestmark = pytest.mark.gpu


def lattice_operator(nx, ny, nz, rng, dirichlet=True):
    """27-point operator on an nx x ny x nz lattice, lexicographic numbering (x fastest), CSR with ascending columns.
    Boundary vertices are Dirichlet rows (diagonal by vertex type, stored zeros towards their existing neighbours);
    interior rows have zeros in the columns of boundary vertices (the eliminated couplings the reference keeps as stored
    zeros); the interior coefficients are the Q1 Laplace stencil's values times h."""
    h = 0.25
    w = np.empty((3, 3, 3))
    for dz in range(3):
        for dy in range(3):
            for dx in range(3):
                m = abs(dz - 1) + abs(dy - 1) + abs(dx - 1)
                w[dz, dy, dx] = h * (8.0 / 3.0, 0.0, -1.0 / 6.0, -1.0 / 12.0)[m]
    z, y, x = np.meshgrid(np.arange(nz, dtype=np.int32), np.arange(ny, dtype=np.int32), np.arange(nx, dtype=np.int32), indexing="ij")
    x, y, z = x.ravel(), y.ravel(), z.ravel()
    n = nx * ny * nz
    bnd = ((x == 0) | (x == nx - 1) | (y == 0) | (y == ny - 1) | (z == 0) | (z == nz - 1)) if dirichlet else np.zeros(n, bool)
    kind = (x == 0).astype(np.int8) + (x == nx - 1) + (y == 0) + (y == ny - 1) + (z == 0) + (z == nz - 1)
    # (n, 27) tables in offset order = ascending column order inside a row: no sort needed
    ok = np.empty((n, 27), dtype=bool)
    col = np.empty((n, 27), dtype=np.int32)
    val = np.empty((n, 27), dtype=np.float64)
    row = np.arange(n, dtype=np.int64)
    j = 0
    for dz in (-1, 0, 1):
        for dy in (-1, 0, 1):
            for dx in (-1, 0, 1):
                o = (x + dx >= 0) & (x + dx < nx) & (y + dy >= 0) & (y + dy < ny) & (z + dz >= 0) & (z + dz < nz)
                c = row + (dx + nx * dy + nx * ny * dz)
                cc = np.where(o, c, 0)
                v = np.where(bnd | bnd[cc], 0.0, w[dz + 1, dy + 1, dx + 1])  # eliminated rows / columns: stored zeros
                if dz == dy == dx == 0:
                    v = np.where(bnd, h * (4.0 / 3.0) / np.maximum(kind, 1), w[1, 1, 1])
                ok[:, j], col[:, j], val[:, j] = o, cc, v
                j += 1
    rp = np.zeros(n + 1, dtype=np.int64)
    rp[1:] = np.cumsum(ok.sum(axis=1))
    return SimpleNamespace(n_rows=n, n_cols=n, rowptr=rp, col=col[ok], val=val[ok], nnz=int(rp[-1]))


# (shape, lattice kernel expected): thin lattices are mostly boundary rows -- SELL padding beyond 12 % keeps them on the CSR
# row-window kernel, which must still agree
SHAPES = [((37, 23, 19), True), ((9, 9, 200), None), ((31, 8, 40), None), ((125, 3, 30), None), ((45, 45, 45), True), ((64, 33, 21), True),
          ((124, 31, 12), True), ((40, 31, 60), True)]


def is_lattice(layout):
    return layout >= 1 and bool((layout - 1) & 32)



@pytest.mark.parametrize("shape,expect", SHAPES)
@pytest.mark.parametrize("segments", [0, 1, 3])
def test_lattice_kernel_bit_exact(shape, expect, segments):
    rng = np.random.default_rng(sum(shape))
    m = lattice_operator(*shape, rng)
    x = rng.standard_normal(m.n_cols)
    c = capi().Context(1)
    if segments:
        c.set_option("lattice_segments", segments)
    c.set_level_matrix(0, m)
    lay = int(c.stats().spmv0_layout)
    if expect:
        assert is_lattice(lay), f"the lattice kernel was not chosen for {shape} (layout {lay})"
    vx, vy = c.vector(m.n_cols, x), c.vector(m.n_rows, np.full(m.n_rows, np.nan))
    c.spmv(0, vy, vx)
    y = vy.download()
    ref = go.spmv(m, x)
    assert not np.isnan(y).any(), f"{int(np.isnan(y).sum())} rows were not written, first {int(np.argmax(np.isnan(y)))}"
    bad = np.nonzero(y != ref)[0]
    assert bad.size == 0, (shape, bad[:10], y[bad[:3]], ref[bad[:3]])
    # the same operator through the pattern-run kernel: the two device paths agree bit for bit as well
    c2 = capi().Context(1)
    c2.set_option("disable_lattice", 1)
    c2.set_level_matrix(0, m)
    assert not is_lattice(int(c2.stats().spmv0_layout))
    vx2, vy2 = c2.vector(m.n_cols, x), c2.vector(m.n_rows)
    c2.spmv(0, vy2, vx2)
    assert np.array_equal(vy2.download(), y)
    c.close(); c2.close()


@pytest.mark.parametrize("shape", [(45, 45, 45), (64, 33, 21), (40, 31, 60)])
@pytest.mark.parametrize("cg", [False, True])
def test_lattice_kernel_with_several_columns_per_wave(shape, cg):
    """Lattices above ~360^3 have more (segment, column) pairs than the grid may have workgroups (one reduction partial each):
    the marching waves then take several pairs (spmv_lattice_kernel<CG, true>).  Forced here on small lattices by capping the
    marching workgroups at 8 (one per XCD): same bits as the oracle, for y = A x and inside the coarse CG."""
    rng = np.random.default_rng(11 + sum(shape))
    m = lattice_operator(*shape, rng)
    c = capi().Context(1)
    c.set_option("lattice_max_blocks", 8)
    c.set_option("lattice_segments", 2)
    if cg:
        c.set_tuning(cg_variant=2)
    c.set_level_matrix(0, m)
    assert is_lattice(int(c.stats().spmv0_layout))
    if not cg:
        x = rng.standard_normal(m.n_cols)
        vx, vy = c.vector(m.n_cols, x), c.vector(m.n_rows, np.full(m.n_rows, np.nan))
        c.spmv(0, vy, vx)
        assert np.array_equal(vy.download(), go.spmv(m, x))
    else:
        b = rng.standard_normal(m.n_rows)
        vb, vx = c.vector(m.n_rows, b), c.vector(m.n_rows)
        it, res, rc = c.coarse_solve(vx, vb)
        ident = np.arange(m.n_rows, dtype=np.int32)
        mg = go.OracleMG(SimpleNamespace(system_matrix=m, level_matrices=[m], edge_matrices=[None], prolongations=[], copy_global=[ident], copy_level=[ident]))
        x_ref, it_ref, res_ref, rc_ref = mg.coarse_solve(b)
        assert rc == 0 and rc_ref == 0 and it == it_ref, (it, it_ref)
        assert np.abs(vx.download() - x_ref).max() <= 1e-9 * np.abs(x_ref).max()
    c.close()


@pytest.mark.parametrize("shape", [(37, 23, 19), (45, 45, 45)])
def test_coarse_cg_on_the_lattice_kernel(shape):
    """The coarse CG (src/step-50.cc:962-967: SolverCG, identity preconditioner, 1e-10 absolute) through the CG = 2 variant
    of the kernel (d.h partials of the owned rows): iteration count = oracle, solution to 1e-9."""
    rng = np.random.default_rng(3)
    m = lattice_operator(*shape, rng)
    b = rng.standard_normal(m.n_rows)
    c = capi().Context(1)
    c.set_tuning(cg_variant=2)  # three-kernel iteration: SpMV + d.h partials in one kernel
    c.set_level_matrix(0, m)
    assert is_lattice(int(c.stats().spmv0_layout))
    vb, vx = c.vector(m.n_rows, b), c.vector(m.n_rows)
    it, res, rc = c.coarse_solve(vx, vb)
    assert rc == 0
    ident = np.arange(m.n_rows, dtype=np.int32)
    mg = go.OracleMG(SimpleNamespace(system_matrix=m, level_matrices=[m], edge_matrices=[None], prolongations=[], copy_global=[ident], copy_level=[ident]))
    x_ref, it_ref, res_ref, rc_ref = mg.coarse_solve(b)
    assert rc_ref == 0 and it == it_ref, (it, it_ref)
    x = vx.download()
    assert np.abs(x - x_ref).max() <= 1e-9 * np.abs(x_ref).max()
    c.close()


def test_cell_by_cell_numbering_of_the_host_side():
    """The host side numbers the lattice as deal.II does (a cell's new vertices in first-touch order): the first three lines
    of every plane and the first three planes are irregular, the interior is a window that repeats with the plane stride.
    BASELINE config 2's level 0 (45^3, src/step-50.cc:1504-1526 mesh): lattice kernel chosen, product bit-exact."""
    from gpu_util import pkg

    S = pkg().step50
    p = S.Problem(S.prm_text(left=0, right=1.0, mesh_size=0.25, vacuum=10, problem="GaussianCharges", dim=3, bc="Homogeneous",
                             cycles=1, r_c=0.5, cutoff=3.5, rhs_optimization=True, quad_rhs=1, global_refinement=0))
    p.set_nacl_atoms(1)
    p.run_cycle(0, on_device=False)
    m = p.matrix("level", 0)
    assert m.n_rows == 45 ** 3
    rng = np.random.default_rng(45)
    x = rng.standard_normal(m.n_cols)
    for seg in (0, 2):
        c = capi().Context(1)
        if seg:
            c.set_option("lattice_segments", seg)
        c.set_level_matrix(0, m)
        assert is_lattice(int(c.stats().spmv0_layout))
        vx, vy = c.vector(m.n_cols, x), c.vector(m.n_rows, np.full(m.n_rows, np.nan))
        c.spmv(0, vy, vx)
        y = vy.download()
        ref = go.spmv(m, x)
        bad = np.nonzero(~(y == ref))[0]
        assert bad.size == 0, (bad[:10], y[bad[:3]], ref[bad[:3]])
        c.close()
    p.close()


@pytest.mark.parametrize("shape", [(5, 5, 5), (17, 9, 12), (45, 45, 45), (61, 37, 29)])
def test_level0_formed_on_the_device_equals_the_csr_path(shape):
    """gmg_set_level_matrix_lattice: the level-0 matrix formed on the device from (vertices per direction,
    8 x 8 cell matrix) against the CSR the host assembles for the same lattice (src/step-50.cc:869-889 restated in
    lattice_operator's sibling below: every cell adds Ke, boundary rows keep sum |Ke[a][a]|): products bit for bit, same
    coarse-CG iteration count."""
    nx, ny, nz = shape
    h = 0.25
    # Q1 Laplace cell matrix on a cube of edge h (the reference's K_e), any symmetric 8 x 8 matrix would do
    rng = np.random.default_rng(nx * 1000 + ny)
    Ke = np.zeros((8, 8))
    for a in range(8):
        for b in range(8):
            m = bin(a ^ b).count("1")
            Ke[a, b] = h * (1.0 / 3.0, 0.0, -1.0 / 12.0, -1.0 / 12.0)[m]
    Ke += 1e-3 * np.diag(rng.random(8))  # (break the symmetry between the vertices of a cell: the sums must follow the cell order)
    # host-style assembly: cells in lexicographic order, every cell adds Ke; CSR pattern = all pairs sharing a cell
    n = nx * ny * nz
    bnd = np.zeros((nz, ny, nx), bool)
    bnd[0], bnd[-1], bnd[:, 0], bnd[:, -1], bnd[:, :, 0], bnd[:, :, -1] = True, True, True, True, True, True
    bnd = bnd.ravel()
    dense = {}
    import itertools
    for cz, cy, cx in itertools.product(range(nz - 1), range(ny - 1), range(nx - 1)):
        d = [cx + (a & 1) + nx * (cy + ((a >> 1) & 1)) + nx * ny * (cz + ((a >> 2) & 1)) for a in range(8)]
        for a in range(8):
            for b in range(8):
                key = (d[a], d[b])
                dense.setdefault(key, 0.0)
            if bnd[d[a]]:
                dense[(d[a], d[a])] += abs(Ke[a, a])
            else:
                for b in range(8):
                    if not bnd[d[b]]:
                        dense[(d[a], d[b])] += Ke[a, b]
    keys = sorted(dense)
    rows = np.array([k[0] for k in keys]); cols = np.array([k[1] for k in keys], dtype=np.int32)
    rp = np.zeros(n + 1, dtype=np.int64)
    np.add.at(rp, rows + 1, 1)
    m = SimpleNamespace(n_rows=n, n_cols=n, rowptr=np.cumsum(rp), col=cols, val=np.array([dense[k] for k in keys]), nnz=len(keys))
    x = rng.standard_normal(n)
    ref = go.spmv(m, x)
    c = capi().Context(1)
    c.set_level_matrix_lattice(0, shape, Ke)
    assert int(c.stats().spmv0_layout) == 127  # every layout bit + 64: formed on the device
    vx, vy = c.vector(n, x), c.vector(n, np.full(n, np.nan))
    c.spmv(0, vy, vx)
    y = vy.download()
    bad = np.nonzero(~(y == ref))[0]
    assert bad.size == 0, (shape, bad[:10], y[bad[:3]], ref[bad[:3]])
    b = rng.standard_normal(n)
    vb, vs = c.vector(n, b), c.vector(n)
    it, res, rc = c.coarse_solve(vs, vb)
    c2 = capi().Context(1)
    c2.set_tuning(cg_variant=2)
    c2.set_level_matrix(0, m)
    vb2, vs2 = c2.vector(n, b), c2.vector(n)
    it2, res2, rc2 = c2.coarse_solve(vs2, vb2)
    assert rc == 0 and rc2 == 0 and it == it2
    assert np.abs(vs.download() - vs2.download()).max() <= 1e-10 * np.abs(vs2.download()).max()
    c.close(); c2.close()
