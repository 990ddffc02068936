"""CPU oracle (TEST INFRASTRUCTURE ONLY) -- problem generator for the Step50 hot path.

This file is a numpy restatement of the *producers* of the hot path's inputs in the
reference (vinayak-gholap1993/Geometric-Multigrid-preconditioners-for-long-range-Coulomb-
interaction).  It exists so that the solver restatement in ``gmg_oracle.c`` can be pinned
against the reference's committed golden logs, which are whole-program outputs.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py``
may import anything under ``oracle/``.  The product path never does.

Reference lines restated here (all in /root/reference):
  * LAMMPS reader ......................... src/step-50.cc:181-258
  * atom -> cell cutoff lists ............. src/step-50.cc:260-306
  * Gaussian charge density at q-points ... src/step-50.cc:509-575
  * dipole moment / zeroed quadrupole ..... src/step-50.cc:577-644
  * constraints, boundary values, pattern . src/step-50.cc:646-732
  * system matrix / rhs assembly .......... src/step-50.cc:735-833
  * level matrices (no edge terms here) ... src/step-50.cc:835-933
  * lattice of cycle 0 .................... src/step-50.cc:1490-1527
  * problem functions ..................... include/step_50.h:216-386
  * energy post-processing ................ src/step-50.cc:1310-1420
Third-party behaviour restated (deal.II >= 9.0, not vendored in the reference):
  * ConstraintMatrix::distribute_local_to_global (diagonal of constrained rows gets
    |K_e(c,c)|, rhs gets -K_ic g_c), MGTransferPrebuilt::build_matrices (Q1 embedding,
    coarse-boundary columns zeroed), FE_Q vertex order (x fastest).

Scope: uniform lattices (cycle 0 of every golden file).  Adaptive cycles are produced by the
product's own host code and are pinned directly by the goldens (DoF counts, norms,
iteration counts); they are not restated a second time here.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np

# --------------------------------------------------------------------------- CSR helper


@dataclass
class CSR:
    """Plain CSR (int64 rowptr, int32 col, float64 val); explicit zeros are kept."""

    n_rows: int
    n_cols: int
    rowptr: np.ndarray
    col: np.ndarray
    val: np.ndarray

    @property
    def nnz(self) -> int:
        return int(self.rowptr[-1])

    def matvec(self, x: np.ndarray) -> np.ndarray:  # slow-ish numpy reference, small cases
        prod = self.val * x[self.col]
        y = np.zeros(self.n_rows)
        nz_rows = np.repeat(np.arange(self.n_rows), np.diff(self.rowptr))
        np.add.at(y, nz_rows, prod)
        return y

    def diagonal(self) -> np.ndarray:
        rows = np.repeat(np.arange(self.n_rows), np.diff(self.rowptr))
        d = np.zeros(self.n_rows)
        m = rows == self.col
        d[rows[m]] = self.val[m]
        return d

    def transpose(self) -> "CSR":
        rows = np.repeat(np.arange(self.n_rows, dtype=np.int64), np.diff(self.rowptr))
        return coo_to_csr(self.col.astype(np.int64), rows, self.val, self.n_cols, self.n_rows)

    # matrix norms as deal.II prints them (src/step-50.cc:950-952)
    def l1_norm(self) -> float:  # max column sum
        s = np.zeros(self.n_cols)
        np.add.at(s, self.col, np.abs(self.val))
        return float(s.max())

    def linfty_norm(self) -> float:  # max row sum
        rows = np.repeat(np.arange(self.n_rows), np.diff(self.rowptr))
        s = np.zeros(self.n_rows)
        np.add.at(s, rows, np.abs(self.val))
        return float(s.max())

    def frobenius_norm(self) -> float:
        return float(math.sqrt(np.sum(self.val * self.val)))


def coo_to_csr(rows, cols, vals, n_rows, n_cols) -> CSR:
    """Sum duplicates, keep explicit zeros, sort columns inside each row."""
    rows = np.asarray(rows, dtype=np.int64)
    cols = np.asarray(cols, dtype=np.int64)
    vals = np.asarray(vals, dtype=np.float64)
    key = rows * np.int64(n_cols) + cols
    order = np.argsort(key, kind="stable")
    key = key[order]
    vals = vals[order]
    uniq, start = np.unique(key, return_index=True)
    summed = np.add.reduceat(vals, start) if len(vals) else vals
    r = uniq // n_cols
    c = (uniq % n_cols).astype(np.int32)
    rowptr = np.zeros(n_rows + 1, dtype=np.int64)
    np.add.at(rowptr, r + 1, 1)
    rowptr = np.cumsum(rowptr)
    return CSR(n_rows, n_cols, rowptr, c, summed)


# --------------------------------------------------------------------------- inputs


def read_lammps(path: str):
    """Token-counting reader of src/step-50.cc:181-258: token #2 is the atom count, token
    #35 starts the records ``id mol type q x y z``."""
    with open(path) as fh:
        tok = fh.read().split()
    n = int(tok[2])
    rec = tok[35 : 35 + 7 * n]
    a = np.array(rec, dtype=np.float64).reshape(n, 7)
    charges = a[:, 3].copy()
    pos = a[:, 4:7].copy()
    return charges, pos


def parse_prm(text: str) -> dict:
    """Flat ``{key: value}`` view of a deal.II .prm file (keys listed in src/step-50.cc:13-95)."""
    out = {}
    for line in text.splitlines():
        line = line.split("#", 1)[0].strip()
        if line.startswith("set "):
            k, v = line[4:].split("=", 1)
            out[k.strip()] = v.strip()
    return out


# --------------------------------------------------------------------------- Q1 element


def gauss01(n: int):
    x, w = np.polynomial.legendre.leggauss(n)
    return 0.5 * (x + 1.0), 0.5 * w


def tensor_quadrature(dim: int, n: int):
    """QGauss<dim>(n) on the unit cell, x fastest (deal.II tensor-product order)."""
    x, w = gauss01(n)
    if dim == 2:
        pts = np.array([[x[i], x[j]] for j in range(n) for i in range(n)])
        wts = np.array([w[i] * w[j] for j in range(n) for i in range(n)])
    else:
        pts = np.array([[x[i], x[j], x[k]] for k in range(n) for j in range(n) for i in range(n)])
        wts = np.array([w[i] * w[j] * w[k] for k in range(n) for j in range(n) for i in range(n)])
    return pts, wts


def q1_shapes(dim: int, pts: np.ndarray):
    """Shape values [q, i] and unit-cell gradients [q, i, d]; vertex i has bit d set if it
    sits at coordinate 1 in direction d (x fastest)."""
    nv = 1 << dim
    nq = len(pts)
    val = np.ones((nq, nv))
    grad = np.ones((nq, nv, dim))
    for i in range(nv):
        for d in range(dim):
            b = (i >> d) & 1
            f = pts[:, d] if b else 1.0 - pts[:, d]
            df = 1.0 if b else -1.0
            val[:, i] *= f
            for e in range(dim):
                grad[:, i, e] *= df if e == d else f
    return val, grad


@dataclass
class Lattice:
    """Uniform lattice of ``n`` cells per direction, vertex DoFs numbered x fastest."""

    dim: int
    n: int
    origin: float
    h: float

    @property
    def nv(self) -> int:
        return self.n + 1

    @property
    def n_dofs(self) -> int:
        return self.nv ** self.dim

    @property
    def n_cells(self) -> int:
        return self.n ** self.dim

    def vertex_coords(self) -> np.ndarray:
        ax = self.origin + self.h * np.arange(self.nv)
        if self.dim == 2:
            Y, X = np.meshgrid(ax, ax, indexing="ij")
            return np.stack([X.ravel(), Y.ravel()], axis=1)
        Z, Y, X = np.meshgrid(ax, ax, ax, indexing="ij")
        return np.stack([X.ravel(), Y.ravel(), Z.ravel()], axis=1)

    def boundary_mask(self) -> np.ndarray:
        idx = np.arange(self.nv)
        edge = (idx == 0) | (idx == self.n)
        if self.dim == 2:
            m = edge[:, None] | edge[None, :]
        else:
            m = edge[:, None, None] | edge[None, :, None] | edge[None, None, :]
        return m.ravel()

    def cell_dofs(self) -> np.ndarray:
        """[cell, 2^dim] vertex DoFs, cells x fastest."""
        c = np.arange(self.n)
        nv = self.nv
        if self.dim == 2:
            J, I = np.meshgrid(c, c, indexing="ij")
            base = (I + nv * J).ravel()
            offs = np.array([(i & 1) + nv * ((i >> 1) & 1) for i in range(4)])
        else:
            K, J, I = np.meshgrid(c, c, c, indexing="ij")
            base = (I + nv * (J + nv * K)).ravel()
            offs = np.array([(i & 1) + nv * (((i >> 1) & 1) + nv * ((i >> 2) & 1)) for i in range(8)])
        return base[:, None] + offs[None, :]

    def cell_origins(self) -> np.ndarray:
        c = self.origin + self.h * np.arange(self.n)
        if self.dim == 2:
            J, I = np.meshgrid(c, c, indexing="ij")
            return np.stack([I.ravel(), J.ravel()], axis=1)
        K, J, I = np.meshgrid(c, c, c, indexing="ij")
        return np.stack([I.ravel(), J.ravel(), K.ravel()], axis=1)


# --------------------------------------------------------------------------- problem functions


def step16_coefficient(p: np.ndarray) -> np.ndarray:  # include/step_50.h:246-254
    return np.where(np.sum(p * p, axis=-1) < 0.25, 5.0, 1.0)


def gaussian_rhs_no_lammps(p: np.ndarray, r_c: float) -> np.ndarray:  # include/step_50.h:321-329
    c = np.sum(p * p, axis=-1) / (r_c * r_c)
    return (8.0 * np.exp(-4.0 * c) - np.exp(-c)) / (r_c ** 3 * math.pi ** 1.5)


def exact_potential(p: np.ndarray, charges, pos, r_c: float) -> np.ndarray:  # include/step_50.h:338-353
    from math import erf

    out = np.zeros(len(p))
    verf = np.vectorize(erf)
    for q, x in zip(charges, pos):
        r = np.sqrt(np.sum((p - x) ** 2, axis=1))
        small = r < 1e-10
        rs = np.where(small, 1.0, r)
        out += np.where(small, q * 2.0 / (math.sqrt(math.pi) * r_c), q * verf(rs / r_c) / rs)
    return out


def dipole_bc(p: np.ndarray, dipole: np.ndarray) -> np.ndarray:
    """NonZeroDBC with x0 = 0 and the quadrupole forced to zero (src/step-50.cc:623-624,
    include/step_50.h:378-385)."""
    r = np.sqrt(np.sum(p * p, axis=1))
    return (p @ dipole) / r ** 3


# --------------------------------------------------------------------------- assembly


def cell_matrices(lat: Lattice, coefficient=None) -> np.ndarray:
    """K_e[c,i,j] = sum_q c(x_q) grad phi_i . grad phi_j JxW with QGauss(2)
    (src/step-50.cc:782-790, :878-884; quadrature_formula_laplace = degree+1, :145)."""
    pts, wts = tensor_quadrature(lat.dim, 2)
    _, grad = q1_shapes(lat.dim, pts)
    B = np.einsum("qid,qjd,q->qij", grad, grad, wts) * lat.h ** (lat.dim - 2)
    if coefficient is None:
        Ke = B.sum(axis=0)
        return np.broadcast_to(Ke, (lat.n_cells,) + Ke.shape)
    xq = lat.cell_origins()[:, None, :] + lat.h * pts[None, :, :]
    cq = coefficient(xq.reshape(-1, lat.dim)).reshape(lat.n_cells, len(pts))
    return np.einsum("cq,qij->cij", cq, B)


def assemble_constrained(lat: Lattice, Ke: np.ndarray, constrained: np.ndarray,
                         values: np.ndarray | None = None, cell_rhs: np.ndarray | None = None):
    """deal.II ``distribute_local_to_global`` with Dirichlet constraints only:
    constrained row/col -> diagonal |K_e(c,c)|, pattern keeps the eliminated positions as
    stored zeros, rhs_i = F_i - sum_c K_ic g_c, rhs_c = 0 (src/step-50.cc:793-795, 825-828)."""
    dofs = lat.cell_dofs()
    nv = dofs.shape[1]
    ci = constrained[dofs]  # [cell, i]
    rows = np.repeat(dofs[:, :, None], nv, axis=2)
    cols = np.repeat(dofs[:, None, :], nv, axis=1)
    free = (~ci)[:, :, None] & (~ci)[:, None, :]
    vals = np.where(free, Ke, 0.0)
    eye = np.eye(nv, dtype=bool)[None, :, :]
    diag_c = ci[:, :, None] & eye
    vals = np.where(diag_c, np.abs(Ke), vals)
    A = coo_to_csr(rows.ravel(), cols.ravel(), vals.ravel(), lat.n_dofs, lat.n_dofs)
    b = None
    if cell_rhs is not None:
        b = np.zeros(lat.n_dofs)
        F = np.where(ci, 0.0, cell_rhs)
        if values is not None:
            g = np.where(ci, values[dofs], 0.0)  # boundary values on constrained local dofs
            corr = np.einsum("cij,cj->ci", Ke, g)
            F = F - np.where(ci, 0.0, corr)
        np.add.at(b, dofs.ravel(), F.ravel())
    return A, b


def atom_cell_mask(lat: Lattice, pos: np.ndarray, cutoff: float) -> np.ndarray:
    """[cell, atom] True where ANY cell vertex is closer than ``cutoff`` to the atom
    (src/step-50.cc:273-284)."""
    vc = lat.vertex_coords()[lat.cell_dofs()]  # [cell, v, d]
    mask = np.zeros((lat.n_cells, len(pos)), dtype=bool)
    for k, x in enumerate(pos):
        d = np.sqrt(np.sum((vc - x) ** 2, axis=2))
        mask[:, k] = np.any(d < cutoff, axis=1)
    return mask


def gaussian_cell_rhs(lat: Lattice, charges, pos, r_c: float, n_q: int, mask=None) -> np.ndarray:
    """F_e[c,i] = sum_q phi_i(x_q) dens(x_q) JxW, dens incl. the 4 pi factor
    (src/step-50.cc:522, 544-570, 813-820)."""
    pts, wts = tensor_quadrature(lat.dim, n_q)
    val, _ = q1_shapes(lat.dim, pts)
    xq = lat.cell_origins()[:, None, :] + lat.h * pts[None, :, :]  # [c, q, d]
    const = 4.0 * math.pi / (r_c ** 3 * math.pi ** 1.5)
    dens = np.zeros(xq.shape[:2])
    for k, (q, x) in enumerate(zip(charges, pos)):
        r2 = np.sum((xq - x) ** 2, axis=2)
        term = const * np.exp(-r2 / (r_c * r_c)) * q
        if mask is not None:
            term = term * mask[:, k][:, None]
        dens += term
    return np.einsum("cq,qi,q->ci", dens, val, wts) * lat.h ** lat.dim


def function_cell_rhs(lat: Lattice, func, n_q: int) -> np.ndarray:
    """rhs_func->value_list branch (src/step-50.cc:799-803)."""
    pts, wts = tensor_quadrature(lat.dim, n_q)
    val, _ = q1_shapes(lat.dim, pts)
    xq = lat.cell_origins()[:, None, :] + lat.h * pts[None, :, :]
    f = func(xq.reshape(-1, lat.dim)).reshape(lat.n_cells, len(pts))
    return np.einsum("cq,qi,q->ci", f, val, wts) * lat.h ** lat.dim


def prolongation(coarse: Lattice, fine: Lattice) -> CSR:
    """MGTransferPrebuilt matrix N_fine x N_coarse for a uniformly refined lattice: the
    (bi/tri)linear embedding with the columns of coarse *boundary* DoFs zeroed
    (deal.II mg_transfer_prebuilt; reference call site src/step-50.cc:957-958)."""
    assert fine.n == 2 * coarse.n
    nvf, nvc, dim = fine.nv, coarse.nv, fine.dim
    # 1D embedding
    r1, c1, v1 = [], [], []
    for i in range(nvf):
        if i % 2 == 0:
            r1.append(i); c1.append(i // 2); v1.append(1.0)
        else:
            r1 += [i, i]; c1 += [i // 2, i // 2 + 1]; v1 += [0.5, 0.5]
    r1, c1, v1 = map(np.array, (r1, c1, v1))
    rows, cols, vals = r1, c1, v1
    for d in range(1, dim):
        rows = (rows[None, :] + (nvf ** d) * r1[:, None]).ravel()
        cols = (cols[None, :] + (nvc ** d) * c1[:, None]).ravel()
        vals = (vals[None, :] * v1[:, None]).ravel()
    bc = coarse.boundary_mask()
    vals = np.where(bc[cols], 0.0, vals)
    return coo_to_csr(rows, cols, vals, fine.n_dofs, coarse.n_dofs)


# --------------------------------------------------------------------------- whole problems


@dataclass
class Hierarchy:
    """Everything ``LaplaceProblem::solve`` consumes (src/step-50.cc:938-1017)."""

    system_matrix: CSR
    system_rhs: np.ndarray
    level_matrices: list  # A_l, l = 0..L
    edge_matrices: list  # I_l or None
    prolongations: list  # P_l: level l -> l+1, l = 0..L-1
    copy_global: list  # per level int32 arrays
    copy_level: list
    constrained: np.ndarray
    boundary_values: np.ndarray
    lattice: Lattice
    info: dict = field(default_factory=dict)


def gaussian_lattice(left: float, right: float, h: float, vacuum: int, dim: int = 3) -> Lattice:
    """src/step-50.cc:1504-1526."""
    a = 2.0 * h
    reps = int(2 * ((right - left) / a + 2 * vacuum))
    return Lattice(dim, reps, left - vacuum * a, h)


def build_gaussian_cycle0(charges, pos, *, left, right, h, vacuum, r_c, cutoff_param, n_q_rhs,
                          bc: str, rhs_optimization: bool = True) -> Hierarchy:
    """Cycle 0 of Problem=GaussianCharges with a LAMMPS file: one level, V-cycle == coarse CG."""
    lat = gaussian_lattice(left, right, h, vacuum, 3)
    Ke = cell_matrices(lat)
    mask = atom_cell_mask(lat, pos, cutoff_param * r_c) if rhs_optimization else None
    Fe = gaussian_cell_rhs(lat, charges, pos, r_c, 1 + n_q_rhs, mask)
    cons = lat.boundary_mask()
    xv = lat.vertex_coords()
    g = np.zeros(lat.n_dofs)
    if bc == "Exact":
        g[cons] = exact_potential(xv[cons], charges, pos, r_c)
    elif bc == "Inhomogeneous":
        dip = (charges[:, None] * pos).sum(axis=0)
        g[cons] = dipole_bc(xv[cons], dip)
    A, b = assemble_constrained(lat, Ke, cons, g, Fe)
    # level matrix 0: same cell matrices, boundary rows -> diagonal (src/step-50.cc:853-889)
    A0, _ = assemble_constrained(lat, Ke, cons)
    ident = np.arange(lat.n_dofs, dtype=np.int32)
    return Hierarchy(A, b, [A0], [None], [], [ident], [ident], cons, g, lat,
                     {"n_cells": lat.n_cells, "dofs_by_level": [lat.n_dofs]})


def build_uniform_hierarchy(dim: int, left: float, right: float, n_refine: int, *, problem: str,
                            r_c: float = 0.5, charges=None, pos=None, n_q_rhs: int = 1,
                            cutoff_param: float = 3.0, rhs_optimization: bool = False) -> Hierarchy:
    """Older-revision goldens: hyper_cube(left,right) + refine_global(n) => levels with 2^l
    cells per direction, homogeneous Dirichlet, per-level assembly."""
    lats = [Lattice(dim, 1 << l, left, (right - left) / (1 << l)) for l in range(n_refine + 1)]
    coeff = step16_coefficient if problem == "Step16" else None
    As = []
    for lat in lats:
        Al, _ = assemble_constrained(lat, cell_matrices(lat, coeff), lat.boundary_mask())
        As.append(Al)
    fine = lats[-1]
    if problem == "Step16":
        Fe = function_cell_rhs(fine, lambda p: np.full(len(p), 10.0), 1 + n_q_rhs)
    elif charges is None:
        Fe = function_cell_rhs(fine, lambda p: gaussian_rhs_no_lammps(p, r_c), 1 + n_q_rhs)
    else:
        mask = atom_cell_mask(fine, pos, cutoff_param * r_c) if rhs_optimization else None
        Fe = gaussian_cell_rhs(fine, charges, pos, r_c, 1 + n_q_rhs, mask)
    cons = fine.boundary_mask()
    A, b = assemble_constrained(fine, cell_matrices(fine, coeff), cons, None, Fe)
    Ps = [prolongation(lats[l], lats[l + 1]) for l in range(n_refine)]
    empty = np.zeros(0, dtype=np.int32)
    ident = np.arange(fine.n_dofs, dtype=np.int32)
    cg = [empty] * n_refine + [ident]
    return Hierarchy(A, b, As, [None] * (n_refine + 1), Ps, cg, list(cg), cons,
                     np.zeros(fine.n_dofs), fine,
                     {"n_cells": fine.n_cells, "dofs_by_level": [l.n_dofs for l in lats]})


# --------------------------------------------------------------------------- energy (src/step-50.cc:1310-1420)


def interpolate_q1(lat: Lattice, u: np.ndarray, p: np.ndarray) -> np.ndarray:
    """phi_h at points on a uniform lattice (find_active_cell_around_point + FEValues)."""
    out = np.zeros(len(p))
    nv = lat.nv
    for a, x in enumerate(p):
        s = (x - lat.origin) / lat.h
        c = np.clip(np.floor(s + 1e-12).astype(int), 0, lat.n - 1)
        t = s - c
        acc = 0.0
        for i in range(1 << lat.dim):
            w = 1.0
            idx = 0
            for d in range(lat.dim):
                bit = (i >> d) & 1
                w *= t[d] if bit else 1.0 - t[d]
                idx += (c[d] + bit) * nv ** d
            acc += w * u[idx]
        out[a] = acc
    return out


def electrostatic_energy(lat: Lattice, solution_distributed: np.ndarray, charges, pos, r_c: float) -> dict:
    from math import erfc

    n = len(charges)
    analytic = short = 0.0
    for i in range(n):
        for j in range(i + 1, n):
            r = float(np.linalg.norm(pos[i] - pos[j]))
            analytic += charges[i] * charges[j] / r
            short += charges[i] * charges[j] * erfc(r / r_c) / r
    fe = float(np.sum(0.5 * charges * interpolate_q1(lat, solution_distributed, pos)))
    self_e = float(np.sum(charges * charges) / (math.sqrt(math.pi) * r_c))
    total = short + fe - self_e
    return {"analytical": analytic, "short": short, "fe_long": fe, "self": self_e, "total": total,
            "abs_error": abs(abs(analytic) - abs(total))}
