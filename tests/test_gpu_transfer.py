"""MGTransferPrebuilt::build_matrices on the device (csrc/gmg_transfer.hpp, gmg_build_transfer; reference:
src/step-50.cc:957-958) against the host-built operators of the same adaptive hierarchies: the CSR arrays of P_l and of its
transpose must be IDENTICAL (same pattern, same column order, same values), on 2D and 3D hierarchies with hanging nodes
(adaptive cycles of the reference's regression problem) and on a uniformly refined one."""
import os

import numpy as np
import pytest

from gpu_util import capi, pkg
from oracle import gmg_oracle as go

pytestmark = pytest.mark.gpu


def _transpose_stable(P):
    """row j of P^T lists the entries of column j in ascending source row (what gmg_set_prolongation derives on the host)"""
    rows = np.repeat(np.arange(P.n_rows), np.diff(P.rowptr))
    order = np.lexsort((rows, P.col))
    rp = np.zeros(P.n_cols + 1, dtype=np.int64)
    np.add.at(rp, P.col + 1, 1)
    return np.cumsum(rp), rows[order].astype(np.int32), P.val[order]


def _check_levels(p, ctx_levels):
    S = pkg().step50
    L = p.n_levels()
    for l in range(L - 1):
        P = p.matrix("prolongation", l)  # host-built on demand (cell by cell, csr_from_triplets)
        dev = ctx_levels.get_transfer(l, False)
        assert (dev.n_rows, dev.n_cols, dev.nnz) == (P.n_rows, P.n_cols, P.nnz), (l, dev.n_rows, dev.n_cols, dev.nnz, P.n_rows, P.n_cols, P.nnz)
        assert np.array_equal(dev.rowptr, P.rowptr) and np.array_equal(dev.col, P.col) and np.array_equal(dev.val, P.val)
        trp, tcol, tval = _transpose_stable(P)
        devt = ctx_levels.get_transfer(l, True)
        assert np.array_equal(devt.rowptr, trp) and np.array_equal(devt.col, tcol) and np.array_equal(devt.val, tval)


@pytest.mark.parametrize("dim,cycles,refine", [(3, 4, 0), (2, 3, 2), (3, 1, 2)])
def test_device_built_transfers_equal_the_host_built_ones(golden_dir, dim, cycles, refine):
    S = pkg().step50
    if dim == 3 and refine == 0:
        # the reference's regression problem (tests/gaussian-charges.prm): adaptive cycles with hanging nodes, 45^3 level 0
        p = S.Problem(S.prm_text(left=0, right=1, mesh_size=0.25, vacuum=10, problem="GaussianCharges", dim=3, bc="Exact", cycles=cycles,
                                 r_c=0.5, cutoff=3.5, rhs_optimization=True, quad_rhs=4, global_refinement=0, smoother="SSOR"))
        p.read_lammps(os.path.join(golden_dir, "atom_n1_2.data"))
    else:
        p = S.Problem(S.prm_text(left=0, right=1, problem="Step16", dim=dim, bc="Homogeneous", cycles=cycles, global_refinement=refine, smoother="Jacobi"))
    for c in range(cycles):
        rep = p.run_cycle(c, on_device=True)
        assert (rep["build_matrices_ms"] > 0.0) == (p.n_levels() > 1)  # the device built them (cycle 0 of an adaptive run has one level)
        ctx = capi().Context.view(p.gmg_context())
        _check_levels(p, ctx)
    assert p.n_levels() >= 2
    p.close()


def test_host_built_and_device_built_hierarchies_solve_alike(golden_dir):
    """Same adaptive cycles with both switches off (operators assembled on the host and handed over as CSR) and on: identical
    iteration counts and printed values."""
    S = pkg().step50
    reps = {}
    for on in (False, True):
        p = S.Problem(S.prm_text(left=0, right=1, mesh_size=0.25, vacuum=10, problem="GaussianCharges", dim=3, bc="Exact", cycles=3, r_c=0.5,
                                 cutoff=3.5, rhs_optimization=True, quad_rhs=4, global_refinement=0, smoother="SSOR",
                                 level0_on_device=on, transfer_on_device=on))
        p.read_lammps(os.path.join(golden_dir, "atom_n1_2.data"))
        reps[on] = [p.run_cycle(c, on_device=True) for c in range(3)]
        p.close()
    for a, b in zip(reps[False], reps[True]):
        assert a["cg_iterations"] == b["cg_iterations"] and a["dofs_by_level"] == b["dofs_by_level"]
        assert a["build_matrices_ms"] == 0.0 and (b["build_matrices_ms"] > 0.0) == (len(b["dofs_by_level"]) > 1)
        for k in ("starting_value", "sol_l1", "sol_l2", "sol_linf", "refine_threshold"):
            assert abs(a[k] - b[k]) <= 1e-12 * abs(a[k]), (k, a[k], b[k])
