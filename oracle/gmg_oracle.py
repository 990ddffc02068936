"""ctypes wrapper of the CPU oracle ``gmg_oracle.c`` (TEST INFRASTRUCTURE ONLY).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module.  See the header of gmg_oracle.c for the reference lines it restates.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libgmg_oracle.so")

JACOBI, SSOR, CHEBYSHEV = 0, 1, 2
PRECOND_GMG, PRECOND_JACOBI, PRECOND_IDENTITY = 0, 1, 2
OK, ERR_OUTER_NOCONV, ERR_COARSE_NOCONV = 0, 2, 3


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "gmg_oracle.c")
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "libgmg_oracle.so"])
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB)
        _lib.oracle_mg_create.restype = C.c_void_p
        _lib.oracle_cheb_lmax.restype = C.c_double
    return _lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def _csr_args(m):
    rp = np.ascontiguousarray(m.rowptr, dtype=np.int64)
    col = np.ascontiguousarray(m.col, dtype=np.int32)
    val = np.ascontiguousarray(m.val, dtype=np.float64)
    return rp, col, val


def spmv(m, x):
    rp, col, val = _csr_args(m)
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.zeros(m.n_rows)
    lib().oracle_spmv(C.c_int64(m.n_rows), _p(rp, C.c_int64), _p(col, C.c_int32), _p(val, C.c_double),
                      _p(x, C.c_double), _p(y, C.c_double))
    return y


def spmv_transpose(m, x, y0=None):
    rp, col, val = _csr_args(m)
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.zeros(m.n_cols) if y0 is None else np.array(y0, dtype=np.float64)
    lib().oracle_spmv_transpose(C.c_int64(m.n_rows), C.c_int64(m.n_cols), _p(rp, C.c_int64), _p(col, C.c_int32),
                                _p(val, C.c_double), _p(x, C.c_double), _p(y, C.c_double),
                                C.c_int(0 if y0 is None else 1))
    return y


def set_threads(n: int):
    lib().oracle_set_threads(C.c_int(n))


def max_threads() -> int:
    return int(lib().oracle_max_threads())


class OracleMG:
    """The object graph of ``LaplaceProblem::solve`` (src/step-50.cc:954-992) on the CPU."""

    def __init__(self, hier, smoother=SSOR, omega=0.5, steps=2, cheb_degree=2, cheb_ratio=30.0, cheb_lmax=0.0,
                 coarse_tol=1e-10, coarse_maxit=1000, ssor_blocks=1):
        L = lib()
        self.n_levels = len(hier.level_matrices)
        self.h = C.c_void_p(L.oracle_mg_create(C.c_int(self.n_levels)))
        self.n = hier.system_matrix.n_rows
        self._keep = []
        rp, col, val = _csr_args(hier.system_matrix)
        L.oracle_mg_set_system_matrix(self.h, C.c_int64(self.n), _p(rp, C.c_int64), _p(col, C.c_int32), _p(val, C.c_double))
        for l, A in enumerate(hier.level_matrices):
            rp, col, val = _csr_args(A)
            L.oracle_mg_set_level_matrix(self.h, C.c_int(l), C.c_int64(A.n_rows), _p(rp, C.c_int64), _p(col, C.c_int32),
                                         _p(val, C.c_double))
            I = hier.edge_matrices[l]
            if I is not None and I.nnz > 0:
                rp, col, val = _csr_args(I)
                L.oracle_mg_set_edge_matrix(self.h, C.c_int(l), C.c_int64(I.n_rows), _p(rp, C.c_int64), _p(col, C.c_int32),
                                            _p(val, C.c_double))
            g = np.ascontiguousarray(hier.copy_global[l], dtype=np.int32)
            v = np.ascontiguousarray(hier.copy_level[l], dtype=np.int32)
            L.oracle_mg_set_copy_indices(self.h, C.c_int(l), C.c_int64(len(g)), _p(g, C.c_int32), _p(v, C.c_int32))
        for l, P in enumerate(hier.prolongations):
            rp, col, val = _csr_args(P)
            L.oracle_mg_set_prolongation(self.h, C.c_int(l), C.c_int64(P.n_rows), C.c_int64(P.n_cols), _p(rp, C.c_int64),
                                         _p(col, C.c_int32), _p(val, C.c_double))
        L.oracle_mg_set_smoother(self.h, C.c_int(smoother), C.c_double(omega), C.c_int(steps), C.c_int(cheb_degree),
                                 C.c_double(cheb_ratio), C.c_double(cheb_lmax))
        L.oracle_mg_set_coarse(self.h, C.c_double(coarse_tol), C.c_int(coarse_maxit))
        L.oracle_mg_set_ssor_blocks(self.h, C.c_int(ssor_blocks))

    def __del__(self):
        try:
            lib().oracle_mg_destroy(self.h)
        except Exception:
            pass

    def vcycle(self, src):
        src = np.ascontiguousarray(src, dtype=np.float64)
        dst = np.zeros(self.n)
        rc = lib().oracle_vcycle(self.h, _p(dst, C.c_double), _p(src, C.c_double))
        return dst, rc

    def solve(self, b, x0=None, rel_tol=1e-8, max_it=500, precond=PRECOND_GMG):
        b = np.ascontiguousarray(b, dtype=np.float64)
        x = np.zeros(self.n) if x0 is None else np.array(x0, dtype=np.float64)
        it = C.c_int(0)
        r0, r = C.c_double(0), C.c_double(0)
        ci = C.c_int64(0)
        rc = lib().oracle_solve(self.h, _p(x, C.c_double), _p(b, C.c_double), C.c_double(rel_tol), C.c_int(max_it),
                                C.c_int(precond), C.byref(it), C.byref(r0), C.byref(r), C.byref(ci))
        return {"x": x, "iterations": it.value, "starting_value": r0.value, "convergence_value": r.value,
                "coarse_iterations": ci.value, "status": rc}

    def coarse_solve(self, b):
        b = np.ascontiguousarray(b, dtype=np.float64)
        x = np.zeros(len(b))
        it = C.c_int(0)
        r = C.c_double(0)
        rc = lib().oracle_coarse_solve(self.h, _p(x, C.c_double), _p(b, C.c_double), C.byref(it), C.byref(r))
        return x, it.value, r.value, rc

    def smooth(self, level, u, rhs, from_zero):
        u = np.array(u, dtype=np.float64)
        rhs = np.ascontiguousarray(rhs, dtype=np.float64)
        lib().oracle_smooth(self.h, C.c_int(level), _p(u, C.c_double), _p(rhs, C.c_double), C.c_int(1 if from_zero else 0))
        return u

    def cheb_lmax(self, level):
        return float(lib().oracle_cheb_lmax(self.h, C.c_int(level)))
