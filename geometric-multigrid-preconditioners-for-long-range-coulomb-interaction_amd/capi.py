"""ctypes binding of the C-ABI in include/gmg_coulomb.h (libgmgcoulomb.so).

This is plumbing for tests and bench.py: it adds nothing to the ABI.  There is NO CPU
fallback: if the HIP library is missing, or there is no GPU, the calls fail loudly.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import build as _build

OK, ERR_INVALID, ERR_OUTER_NOCONV, ERR_COARSE_NOCONV, ERR_HIP, ERR_COMM, ERR_UNSUPPORTED = range(7)
SYSTEM = -1
JACOBI, SSOR, CHEBYSHEV = 0, 1, 2
PRECOND_GMG, PRECOND_JACOBI, PRECOND_IDENTITY = 0, 1, 2
UNIQUE_ID_BYTES = 128

# every symbol include/gmg_coulomb.h declares (checked by tests/test_abi.py)
SYMBOLS = [
    "gmg_create", "gmg_destroy", "gmg_reset", "gmg_last_error", "gmg_synchronize",
    "gmg_set_system_matrix", "gmg_set_level_matrix", "gmg_set_level_matrix_lattice", "gmg_set_edge_matrix", "gmg_set_prolongation", "gmg_build_transfer", "gmg_get_transfer",
    "gmg_set_copy_indices", "gmg_set_smoother", "gmg_set_coarse",
    "gmg_vec_alloc", "gmg_vec_free", "gmg_vec_upload", "gmg_vec_download", "gmg_vec_set_zero", "gmg_vec_equ",
    "gmg_vec_add", "gmg_vec_sadd", "gmg_vec_dot", "gmg_vec_norms", "gmg_vec_all_zero",
    "gmg_spmv", "gmg_precondition", "gmg_precondition_jacobi", "gmg_coarse_solve", "gmg_smoother_step",
    "gmg_prolongate", "gmg_restrict_and_add", "gmg_cg_solve",
    "gmg_comm_unique_id", "gmg_comm_init", "gmg_comm_barrier", "gmg_comm_info", "gmg_set_halo_plan", "gmg_set_global_sizes", "gmg_partition_range",
    "gmg_vec_allgather",
    "gmg_stats_reset", "gmg_stats_get", "gmg_set_profiling", "gmg_set_tuning", "gmg_set_option", "gmg_set_ssor_blocks", "gmg_calibrate_hbm", "gmg_charge_density", "gmg_get_charge_density", "gmg_rhs_assemble",
]


class Stats(C.Structure):
    _fields_ = [("coarse_solves", C.c_int64), ("coarse_iterations", C.c_int64), ("vcycles", C.c_int64),
                ("spmv0_samples", C.c_int64), ("spmv0_ms_total", C.c_double), ("spmv0_rows", C.c_int64),
                ("spmv0_nnz", C.c_int64), ("cgupd_samples", C.c_int64), ("cgupd_ms_total", C.c_double),
                ("coarse_variant", C.c_int64), ("spmv0_layout", C.c_int64), ("spmv0_matrix_bytes", C.c_int64),
                ("spmv0_pattern_slices", C.c_int64), ("spmv0_slices", C.c_int64), ("coarse_enqueued", C.c_int64),
                ("spmv0_noop_samples", C.c_int64), ("spmv0_noop_ms_total", C.c_double),
                ("sgs_samples", C.c_int64), ("sgs_ms_total", C.c_double), ("sgs_substeps", C.c_int64), ("sgs_stream_bytes", C.c_int64),
                ("sgs_launches", C.c_int64), ("build_matrices_ms", C.c_double)]


class GMGError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"gmg error {code}: {msg}")
        self.code = code


_lib = None


def load(build_if_missing: bool = False):
    """dlopen libgmgcoulomb.so from the source tree.  Raises if it has not been built."""
    global _lib
    if _lib is None:
        # (GMG_DEVICE_LIB: measurement scripts only -- tools/build_experiments.sh; the host-side C++ always links the shipped library)
        path = os.environ.get("GMG_DEVICE_LIB") or _build.LIB_DEVICE
        if build_if_missing:
            _build.build_device()
        if not os.path.exists(path):
            raise FileNotFoundError(f"{path} missing: run __graft_entry__.build() (hipcc --offload-arch=gfx950)")
        _lib = C.CDLL(path)
        _lib.gmg_last_error.restype = C.c_char_p
        _lib.gmg_last_error.argtypes = [C.c_void_p]
    return _lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def _csr(m):
    return (np.ascontiguousarray(m.rowptr, dtype=np.int64), np.ascontiguousarray(m.col, dtype=np.int32),
            np.ascontiguousarray(m.val, dtype=np.float64))


class DeviceVector:
    def __init__(self, ctx: "Context", n: int):
        self.ctx, self.n = ctx, int(n)
        self.ptr = C.POINTER(C.c_double)()
        ctx._chk(ctx.L.gmg_vec_alloc(ctx.h, C.c_int64(self.n), C.byref(self.ptr)))

    def upload(self, a):
        a = np.ascontiguousarray(a, dtype=np.float64)
        assert a.size == self.n
        self.ctx._chk(self.ctx.L.gmg_vec_upload(self.ctx.h, self.ptr, _p(a, C.c_double), C.c_int64(self.n)))
        return self

    def download(self):
        out = np.empty(self.n)
        self.ctx._chk(self.ctx.L.gmg_vec_download(self.ctx.h, _p(out, C.c_double), self.ptr, C.c_int64(self.n)))
        return out

    def free(self):
        if self.ptr:
            self.ctx.L.gmg_vec_free(self.ctx.h, self.ptr)
            self.ptr = C.POINTER(C.c_double)()


class Context:
    """Thin object view of a gmg_context*."""

    def __init__(self, n_levels: int, device: int = 0):
        self.L = load()
        self.h = C.c_void_p()
        rc = self.L.gmg_create(C.byref(self.h), C.c_int(device), C.c_int(n_levels))
        if rc != OK:
            raise GMGError(rc, "gmg_create failed (no MI355X visible?)")
        self.n_levels = n_levels
        self.n_system = 0

    @classmethod
    def view(cls, handle):
        """Non-owning view of a gmg_context* created elsewhere (the host-side C++)."""
        self = cls.__new__(cls)
        self.L, self.h, self.owned = load(), handle, False
        self.n_levels = self.n_system = 0
        return self

    owned = True

    def close(self):
        if self.h and self.owned:
            self.L.gmg_destroy(self.h)
        self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != OK:
            raise GMGError(rc, self.L.gmg_last_error(self.h).decode())

    # ---- operators
    def set_system_matrix(self, m):
        rp, c, v = _csr(m)
        self._chk(self.L.gmg_set_system_matrix(self.h, C.c_int64(m.n_rows), C.c_int64(m.n_cols), _p(rp, C.c_int64),
                                               _p(c, C.c_int32), _p(v, C.c_double)))
        self.n_system = m.n_rows

    def set_level_matrix(self, level, m):
        rp, c, v = _csr(m)
        self._chk(self.L.gmg_set_level_matrix(self.h, C.c_int(level), C.c_int64(m.n_rows), C.c_int64(m.n_cols),
                                              _p(rp, C.c_int64), _p(c, C.c_int32), _p(v, C.c_double)))

    def set_level_matrix_lattice(self, level, nv, Ke):
        """level 0 of an undivided lattice formed on the device: nv vertices per direction, Ke the 8 x 8 cell matrix"""
        nv3 = (C.c_int32 * 3)(*[int(v) for v in nv])
        ke = np.ascontiguousarray(Ke, dtype=np.float64).reshape(64)
        self._chk(self.L.gmg_set_level_matrix_lattice(self.h, C.c_int(level), nv3, _p(ke, C.c_double)))

    def set_edge_matrix(self, level, m):
        rp, c, v = _csr(m)
        self._chk(self.L.gmg_set_edge_matrix(self.h, C.c_int(level), C.c_int64(m.n_rows), C.c_int64(m.n_cols),
                                             _p(rp, C.c_int64), _p(c, C.c_int32), _p(v, C.c_double)))

    def set_prolongation(self, level, m):
        rp, c, v = _csr(m)
        self._chk(self.L.gmg_set_prolongation(self.h, C.c_int(level), C.c_int64(m.n_rows), C.c_int64(m.n_cols),
                                              _p(rp, C.c_int64), _p(c, C.c_int32), _p(v, C.c_double)))

    def build_transfer(self, level, dim, coarse_vertex, coarse_boundary, fine_vertex, fine_spacing):
        """MGTransferPrebuilt::build_matrices on the device; returns the device time in ms"""
        cv = np.ascontiguousarray(coarse_vertex, dtype=np.uint64)
        cb = np.ascontiguousarray(coarse_boundary, dtype=np.uint8)
        fv = np.ascontiguousarray(fine_vertex, dtype=np.uint64)
        ms = C.c_double(0)
        self._chk(self.L.gmg_build_transfer(self.h, C.c_int(level), C.c_int(dim), C.c_int64(len(cv)), _p(cv, C.c_uint64), _p(cb, C.c_uint8),
                                            C.c_int64(len(fv)), _p(fv, C.c_uint64), C.c_uint64(int(fine_spacing)), C.byref(ms)))
        return ms.value

    def get_transfer(self, level, transposed=False):
        from types import SimpleNamespace
        nr, nc, nz = C.c_int64(0), C.c_int64(0), C.c_int64(0)
        self._chk(self.L.gmg_get_transfer(self.h, C.c_int(level), C.c_int(1 if transposed else 0), C.byref(nr), C.byref(nc), C.byref(nz), None, None, None))
        rp = np.zeros(nr.value + 1, dtype=np.int64)
        col, val = np.zeros(max(nz.value, 1), dtype=np.int32), np.zeros(max(nz.value, 1))
        self._chk(self.L.gmg_get_transfer(self.h, C.c_int(level), C.c_int(1 if transposed else 0), C.byref(nr), C.byref(nc), C.byref(nz),
                                          _p(rp, C.c_int64), _p(col, C.c_int32), _p(val, C.c_double)))
        return SimpleNamespace(n_rows=nr.value, n_cols=nc.value, nnz=nz.value, rowptr=rp, col=col[:nz.value], val=val[:nz.value])

    def set_copy_indices(self, level, global_idx, level_idx):
        g = np.ascontiguousarray(global_idx, dtype=np.int32)
        l = np.ascontiguousarray(level_idx, dtype=np.int32)
        self._chk(self.L.gmg_set_copy_indices(self.h, C.c_int(level), C.c_int64(len(g)), _p(g, C.c_int32), _p(l, C.c_int32)))

    def set_smoother(self, kind, omega=0.5, steps=2, cheb_degree=2, cheb_ratio=30.0, cheb_lmax=0.0):
        self._chk(self.L.gmg_set_smoother(self.h, C.c_int(kind), C.c_double(omega), C.c_int(steps), C.c_int(cheb_degree),
                                          C.c_double(cheb_ratio), C.c_double(cheb_lmax)))

    def set_coarse(self, abs_tol=1e-10, max_it=1000):
        self._chk(self.L.gmg_set_coarse(self.h, C.c_double(abs_tol), C.c_int(max_it)))

    def load_hierarchy(self, hier):
        """Upload everything LaplaceProblem::solve consumes (any object with the attribute
        names of the hierarchy the host side produces)."""
        self.set_system_matrix(hier.system_matrix)
        for l, A in enumerate(hier.level_matrices):
            self.set_level_matrix(l, A)
            I = hier.edge_matrices[l]
            if I is not None and I.nnz > 0:
                self.set_edge_matrix(l, I)
            self.set_copy_indices(l, hier.copy_global[l], hier.copy_level[l])
        for l, P in enumerate(hier.prolongations):
            self.set_prolongation(l, P)

    # ---- vectors
    def vector(self, n, data=None):
        v = DeviceVector(self, n)
        if data is not None:
            v.upload(data)
        return v

    def dot(self, x, y):
        out = C.c_double(0)
        self._chk(self.L.gmg_vec_dot(self.h, x.ptr, y.ptr, C.c_int64(x.n), C.byref(out)))
        return out.value

    def norms(self, x):
        a, b, c = C.c_double(0), C.c_double(0), C.c_double(0)
        self._chk(self.L.gmg_vec_norms(self.h, x.ptr, C.c_int64(x.n), C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def all_zero(self, x):
        out = C.c_int(0)
        self._chk(self.L.gmg_vec_all_zero(self.h, x.ptr, C.c_int64(x.n), C.byref(out)))
        return bool(out.value)

    def equ(self, y, a, x):
        self._chk(self.L.gmg_vec_equ(self.h, y.ptr, C.c_double(a), x.ptr, C.c_int64(y.n)))

    def add(self, y, a, x):
        self._chk(self.L.gmg_vec_add(self.h, y.ptr, C.c_double(a), x.ptr, C.c_int64(y.n)))

    def sadd(self, y, s, a, x):
        self._chk(self.L.gmg_vec_sadd(self.h, y.ptr, C.c_double(s), C.c_double(a), x.ptr, C.c_int64(y.n)))

    def set_zero(self, x):
        self._chk(self.L.gmg_vec_set_zero(self.h, x.ptr, C.c_int64(x.n)))

    # ---- concepts
    def spmv(self, which, dst, src):
        self._chk(self.L.gmg_spmv(self.h, C.c_int(which), dst.ptr, src.ptr))

    def precondition(self, dst, src):
        self._chk(self.L.gmg_precondition(self.h, dst.ptr, src.ptr))

    def precondition_jacobi(self, omega, dst, src):
        self._chk(self.L.gmg_precondition_jacobi(self.h, C.c_double(omega), dst.ptr, src.ptr))

    def coarse_solve(self, dst, src):
        it, res = C.c_int(0), C.c_double(0)
        rc = self.L.gmg_coarse_solve(self.h, dst.ptr, src.ptr, C.byref(it), C.byref(res))
        return it.value, res.value, rc

    def smoother_step(self, level, u, rhs, from_zero):
        self._chk(self.L.gmg_smoother_step(self.h, C.c_int(level), u.ptr, rhs.ptr, C.c_int(1 if from_zero else 0)))

    def prolongate(self, level, dst, src):
        self._chk(self.L.gmg_prolongate(self.h, C.c_int(level), dst.ptr, src.ptr))

    def restrict_and_add(self, level, dst, src):
        self._chk(self.L.gmg_restrict_and_add(self.h, C.c_int(level), dst.ptr, src.ptr))

    def cg_solve(self, x, b, rel_tol=1e-8, max_it=500, precond=PRECOND_GMG):
        it, r0, r = C.c_int(0), C.c_double(0), C.c_double(0)
        rc = self.L.gmg_cg_solve(self.h, x.ptr, b.ptr, C.c_double(rel_tol), C.c_int(max_it), C.c_int(precond),
                                 C.byref(it), C.byref(r0), C.byref(r))
        return {"iterations": it.value, "starting_value": r0.value, "convergence_value": r.value, "status": rc}

    def synchronize(self):
        self._chk(self.L.gmg_synchronize(self.h))

    # ---- distributed
    @staticmethod
    def unique_id() -> bytes:
        buf = C.create_string_buffer(UNIQUE_ID_BYTES)
        rc = load().gmg_comm_unique_id(buf)
        if rc != OK:
            raise GMGError(rc, "gmg_comm_unique_id failed")
        return buf.raw

    def comm_init(self, rank, n_ranks, uid: bytes):
        buf = C.create_string_buffer(uid, UNIQUE_ID_BYTES)
        self._chk(self.L.gmg_comm_init(self.h, C.c_int(rank), C.c_int(n_ranks), buf))

    def comm_info(self) -> dict:
        out = (C.c_int64 * 8)()
        self._chk(self.L.gmg_comm_info(self.h, out))
        return {"ranks": int(out[0]), "transport": {0: "none", 1: "rccl", 2: "peer"}[int(out[1])], "mailbox_finegrained": bool(out[2]),
                "ring_memory": {-1: "not allocated", 0: "hipMalloc (coarse-grained)", 1: "fine-grained"}[int(out[3])],
                "devices": int(out[4]), "level0_partitioned": bool(out[5])}

    def set_global_sizes(self, n_system, n_level0):
        self._chk(self.L.gmg_set_global_sizes(self.h, C.c_int64(n_system), C.c_int64(n_level0)))

    def allgather(self, n_global, dst_full, src_local):
        self._chk(self.L.gmg_vec_allgather(self.h, C.c_int64(n_global), dst_full.ptr, src_local.ptr))

    def set_halo_plan(self, which, neighbor_rank, send_count, send_idx, recv_count):
        nr = np.ascontiguousarray(neighbor_rank, dtype=np.int32)
        sc = np.ascontiguousarray(send_count, dtype=np.int32)
        si = np.ascontiguousarray(send_idx, dtype=np.int32)
        rc_ = np.ascontiguousarray(recv_count, dtype=np.int32)
        self._chk(self.L.gmg_set_halo_plan(self.h, C.c_int(which), C.c_int(len(nr)), _p(nr, C.c_int32), _p(sc, C.c_int32),
                                           _p(si, C.c_int32), _p(rc_, C.c_int32)))

    # ---- measurement
    def stats(self) -> Stats:
        s = Stats()
        self._chk(self.L.gmg_stats_get(self.h, C.byref(s)))
        return s

    def stats_reset(self):
        self._chk(self.L.gmg_stats_reset(self.h))

    def set_profiling(self, every):
        self._chk(self.L.gmg_set_profiling(self.h, C.c_int(every)))

    def calibrate_hbm(self, n_bytes=1 << 30, reps=10):
        r, c = C.c_double(0), C.c_double(0)
        self._chk(self.L.gmg_calibrate_hbm(self.h, C.c_int64(n_bytes), C.c_int(reps), C.byref(r), C.byref(c)))
        return r.value, c.value

    def set_tuning(self, coarse_chunk=0, cg_variant=0, ssor_blocks=0):
        self._chk(self.L.gmg_set_tuning(self.h, C.c_int(coarse_chunk), C.c_int(cg_variant)))
        if ssor_blocks:
            self._chk(self.L.gmg_set_ssor_blocks(self.h, C.c_int(ssor_blocks)))

    def set_option(self, key: str, value: float = 1.0):
        """Diagnostic / measurement options by name (include/gmg_coulomb.h: gmg_set_option)."""
        self._chk(self.L.gmg_set_option(self.h, key.encode(), C.c_double(value)))
