"""ctypes binding of the host-side C++ (csrc/host/libstep50host.so): the mirror of the
reference's LaplaceProblem<dim> and main.cc.  Drives mesh/assembly on the CPU and the solve
on the MI355X through the C-ABI.  No CPU fallback for the solve."""
from __future__ import annotations

import ctypes as C
import os
from types import SimpleNamespace

import numpy as np

from . import build as _build

_lib = None


class Report(C.Structure):
    _fields_ = [("cycle", C.c_int32), ("cg_iterations", C.c_int32), ("status", C.c_int32), ("has_energy", C.c_int32),
                ("n_levels", C.c_int32), ("pad", C.c_int32),
                ("active_cells", C.c_int64), ("dofs", C.c_int64), ("coarse_iterations", C.c_int64),
                ("dofs_by_level", C.c_int64 * 16),
                ("rhs_l1", C.c_double), ("rhs_l2", C.c_double), ("rhs_linf", C.c_double),
                ("matrix_l1", C.c_double), ("matrix_linf", C.c_double), ("matrix_frobenius", C.c_double),
                ("starting_value", C.c_double), ("convergence_value", C.c_double),
                ("sol_l1", C.c_double), ("sol_l2", C.c_double), ("sol_linf", C.c_double), ("refine_threshold", C.c_double),
                ("energy_analytical", C.c_double), ("energy_short", C.c_double), ("energy_fe_long", C.c_double),
                ("energy_self", C.c_double), ("energy_total", C.c_double), ("energy_abs_error", C.c_double),
                ("solve_seconds", C.c_double), ("energy_norm_error", C.c_double), ("build_matrices_ms", C.c_double)]

    def as_dict(self):
        d = {k: getattr(self, k) for k, _ in self._fields_ if k not in ("dofs_by_level", "pad")}
        d["dofs_by_level"] = [int(self.dofs_by_level[i]) for i in range(self.n_levels)]
        return d


def load(build_if_missing: bool = False):
    global _lib
    if _lib is None:
        if build_if_missing:
            _build.build_all()
        if not os.path.exists(_build.LIB_HOST):
            raise FileNotFoundError(f"{_build.LIB_HOST} missing: run __graft_entry__.build()")
        _lib = C.CDLL(_build.LIB_HOST)
        _lib.step50_create.restype = C.c_void_p
        _lib.step50_last_error.restype = C.c_char_p
        _lib.step50_log.restype = C.c_char_p
        _lib.step50_gmg_context.restype = C.c_void_p
        for f in ("step50_n_atoms", "step50_copy_indices_size", "step50_n_dofs"):
            getattr(_lib, f).restype = C.c_int64
        # host-side setup threads: a GPU box shares its cores between GPUs (16 per GPU)
        _lib.step50_set_threads(C.c_int(max(1, min(16, os.cpu_count() or 1))))
    return _lib


def set_threads(n: int):
    load().step50_set_threads(C.c_int(n))


def prm_text(**kw) -> str:
    """A .prm file body with the reference's keys (src/step-50.cc:13-95)."""
    keymap = {
        "global_refinement": ("Geometry", "Number of global refinement"),
        "left": ("Geometry", "Domain limit left"), "right": ("Geometry", "Domain limit right"),
        "mesh_size": ("Geometry", "Mesh size"), "vacuum": ("Geometry", "Vacuum repetitions"),
        "problem": ("Problem Selection", "Problem"), "dim": ("Problem Selection", "Dimension"),
        "bc": ("Problem Selection", "Boundary conditions selection"),
        "cycles": ("Misc", "Number of Adaptive Refinement"), "r_c": ("Misc", "smoothing length"),
        "cutoff": ("Misc", "Nonzero Density radius parameter around each charge"),
        "rhs_optimization": ("Misc", "Flag for RHS evaluation optimization"),
        "quad_rhs": ("Misc", "Quadrature points for RHS function"),
        "preconditioner": ("Solver input data", "Preconditioner"),
        "lammps": ("Lammps data", "Lammps input file"),
        "smoother": ("Solver input data", "Smoother"), "omega": ("Solver input data", "Smoother damping"),
        "steps": ("Solver input data", "Smoother steps"), "cheb_degree": ("Solver input data", "Chebyshev degree"),
        "device_cg": ("Solver input data", "Device resident outer CG"),
        "ssor_blocks": ("Solver input data", "SSOR blocks"),
        "densities_on_device": ("Misc", "Charge densities on device"),
        "partition_level0": ("Solver input data", "Partition level 0"),
        "refinement_estimator": ("Misc", "Refinement estimator"),
        "level0_numbering": ("Misc", "Level 0 numbering"),
        "level0_on_device": ("Misc", "Level 0 matrix on device"),
        "transfer_on_device": ("Misc", "Transfer matrices on device"),
        "rhs_on_device": ("Misc", "RHS on device"),
        "short_range_cutoff": ("Misc", "Short-range cutoff in smoothing lengths"),
        "energy_for_large_systems": ("Misc", "Energy for large systems"),
    }
    sections = {}
    for k, v in kw.items():
        sec, key = keymap[k]
        if isinstance(v, bool):
            v = "true" if v else "false"
        sections.setdefault(sec, []).append(f"  set {key} = {v}")
    out = ["set Polynomial degree = 1"]
    for sec, lines in sections.items():
        out += [f"subsection {sec}"] + lines + ["end"]
    return "\n".join(out) + "\n"


class Problem:
    def __init__(self, prm: str):
        self.L = load()
        err = C.create_string_buffer(512)
        self.h = C.c_void_p(self.L.step50_create(prm.encode(), err, 512))
        if not self.h:
            raise RuntimeError(err.value.decode())

    def close(self):
        if self.h:
            self.L.step50_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc, what):
        if rc != 0:
            raise RuntimeError(f"{what} failed (rc={rc}): {self.L.step50_last_error(self.h).decode()}")

    def read_lammps(self, path):
        self._chk(self.L.step50_read_lammps(self.h, path.encode()), "read_lammps_input_file")

    def set_atoms(self, q, xyz):
        q = np.ascontiguousarray(q, dtype=np.float64)
        x = np.ascontiguousarray(xyz, dtype=np.float64)
        self._chk(self.L.step50_set_atoms(self.h, C.c_int64(len(q)), q.ctypes.data_as(C.POINTER(C.c_double)),
                                          x.ctypes.data_as(C.POINTER(C.c_double))), "set_atoms")

    def set_nacl_atoms(self, n_cells):
        self._chk(self.L.step50_set_nacl_atoms(self.h, C.c_int(n_cells)), "set_nacl_atoms")

    def atoms(self):
        n = self.L.step50_n_atoms(self.h)
        q, x = np.empty(n), np.empty((n, 3))
        self.L.step50_get_atoms(self.h, q.ctypes.data_as(C.POINTER(C.c_double)), x.ctypes.data_as(C.POINTER(C.c_double)))
        return q, x

    def run_cycle(self, cycle: int, on_device: bool = True):
        self._chk(self.L.step50_run_cycle(self.h, C.c_int(cycle), C.c_int(1 if on_device else 0)), f"cycle {cycle}")
        return self.report(-1)

    def finish_cycle_with(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        self._chk(self.L.step50_finish_cycle_with(self.h, x.ctypes.data_as(C.POINTER(C.c_double)), C.c_int64(len(x))), "finish_cycle")
        return self.report(-1)

    def solve_again(self):
        self._chk(self.L.step50_solve_again(self.h), "solve")
        return self.report(-1)

    def estimator_components(self):
        """Per active cell of the cycle just estimated: (Kelly face sum eta_K^2, residual term, level, centre)."""
        self.L.step50_n_active_cells.restype = C.c_int64
        n = self.L.step50_n_active_cells(self.h)
        k, r, lv, ctr = np.zeros(n), np.zeros(n), np.zeros(n, dtype=np.int32), np.zeros((n, 3))
        self._chk(self.L.step50_estimator_components(self.h, k.ctypes.data_as(C.POINTER(C.c_double)), r.ctypes.data_as(C.POINTER(C.c_double)),
                                                     lv.ctypes.data_as(C.POINTER(C.c_int32)), ctr.ctypes.data_as(C.POINTER(C.c_double))), "estimator_components")
        return k, r, lv, ctr

    def set_smoother(self, smoother: str, ssor_blocks: int = 1):
        """Another smoother on the operators of the cycle just run (re-uploads them; the next solve_again uses it)."""
        self._chk(self.L.step50_set_smoother(self.h, smoother.encode(), C.c_int(ssor_blocks)), "set_smoother")

    def report(self, i=-1) -> dict:
        r = Report()
        self._chk(self.L.step50_get_report(self.h, C.c_int(i), C.byref(r)), "get_report")
        return r.as_dict()

    def log(self) -> str:
        return self.L.step50_log(self.h).decode()

    def n_levels(self):
        return int(self.L.step50_n_levels(self.h))

    def n_dofs(self):
        return int(self.L.step50_n_dofs(self.h))

    def matrix(self, kind, level=0):
        """kind: 'system' | 'level' | 'edge' | 'prolongation' -> namespace(n_rows, n_cols, rowptr, col, val, nnz)"""
        k = {"system": 0, "level": 1, "edge": 2, "prolongation": 3}[kind]
        nr, nc, nz = C.c_int64(0), C.c_int64(0), C.c_int64(0)
        self._chk(self.L.step50_matrix_shape(self.h, C.c_int(k), C.c_int(level), C.byref(nr), C.byref(nc), C.byref(nz)), "matrix")
        rp = np.zeros(nr.value + 1, dtype=np.int64)
        col = np.zeros(max(nz.value, 1), dtype=np.int32)
        val = np.zeros(max(nz.value, 1), dtype=np.float64)
        if nr.value > 0:
            self.L.step50_matrix_copy(self.h, C.c_int(k), C.c_int(level), rp.ctypes.data_as(C.POINTER(C.c_int64)),
                                      col.ctypes.data_as(C.POINTER(C.c_int32)), val.ctypes.data_as(C.POINTER(C.c_double)))
        return SimpleNamespace(n_rows=nr.value, n_cols=nc.value, rowptr=rp, col=col[:nz.value], val=val[:nz.value], nnz=nz.value)

    def matrix_shape(self, kind, level=0):
        """(n_rows, nnz) of an operator of the current cycle without copying it."""
        k = {"system": 0, "level": 1, "edge": 2, "prolongation": 3}[kind]
        nr, nc, nz = C.c_int64(0), C.c_int64(0), C.c_int64(0)
        self._chk(self.L.step50_matrix_shape(self.h, C.c_int(k), C.c_int(level), C.byref(nr), C.byref(nc), C.byref(nz)), "matrix_shape")
        return int(nr.value), int(nz.value)

    def copy_indices(self, level):
        n = self.L.step50_copy_indices_size(self.h, C.c_int(level))
        g, l = np.zeros(max(n, 1), dtype=np.int32), np.zeros(max(n, 1), dtype=np.int32)
        if n:
            self.L.step50_copy_indices(self.h, C.c_int(level), g.ctypes.data_as(C.POINTER(C.c_int32)),
                                       l.ctypes.data_as(C.POINTER(C.c_int32)))
        return g[:n], l[:n]

    def vector(self, which):
        w = {"rhs": 0, "solution": 1, "initial_guess": 2}[which]
        out = np.zeros(self.n_dofs())
        self.L.step50_get_vector(self.h, C.c_int(w), out.ctypes.data_as(C.POINTER(C.c_double)))
        return out

    def total_charge_density(self):
        """Per DoF, the integrated charge density of its cells (the check vector of tests_rhs_rc_variation)."""
        out = np.zeros(self.n_dofs())
        self.L.step50_total_charge_density(self.h, out.ctypes.data_as(C.POINTER(C.c_double)))
        return out

    def dof_coordinates(self):
        out = np.zeros((self.n_dofs(), 3))
        self.L.step50_dof_coordinates(self.h, out.ctypes.data_as(C.POINTER(C.c_double)))
        return out

    def constrained_mask(self):
        out = np.zeros(self.n_dofs(), dtype=np.int8)
        self.L.step50_constrained_mask(self.h, out.ctypes.data_as(C.POINTER(C.c_int8)))
        return out.astype(bool)

    def hierarchy(self):
        """What LaplaceProblem::solve consumes, as plain arrays (for the oracle in tests)."""
        L = self.n_levels()
        return SimpleNamespace(
            system_matrix=self.matrix("system"), system_rhs=self.vector("rhs"),
            level_matrices=[self.matrix("level", l) for l in range(L)],
            edge_matrices=[self.matrix("edge", l) for l in range(L)],
            prolongations=[self.matrix("prolongation", l) for l in range(L - 1)],
            copy_global=[self.copy_indices(l)[0] for l in range(L)],
            copy_level=[self.copy_indices(l)[1] for l in range(L)],
            constrained=self.constrained_mask())

    def set_communicator(self, rank, n_ranks, uid: bytes):
        buf = C.create_string_buffer(uid, 128)
        self._chk(self.L.step50_set_communicator(self.h, C.c_int(rank), C.c_int(n_ranks), buf), "set_communicator")

    def localize(self, kind, level, rank, n_ranks):
        """partition.h applied to one operator: local CSR ([owned | ghost] columns) + halo plan."""
        k = {"system": 0, "level": 1}[kind]

        class Info(C.Structure):
            _fields_ = [(n, C.c_int64) for n in ("n_rows", "n_cols", "nnz", "row_begin", "n_neighbors", "n_send")]

        info = Info()
        self._chk(self.L.step50_localize(self.h, C.c_int(k), C.c_int(level), C.c_int(rank), C.c_int(n_ranks), C.byref(info)), "localize")
        rp = np.zeros(info.n_rows + 1, dtype=np.int64)
        col = np.zeros(max(info.nnz, 1), dtype=np.int32)
        val = np.zeros(max(info.nnz, 1))
        nb = np.zeros(max(info.n_neighbors, 1), dtype=np.int32)
        sc, rc = np.zeros_like(nb), np.zeros_like(nb)
        si = np.zeros(max(info.n_send, 1), dtype=np.int32)
        gg = np.zeros(max(info.n_cols - info.n_rows, 1), dtype=np.int64)
        P = lambda a, t: a.ctypes.data_as(C.POINTER(t))
        self.L.step50_localize_copy(P(rp, C.c_int64), P(col, C.c_int32), P(val, C.c_double), P(nb, C.c_int32), P(sc, C.c_int32),
                                    P(si, C.c_int32), P(rc, C.c_int32), P(gg, C.c_int64))
        nn = info.n_neighbors
        return SimpleNamespace(n_rows=info.n_rows, n_cols=info.n_cols, rowptr=rp, col=col[:info.nnz], val=val[:info.nnz],
                               nnz=info.nnz, row_begin=info.row_begin, neighbor_rank=nb[:nn], send_count=sc[:nn],
                               recv_count=rc[:nn], send_idx=si[:info.n_send], ghost_global=gg[:info.n_cols - info.n_rows])

    def gmg_context(self):
        return C.c_void_p(self.L.step50_gmg_context(self.h))
