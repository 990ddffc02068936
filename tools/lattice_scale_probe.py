#!/usr/bin/env python3
"""The level-0 operator formed on the device (gmg_set_level_matrix_lattice: no CSR exists anywhere) on lattices far beyond the
caches, timed: python tools/lattice_scale_probe.py N [N ...].  Bytes: 17 per row (x read once, y written, one class byte)."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
pkg = importlib.import_module("geometric-multigrid-preconditioners-for-long-range-coulomb-interaction_amd")
capi = pkg.capi
# Q1 Laplace cell matrix of a cube of edge h (vertex a = bx + 2 by + 4 bz): 1/3 on the diagonal, 0 along edges, -1/12 across
# faces and the body diagonal -- times h
h = 0.25
Ke = np.empty((8, 8))
for a in range(8):
    for b in range(8):
        d = bin(a ^ b).count("1")
        Ke[a, b] = h * (1.0 / 3.0, 0.0, -1.0 / 12.0, -1.0 / 12.0)[d]
for n in [int(v) for v in sys.argv[1:]] or [201, 301, 401]:
    c = capi.Context(1)
    c.set_level_matrix_lattice(0, (n, n, n), Ke)
    N = n ** 3
    x, y = c.vector(N + 2, np.random.default_rng(0).standard_normal(N + 2)), c.vector(N + 2)
    reps = max(20, int(2e9 / (17 * N)))
    for _ in range(3):
        c.spmv(0, y, x)
    c.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        c.spmv(0, y, x)
    c.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"{n}^3 = {N} rows: {dt*1e6:.1f} us per y = A x, {17 * N / 1e6:.0f} MB algorithmic -> {17 * N / dt / 1e9:.0f} GB/s = {17 * N / dt / 8e12:.3f} of the HBM peak", flush=True)
    c.close()
