#!/usr/bin/env python3
"""Wall time of the HOST phases of the adaptive loop without a GPU: cycle 0 .. n of the bench's workload, the solve replaced
by a made-up solution (a sum of a few smooth bumps: the marking only has to refine somewhere).  STEP50_TIMING=1 is set, the
phases print to stderr.  usage: host_setup_profile.py [workload] [cycles] [threads]"""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["STEP50_TIMING"] = "1"
import bench  # noqa: E402

pkg = importlib.import_module(bench.PKG)
S = pkg.step50


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "atoms64000"
    cycles = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    if len(sys.argv) > 3:
        S.set_threads(int(sys.argv[3]))
    w = bench.WORKLOADS[wl]
    t0 = time.time()
    p = S.Problem(S.prm_text(left=0, right=w["box"], mesh_size=0.25, vacuum=10, problem="GaussianCharges", dim=3,
                             bc="Inhomogeneous", cycles=cycles, r_c=0.5, cutoff=3.5, rhs_optimization=True, quad_rhs=1,
                             global_refinement=0, smoother="SSOR", refinement_estimator="Kelly", rhs_on_device=False,
                             level0_on_device=False, transfer_on_device=False))
    p.set_nacl_atoms(w["nacl"])
    print("problem %.2f s" % (time.time() - t0), file=sys.stderr)
    for c in range(cycles):
        t0 = time.time()
        p.run_cycle(c, on_device=False)
        t1 = time.time()
        if c + 1 < cycles:
            xyz = p.dof_coordinates()
            ctr = 0.5 * w["box"]
            x = np.exp(-((xyz - ctr) ** 2).sum(axis=1) / (0.05 * w["box"]) ** 2)
            t2 = time.time()
            p.finish_cycle_with(x)
            print("cycle %d: host phases %.2f s, made-up solution %.2f s, estimate + mark %.2f s" % (c, t1 - t0, t2 - t1, time.time() - t2), file=sys.stderr)
        else:
            print("cycle %d: host phases %.2f s" % (c, t1 - t0), file=sys.stderr)


main()
