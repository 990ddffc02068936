#!/usr/bin/env python3
"""One rank of tests/test_gpu_two_ranks.py: the six adaptive cycles of the reference's regression
test with the solve on the GPU in the one-process-per-GPU layout (partitioned system matrix and
level 0, halo exchange, all-reduces, all-gathers), on the communicator named by the id.

usage: two_rank_worker.py RANK N_RANKS ID_HEX GOLDEN_DIR OUT_JSON [nacl [auto|always|never [ssor_blocks]]]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # repo root: the package and oracle/
from gpu_util import pkg  # noqa: E402


def main():
    rank, n_ranks, uid = int(sys.argv[1]), int(sys.argv[2]), bytes.fromhex(sys.argv[3])
    golden_dir, out = sys.argv[4], sys.argv[5]
    nacl = int(sys.argv[6]) if len(sys.argv) > 6 else 0
    part = sys.argv[7] if len(sys.argv) > 7 else "always"  # small problems: "auto" would keep level 0 replicated
    blocks = int(sys.argv[8]) if len(sys.argv) > 8 else 1   # SSOR blocks (0: one per rank, the reference on that many ranks)
    S = pkg().step50
    if nacl:  # the cluster runs' problem (BASELINE configs 2-5): NaCl lattice of 8 nacl^3 atoms, SSOR smoother, Kelly marking
        cycles = 2
        p = S.Problem(S.prm_text(left=0, right=float(nacl), mesh_size=0.25, vacuum=10, problem="GaussianCharges", dim=3,
                                 bc="Inhomogeneous", cycles=cycles, r_c=0.5, cutoff=3.5, rhs_optimization=True, quad_rhs=1,
                                 global_refinement=0, smoother="SSOR", ssor_blocks=blocks, partition_level0=part,
                                 refinement_estimator="Kelly"))
        p.set_nacl_atoms(nacl)
    else:
        cycles = 6
        p = S.Problem(S.prm_text(left=0, right=1, mesh_size=0.25, vacuum=10, problem="GaussianCharges", dim=3, bc="Exact",
                                 cycles=cycles, r_c=0.5, cutoff=3.5, rhs_optimization=True, quad_rhs=4, global_refinement=0,
                                 smoother="SSOR", partition_level0=part, ssor_blocks=blocks))
        p.read_lammps(os.path.join(golden_dir, "atom_n1_2.data"))
    if n_ranks > 0:
        p.set_communicator(rank, n_ranks, uid)
    reps = [dict(p.run_cycle(c, on_device=True)) for c in range(cycles)]
    with open(out, "w") as fh:
        json.dump(reps, fh)


if __name__ == "__main__":
    main()
