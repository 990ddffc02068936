"""The right-hand side integrated on the device (gmg_rhs_assemble; reference: assemble_system, src/step-50.cc:813-828) from
charge densities that never leave HBM, against the host loop on the same meshes -- adaptive cycles with hanging nodes, all
three boundary-condition kinds (the Dirichlet terms -K_ij g_j of :825-828) -- bit for bit, and against the reference's own
numbers: the rhs norms of tests/gaussian-charges.mpirun=1.output and of tests_rhs_rc_variation."""
import os

import numpy as np
import pytest

from conftest import rel_close
from gpu_util import pkg

pytestmark = pytest.mark.gpu


def _problem(S, golden_dir, bc, on, cycles=3, quad=4):
    p = S.Problem(S.prm_text(left=0, right=1, mesh_size=0.25, vacuum=10, problem="GaussianCharges", dim=3, bc=bc, cycles=cycles, r_c=0.5,
                             cutoff=3.5, rhs_optimization=True, quad_rhs=quad, global_refinement=0, smoother="SSOR", rhs_on_device=on))
    p.read_lammps(os.path.join(golden_dir, "atom_n1_2.data"))
    return p


@pytest.mark.parametrize("bc", ["Exact", "Inhomogeneous", "Homogeneous"])
def test_device_rhs_equals_host_rhs_on_adaptive_meshes(golden_dir, bc):
    S = pkg().step50
    pd, ph = _problem(S, golden_dir, bc, True), _problem(S, golden_dir, bc, False)
    for c in range(3):
        rd, rh = pd.run_cycle(c, on_device=True), ph.run_cycle(c, on_device=True)
        assert rd["dofs_by_level"] == rh["dofs_by_level"]
        bd, bh = pd.vector("rhs"), ph.vector("rhs")
        assert np.array_equal(bd, bh), (c, float(np.abs(bd - bh).max()))
        assert rd["cg_iterations"] == rh["cg_iterations"] and rd["rhs_l2"] == rh["rhs_l2"]
    pd.close(); ph.close()


def test_device_rhs_reproduces_the_reference_log(golden, golden_dir):
    """tests/gaussian-charges.mpirun=1.output: L1 / L2 / Linfty norms of the rhs of every adaptive cycle, to every printed digit."""
    S = pkg().step50
    G = golden["tests/gaussian-charges.mpirun=1"]["runs"][0]["cycles"]
    p = _problem(S, golden_dir, "Exact", True, cycles=4)
    for c in range(4):
        r = p.run_cycle(c, on_device=True)
        for k in ("rhs_l1", "rhs_l2", "rhs_linf"):
            assert rel_close(r[k], G[c][k], 11), (c, k, r[k], G[c][k])
    p.close()
