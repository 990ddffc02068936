"""MI355X-native GMG-preconditioned CG hot path of the Step50 Poisson/Coulomb solver.

The product is native code: ``csrc/libgmgcoulomb.so`` (hand-written HIP for gfx950 behind the
C-ABI of ``include/gmg_coulomb.h``) and the host-side C++ in ``csrc/host`` that mirrors the
reference's ``LaplaceProblem``.  This Python package only binds them for tests and bench.py.
The directory name contains hyphens; import it with

    importlib.import_module("geometric-multigrid-preconditioners-for-long-range-coulomb-interaction_amd")
"""
from . import build, capi, step50  # noqa: F401
