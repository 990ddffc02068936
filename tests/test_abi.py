"""CPU-side checks of the drop-in boundary: the C-ABI library builds for gfx950, loads, and
exports every symbol include/gmg_coulomb.h declares.  No compute calls (no GPU here)."""
import os
import re

from gpu_util import capi, pkg

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "gmg_coulomb.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gmg_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    pkg().build.build_device()
    lib = capi().load()
    declared = _declared_symbols()
    assert len(declared) >= 35
    missing = [s for s in declared if not hasattr(lib, s)]
    assert not missing, missing
    assert sorted(capi().SYMBOLS) == declared


def test_header_cites_reference_call_sites():
    text = open(os.path.join(ROOT, "include", "gmg_coulomb.h")).read()
    for cite in ("src/step-50.cc:938-1017", ":991", ":957-958", ":970", ":962", "include/step_50.h:154"):
        assert cite in text


def test_create_without_gpu_fails_loudly():
    import torch

    if torch.cuda.is_available():
        import pytest

        pytest.skip("GPU present")
    import pytest

    with pytest.raises(capi().GMGError):
        capi().Context(1, 0)
