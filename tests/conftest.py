import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    with open(os.path.join(GOLDEN_DIR, "reference_logs.json")) as fh:
        return json.load(fh)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN_DIR


def rel_close(a, b, digits):
    """True when ``a`` equals the printed golden ``b`` to its printed precision (half a unit
    in the last printed digit, expressed relatively)."""
    if b == 0:
        return abs(a) < 10.0 ** (-digits)
    return abs(a - b) <= 0.6 * 10.0 ** (-digits + 1) * abs(b)


@pytest.fixture(scope="session", autouse=True)
def _device_context_first(request):
    """On a GPU box the test process opens its device context before any test starts rank processes of its own (the
    multi-rank tests run 2-3 children on the same GPU; a parent that first touches the device after several such groups was
    once refused a context).  No GPU: nothing happens."""
    if "not gpu" in (request.config.getoption("-m") or ""):
        return
    try:
        from gpu_util import capi

        capi().Context(1).close()
    except Exception:
        pass
