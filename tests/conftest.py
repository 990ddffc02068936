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
