import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


@pytest.fixture(scope="session")
def golden():
    return load_golden


def relerr(a, b):
    a = np.asarray(a)
    b = np.asarray(b)
    denom = np.max(np.abs(b))
    if denom == 0:
        return float(np.max(np.abs(a - b)))
    return float(np.max(np.abs(a - b)) / denom)


def elem_err(a, b, rtol=1e-10, floor=1e-13):
    """Element-wise parity measure: max over entries of |a-b| / (rtol |b| + floor max|b|).
    <= 1 means every entry agrees to ``rtol`` relative, except entries smaller than
    floor/rtol of the largest one, which must agree to ``floor`` of the largest (a sum of
    thousands of terms cannot do better than its own rounding).  ``relerr`` alone is a max-norm
    test: a source 1e-6 of the brightest could be wrong at 1e-4 and pass it."""
    a = np.asarray(a)
    b = np.asarray(b)
    scale = float(np.max(np.abs(b))) if b.size else 0.0
    if scale == 0.0:
        return float(np.max(np.abs(a))) if a.size else 0.0
    return float(np.max(np.abs(a - b) / (rtol * np.abs(b) + floor * scale)))


def operr(a, b, rtol=1e-10, floor=1e-13):
    """Parity measure for operator entries, to be compared with 1e-10: the larger of the max-norm
    relative error and ``rtol`` times the element-wise measure -- i.e. ``operr < 1e-10`` demands
    BOTH max|a-b| < 1e-10 max|b| and |a-b| <= 1e-10 |b| + 1e-13 max|b| for every entry."""
    return max(relerr(a, b), rtol * elem_err(a, b, rtol=rtol, floor=floor))
