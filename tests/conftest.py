import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def assert_bit_equal(got, want, what=""):
    """Bit-exact comparison with a readable failure (first mismatches, max error)."""
    got, want = np.asarray(got), np.asarray(want)
    assert got.shape == want.shape, "%s: shape %s vs %s" % (what, got.shape, want.shape)
    bad = bits(got) != bits(want)
    if bad.any():
        idx = np.argwhere(bad)
        i, j = idx[0]
        err = np.abs(got.astype(np.float64) - want.astype(np.float64))
        raise AssertionError(
            "%s: %d of %d cells differ bitwise; first at (row %d, col %d): got %r want %r; max abs err %.3g"
            % (what, bad.sum(), bad.size, i, j, got[i, j], want[i, j], np.nanmax(err)))


def assert_close_fp32(got, want, what="", rel=1e-5):
    """The contractual tolerance of BASELINE.json's north_star: 1e-5 relative fp32
    per cell, with the absolute floor SURVEY.md section 7 defines for cells -> 0:
    |a-b| <= 1e-5*max(|a|,|b|) + 1e-7*max|field|."""
    got, want = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
    tol = rel * np.maximum(np.abs(got), np.abs(want)) + 1e-7 * np.abs(want).max()
    bad = np.abs(got - want) > tol
    assert not bad.any(), "%s: %d cells outside 1e-5 relative tolerance (max err %.3g)" % (
        what, bad.sum(), np.abs(got - want).max())


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def rnd(rng, n, lo=-1.0, hi=1.0):
    return rng.uniform(lo, hi, size=(n + 2, n + 2)).astype(np.float32)


@pytest.fixture(scope="session")
def oracle():
    from oracle.oracle import Oracle
    return Oracle()
