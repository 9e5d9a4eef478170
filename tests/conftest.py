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


def tb_schedule(iters, max_t, deep=True):
    """Launch depths of one fp32 solve through the fused Jacobi kernel, as fluid_solver.hip plans them
    (pick_sweeps): where deep launches pay (`deep`: large grids, or PARAM_TB_T16_MIN_CELLS = 0 in tests) the
    multiset of depths 16 / 12 / 8 / 4 / 2 (<= max_t) that adds up to the sweeps left at the least estimated
    cost, deepest first; otherwise (and for remainders below 12) greedy 8 / 4 / 2."""
    depth = (16, 12, 8, 4, 2)
    w = (1.00, 1.04, 1.30, 2.6, 5.0)
    out = []
    left = iters
    while left:
        greedy = next(t for t in (8, 4, 2) if t <= min(max_t, left))
        if not deep or max_t < 12 or left < 12:
            t = greedy
        else:
            ok = [d <= max_t for d in depth]
            cost = [0.0] + [1e300] * left
            for r in range(2, left + 1, 2):
                for k, d in enumerate(depth):
                    if ok[k] and d <= r:
                        cost[r] = min(cost[r], cost[r - d] + d * w[k] + 0.5)
            t = next(d for k, d in enumerate(depth) if ok[k] and d <= left and abs(cost[left - d] + d * w[k] + 0.5 - cost[left]) < 1e-9)
        out.append(t)
        left -= t
    return out
