"""CPU, build container only: the restatement against the reference itself
(oracle/_ref/*.so, built from /root/reference by oracle/build_ref.sh) on fresh
random inputs.  Skipped where neither the reference tree nor a prebuilt
oracle/_ref is present."""
import numpy as np
import pytest

from conftest import assert_bit_equal, rnd
from oracle.oracle import DIFF, DT, VISC, Reference, can_build_ref, have_ref


def _ref(n, iters=40):
    if not (have_ref(n, iters) or can_build_ref()):
        pytest.skip("reference build for N=%d not available here" % n)
    return Reference(n, iters)


@pytest.mark.parametrize("n", [14, 30, 61, 126])
def test_each_operator(oracle, n):
    r = _ref(n)
    rng = np.random.default_rng(n)
    for b in (0, 1, 2):
        x = rnd(rng, n)
        y = x.copy()
        r.set_bnd(b, x)
        oracle.set_bnd(b, y)
        assert_bit_equal(y, x, "set_bnd")
    x, s = rnd(rng, n), rnd(rng, n)
    y = x.copy()
    r.add_source(x, s)
    oracle.add_source(y, s)
    assert_bit_equal(y, x, "add_source")
    for b, coef in ((1, VISC), (2, VISC), (0, DIFF), (0, None)):
        a, be = (1.0, 4.0) if coef is None else oracle.coefficients(n, DT, coef)
        x, x0 = rnd(rng, n), rnd(rng, n)
        y = x.copy()
        r.diffuse(b, x, x0.copy(), a, be)
        oracle.diffuse(b, y, x0, a, be, 40)
        assert_bit_equal(y, x, "diffuse b=%d" % b)
    u, v = rnd(rng, n), rnd(rng, n)
    p, d = rnd(rng, n), rnd(rng, n)
    p2, d2 = p.copy(), d.copy()
    r.divergence(u, v, p, d)
    oracle.divergence(u, v, p2, d2)
    assert_bit_equal(p2, p, "p")
    assert_bit_equal(d2, d, "div")
    u2, v2 = u.copy(), v.copy()
    pr = rnd(rng, n)
    r.subtract_gradient(u, v, pr)
    oracle.subtract_gradient(u2, v2, pr)
    assert_bit_equal(u2, u, "grad u")
    assert_bit_equal(v2, v, "grad v")
    for amp in (0.01, 1.0, 4.0 / DT):
        u, v, d0 = rnd(rng, n, -amp, amp), rnd(rng, n, -amp, amp), rnd(rng, n)
        for b in (0, 1, 2):
            d, e = rnd(rng, n), rnd(rng, n)
            r.advect(b, d, d0, u, v)
            oracle.advect(b, e, d0, u, v)
            assert_bit_equal(e, d, "advect amp=%g b=%d" % (amp, b))


@pytest.mark.parametrize("n,iters", [(30, 40), (61, 40), (126, 40), (126, 20), (254, 40)])
def test_steps_from_reference_init(oracle, n, iters):
    r = _ref(n, iters)
    a = r.initialize()
    b = oracle.initialize_glibc(n)
    for x, y in zip(a, b):
        assert_bit_equal(y, x, "initializeParameters")
    dens, dens0, u, u0, v, v0 = a
    d2, d02, u2, u02, v2, v02 = b
    r.step_src(u, v, dens, u0, v0, dens0)
    oracle.step_src(u2, v2, d2, u02, v02, d02, iters=iters)
    for z in range(3):
        for x, y, w in ((u, u2, "u"), (v, v2, "v"), (dens, d2, "dens"), (u0, u02, "u_prev"), (v0, v02, "v_prev"),
                        (dens0, d02, "dens_prev")):
            assert_bit_equal(y, x, "%s after step %d" % (w, z + 1))
        r.step(u, v, dens, u0, v0, dens0)
        oracle.step(u2, v2, d2, u02, v02, d02, iters=iters)


def test_portable_init_shape(oracle):
    """The portable generator follows the same recipe as initializeParameters
    (FluidSequential.c:244-271): density only in the centred square, velocities
    k/100 everywhere, current fields zero."""
    n = 62
    dens, dens0, u, u0, v, v0 = oracle.initialize_portable(n, seed=1)
    w, c, r = n + 2, (n + 2) // 2, (n + 2) // 8
    mask = np.zeros((w, w), bool)
    mask[c - r:c + r, c - r:c + r] = True
    assert (dens0[~mask] == 0).all() and dens0[mask].max() < 0.1
    assert not u.any() and not v.any() and not dens.any()
    for f in (u0, v0):
        k = np.rint(f * 100)
        assert np.array_equal((k / 100.0).astype(np.float32), f) and k.min() >= 0 and k.max() <= 99
