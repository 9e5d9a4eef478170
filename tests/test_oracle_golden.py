"""CPU: the restatement (oracle/fluid_oracle.c) against the committed vectors
captured from the compiled reference (tests/golden/make_golden.py).  Bit-exact."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, assert_bit_equal, load_golden


@pytest.mark.parametrize("n", [14, 30, 61])
def test_operators(oracle, n):
    g = load_golden("ops_n%d.npz" % n)
    for b in (0, 1, 2):
        x = g["bnd_in"].copy()
        oracle.set_bnd(b, x)
        assert_bit_equal(x, g["bnd_out_b%d" % b], "set_bnd b=%d" % b)
    x = g["src_x"].copy()
    oracle.add_source(x, g["src_s"])
    assert_bit_equal(x, g["src_out"], "add_source")
    for k, (b, a, be) in enumerate(g["dif_params"]):
        for iters in (2, 40):
            x = g["dif%d_x" % k].copy()
            oracle.diffuse(int(b), x, g["dif%d_x0" % k].copy(), float(a), float(be), iters)
            assert_bit_equal(x, g["dif%d_out%d" % (k, iters)], "diffuse case %d, %d sweeps" % (k, iters))
    p, div = g["div_p_in"].copy(), g["div_div_in"].copy()
    oracle.divergence(g["div_u"], g["div_v"], p, div)
    assert_bit_equal(p, g["div_p"], "pressure clear")
    assert_bit_equal(div, g["div_div"], "divergence")
    u, v = g["grad_u"].copy(), g["grad_v"].copy()
    oracle.subtract_gradient(u, v, g["grad_p"])
    assert_bit_equal(u, g["grad_u_out"], "gradient u")
    assert_bit_equal(v, g["grad_v_out"], "gradient v")
    for tag in ("small", "clamp"):
        u, v, d0 = g["adv_%s_u" % tag], g["adv_%s_v" % tag], g["adv_%s_d0" % tag]
        for b in (0, 1, 2):
            d = np.full_like(d0, 7.0)
            oracle.advect(b, d, d0, u, v)
            assert_bit_equal(d, g["adv_%s_out_b%d" % (tag, b)], "advect %s b=%d" % (tag, b))
        d = np.full_like(d0, 7.0)
        oracle.advect(1, d, u, u, v)
        assert_bit_equal(d, g["adv_%s_self_u" % tag], "self-advect u")
        d = np.full_like(d0, 7.0)
        oracle.advect(2, d, v, u, v)
        assert_bit_equal(d, g["adv_%s_self_v" % tag], "self-advect v")


def test_coefficients(oracle):
    g = load_golden("ops_n30.npz")
    from oracle.oracle import DIFF, DT, VISC
    av, bv = oracle.coefficients(30, DT, VISC)
    ad, bd = oracle.coefficients(30, DT, DIFF)
    want = g["dif_params"]
    assert (np.float32(want[0][1]), np.float32(want[0][2])) == (np.float32(av), np.float32(bv))
    assert (np.float32(want[2][1]), np.float32(want[2][2])) == (np.float32(ad), np.float32(bd))


@pytest.mark.parametrize("n,iters,steps", [(30, 40, (1, 2, 5)), (61, 40, (1, 2, 5)), (126, 40, (1, 2, 5)),
                                           (126, 20, (1, 2))])
def test_full_steps(oracle, n, iters, steps):
    g = load_golden("step_n%d_k%d.npz" % (n, iters))
    # the reference's own initializeParameters: glibc rand(), default seed
    dens, dens0, u, u0, v, v0 = oracle.initialize_glibc(n, seed=1)
    assert_bit_equal(u0, g["init_u_prev"], "init u_prev")
    assert_bit_equal(v0, g["init_v_prev"], "init v_prev")
    assert_bit_equal(dens0, g["init_dens_prev"], "init dens_prev")
    for z in range(1, max(steps) + 1):
        if z == 1:
            oracle.step_src(u, v, dens, u0, v0, dens0, iters=iters)
            assert_bit_equal(u0, g["s1_u_prev"], "pressure left in u_prev")
            assert_bit_equal(v0, g["s1_v_prev"], "divergence left in v_prev")
            assert_bit_equal(dens0, g["s1_dens_prev"], "diffused density left in dens_prev")
        else:
            oracle.step(u, v, dens, u0, v0, dens0, iters=iters)
        if z in steps:
            assert_bit_equal(u, g["s%d_u" % z], "u after step %d" % z)
            assert_bit_equal(v, g["s%d_v" % z], "v after step %d" % z)
            assert_bit_equal(dens, g["s%d_dens" % z], "dens after step %d" % z)


def test_checksums_survey_values(oracle):
    """SURVEY.md 8(c) quotes these sums for N=126 step 1; checksums.json holds
    the larger grids (N=254 checked here, 1022/4094 on the GPU box)."""
    rows = json.load(open(os.path.join(GOLDEN, "checksums.json")))
    row = [r for r in rows if r["n"] == 254][0]
    dens, dens0, u, u0, v, v0 = oracle.initialize_glibc(254, seed=1)
    oracle.step_src(u, v, dens, u0, v0, dens0)
    assert float(u.sum(dtype=np.float64)) == row["sum_u"]
    assert float(v.sum(dtype=np.float64)) == row["sum_v"]
    assert float(dens.sum(dtype=np.float64)) == row["sum_dens"]
    import zlib
    for n in (254, 1022):          # CRC-32 of the reference's bytes (the GPU tests pin 4094 and 8190 the same way)
        row = [r for r in rows if r["n"] == n][0]
        dens, dens0, u, u0, v, v0 = oracle.initialize_glibc(n, seed=1)
        oracle.step_src(u, v, dens, u0, v0, dens0)
        for name, a in (("u", u), ("v", v), ("dens", dens)):
            assert zlib.crc32(a.view(np.uint8).reshape(-1)) == row["crc_" + name], "%s at N=%d" % (name, n)
    g = load_golden("step_n126_k40.npz")
    assert abs(float(g["s1_u"].sum(dtype=np.float64)) - 117.14561) < 1e-5
    assert abs(float(g["s1_dens"].sum(dtype=np.float64)) - 35.0967363) < 1e-6


def test_trajectory_checksums_at_1022(oracle):
    """The reference's loop over ten steps at N = 1022 (sources at step 0 only; the fields decay through five orders of
    magnitude): the restatement reproduces the CRC-32 of u, v and dens the compiled reference left after every step
    (tests/golden/trajectory_checksums.json; the GPU follows the same file, and N = 4094, in test_gpu_large.py)."""
    import zlib
    rows = [r for r in json.load(open(os.path.join(GOLDEN, "trajectory_checksums.json"))) if r["n"] == 1022]
    assert [r["step"] for r in rows] == list(range(1, 11))
    dens, dens0, u, u0, v, v0 = oracle.initialize_glibc(1022, seed=1)
    for row in rows:
        if row["step"] == 1:
            oracle.step_src(u, v, dens, u0, v0, dens0)
        else:
            oracle.step(u, v, dens, u0, v0, dens0)
        for name, a in (("u", u), ("v", v), ("dens", dens)):
            assert zlib.crc32(a.view(np.uint8).reshape(-1)) == row["crc_" + name], "%s after step %d" % (name, row["step"])


def test_odd_and_zero_sweeps(oracle):
    """The restatement accepts an odd count (result copied back) -- the ABI
    rejects it; zero sweeps leaves x untouched."""
    rng = np.random.default_rng(5)
    x, x0 = rng.random((16, 16), dtype=np.float32), rng.random((16, 16), dtype=np.float32)
    a = x.copy()
    oracle.diffuse(0, a, x0, 1.0, 4.0, 0)
    assert_bit_equal(a, x, "zero sweeps")
    one = x.copy()
    oracle.diffuse(0, one, x0, 1.0, 4.0, 1)
    out = np.zeros_like(x)
    oracle.jacobi_sweep(0, x, x0, out, 1.0, 4.0)
    assert_bit_equal(one, out, "one sweep == jacobi_sweep")
