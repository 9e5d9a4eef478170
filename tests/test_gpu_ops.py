"""GPU: every operator of the hot path, through the C ABI, against the oracle
(fresh random inputs) and the committed golden vectors from the compiled
reference.  Bar: bit-exact (stricter than the 1e-5 relative of north_star)."""
import numpy as np
import pytest

from conftest import assert_bit_equal, load_golden, rnd, tb_schedule

pytestmark = pytest.mark.gpu

DT, VISC, DIFF = 0.016, 0.0025, 0.1
SIZES = [1, 2, 3, 4, 5, 14, 30, 61, 64, 126, 255, 256, 257, 1022]
VARIANTS = [0, 1, 2, 3]       # stream, LDS-tiled, naive-global, temporally blocked


@pytest.fixture(scope="module")
def F():
    import fluidsimulationcuda_amd as F
    return F


@pytest.mark.parametrize("n", SIZES)
def test_set_bnd(F, oracle, n):
    rng = np.random.default_rng(n)
    with F.FluidSolver(n) as s:
        for b in (0, 1, 2):
            x = rnd(rng, n)
            s.upload(u=x)
            s.set_bnd(b, "u")
            want = x.copy()
            oracle.set_bnd(b, want)
            assert_bit_equal(s.download("u"), want, "set_bnd n=%d b=%d" % (n, b))


@pytest.mark.parametrize("n", [1, 14, 61, 255, 1022])
def test_add_source(F, oracle, n):
    rng = np.random.default_rng(n)
    x, src = rnd(rng, n), rnd(rng, n)
    with F.FluidSolver(n) as s:
        s.upload(u=x, u_prev=src)
        s.add_source("u", "u_prev", DT)
        want = x.copy()
        oracle.add_source(want, src, DT)
        assert_bit_equal(s.download("u"), want, "add_source")
        assert_bit_equal(s.download("u_prev"), src, "source untouched")


@pytest.mark.parametrize("variant", VARIANTS)
@pytest.mark.parametrize("n", SIZES)
def test_jacobi_single_sweep(F, oracle, n, variant):
    rng = np.random.default_rng(100 + n)
    with F.FluidSolver(n, jacobi=variant, params={4: 0}) as s:
        for b, (alpha, beta) in ((0, (1.0, 4.0)), (1, F.coefficients(n, DT, VISC)), (2, F.coefficients(n, DT, DIFF))):
            x, x0, stale = rnd(rng, n), rnd(rng, n), rnd(rng, n)
            s.upload(u=x, v=x0, dens=stale)
            s.jacobi_sweep(b, "u", "v", "dens", alpha, beta)
            want = stale.copy()
            oracle.jacobi_sweep(b, x, x0, want, alpha, beta)
            assert_bit_equal(s.download("dens"), want, "sweep n=%d b=%d variant=%d" % (n, b, variant))
            assert_bit_equal(s.download("u"), x, "x untouched")


@pytest.mark.parametrize("variant", VARIANTS)
@pytest.mark.parametrize("n", [3, 30, 126, 257])
def test_diffuse_40(F, oracle, n, variant):
    rng = np.random.default_rng(200 + n)
    with F.FluidSolver(n, jacobi=variant, params={4: 0}) as s:
        for b, coef in ((1, VISC), (2, VISC), (0, DIFF), (0, None)):
            alpha, beta = (1.0, 4.0) if coef is None else F.coefficients(n, DT, coef)
            x, x0 = rnd(rng, n), rnd(rng, n)
            s.upload(u=x, v=x0)
            s.diffuse(b, "u", "v", alpha, beta, 40)
            want = x.copy()
            oracle.diffuse(b, want, x0, alpha, beta, 40)
            assert_bit_equal(s.download("u"), want, "diffuse n=%d b=%d" % (n, b))


def test_diffuse_rejects_odd_and_aliases(F):
    from fluidsimulationcuda_amd import capi
    with F.FluidSolver(14) as s:
        for bad in (1, 39, -2):
            with pytest.raises(capi.FluidError) as e:
                s.diffuse(0, "u", "v", 1.0, 4.0, bad)
            assert e.value.code == capi.E_INVALID
        with pytest.raises(capi.FluidError):
            s.diffuse(0, "u", "u", 1.0, 4.0, 2)
        with pytest.raises(capi.FluidError):
            s.diffuse(3, "u", "v", 1.0, 4.0, 2)
        with pytest.raises(capi.FluidError):
            s.advect(0, "u", "u", "v", "dens")
        x = np.ones((16, 16), np.float32)
        s.upload(u=x)
        s.diffuse(0, "u", "v", 1.0, 4.0, 0)          # zero sweeps: no-op
        assert_bit_equal(s.download("u"), x, "zero sweeps")


@pytest.mark.parametrize("n", SIZES)
def test_divergence_and_gradient(F, oracle, n):
    rng = np.random.default_rng(300 + n)
    with F.FluidSolver(n) as s:
        u, v, p, d = rnd(rng, n), rnd(rng, n), rnd(rng, n), rnd(rng, n)
        s.upload(u=u, v=v, u_prev=p, v_prev=d)
        s.computeDivergenceAndPressure("u", "v", "u_prev", "v_prev")
        oracle.divergence(u, v, p, d)
        assert_bit_equal(s.download("u_prev"), p, "p cleared incl. ghosts")
        assert_bit_equal(s.download("v_prev"), d, "div")
        p = rnd(rng, n)
        s.upload(u_prev=p)
        s.lastProject("u", "v", "u_prev")
        oracle.subtract_gradient(u, v, p)
        assert_bit_equal(s.download("u"), u, "u - grad p")
        assert_bit_equal(s.download("v"), v, "v - grad p")


@pytest.mark.parametrize("amp", [0.0, 0.01, 1.0, 250.0])
@pytest.mark.parametrize("n", [1, 2, 5, 30, 61, 256, 1022])
def test_advect(F, oracle, n, amp):
    """amp=0: identity on the interior; amp=250: dt0*|vel| spans the grid, all
    four clamps hit (FluidSequential.c:117-127)."""
    rng = np.random.default_rng(400 + n)
    u, v, d0 = rnd(rng, n, -amp, amp), rnd(rng, n, -amp, amp), rnd(rng, n)
    with F.FluidSolver(n) as s:
        for b in (0, 1, 2):
            s.upload(u=u, v=v, dens_prev=d0, dens=rnd(rng, n))
            s.advect(b, "dens", "dens_prev", "u", "v", DT)
            want = np.zeros_like(d0)
            oracle.advect(b, want, d0, u, v, DT)
            got = s.download("dens")
            assert_bit_equal(got, want, "advect n=%d b=%d amp=%g" % (n, b, amp))
            if amp == 0.0:
                assert_bit_equal(got[1:-1, 1:-1], d0[1:-1, 1:-1], "zero velocity => identity")
        # self-advection as in vel_step (FluidSequential.c:232,237): d0 aliases u / v
        s.upload(u_prev=u, v_prev=v)
        s.advect(1, "u", "u_prev", "u_prev", "v_prev", DT)
        s.advect(2, "v", "v_prev", "u_prev", "v_prev", DT)
        wu, wv = np.zeros_like(u), np.zeros_like(u)
        oracle.advect(1, wu, u, u, v, DT)
        oracle.advect(2, wv, v, u, v, DT)
        assert_bit_equal(s.download("u"), wu, "self-advect u")
        assert_bit_equal(s.download("v"), wv, "self-advect v")


@pytest.mark.parametrize("n", [14, 30, 61])
def test_golden_operators(F, n):
    """The same vectors the oracle is pinned with, straight against the GPU."""
    g = load_golden("ops_n%d.npz" % n)
    with F.FluidSolver(n) as s:
        for b in (0, 1, 2):
            s.upload(u=g["bnd_in"])
            s.set_bnd(b, "u")
            assert_bit_equal(s.download("u"), g["bnd_out_b%d" % b], "golden set_bnd")
        s.upload(u=g["src_x"], v=g["src_s"])
        s.add_source("u", "v", DT)
        assert_bit_equal(s.download("u"), g["src_out"], "golden add_source")
        for k, (b, a, be) in enumerate(g["dif_params"]):
            for iters in (2, 40):
                for variant in VARIANTS:
                    s.set_jacobi_variant(variant)
                    s.upload(u=g["dif%d_x" % k], v=g["dif%d_x0" % k])
                    s.diffuse(int(b), "u", "v", float(np.float32(a)), float(np.float32(be)), iters)
                    assert_bit_equal(s.download("u"), g["dif%d_out%d" % (k, iters)], "golden diffuse %d/%d" % (k, iters))
        s.set_jacobi_variant(0)
        s.upload(u=g["div_u"], v=g["div_v"], u_prev=g["div_p_in"], v_prev=g["div_div_in"])
        s.computeDivergenceAndPressure("u", "v", "u_prev", "v_prev")
        assert_bit_equal(s.download("u_prev"), g["div_p"], "golden p")
        assert_bit_equal(s.download("v_prev"), g["div_div"], "golden div")
        s.upload(u=g["grad_u"], v=g["grad_v"], u_prev=g["grad_p"])
        s.lastProject("u", "v", "u_prev")
        assert_bit_equal(s.download("u"), g["grad_u_out"], "golden grad u")
        assert_bit_equal(s.download("v"), g["grad_v_out"], "golden grad v")
        for tag in ("small", "clamp"):
            s.upload(u=g["adv_%s_u" % tag], v=g["adv_%s_v" % tag], dens_prev=g["adv_%s_d0" % tag])
            for b in (0, 1, 2):
                s.advect(b, "dens", "dens_prev", "u", "v", DT)
                assert_bit_equal(s.download("dens"), g["adv_%s_out_b%d" % (tag, b)], "golden advect")
            s.upload(u_prev=g["adv_%s_u" % tag], v_prev=g["adv_%s_v" % tag])
            s.advect(1, "u", "u_prev", "u_prev", "v_prev", DT)
            assert_bit_equal(s.download("u"), g["adv_%s_self_u" % tag], "golden self-advect u")
            s.upload(u=g["adv_%s_u" % tag])
            s.advect(2, "v", "v_prev", "u_prev", "v_prev", DT)
            assert_bit_equal(s.download("v"), g["adv_%s_self_v" % tag], "golden self-advect v")


def test_reductions(F, oracle):
    n = 300
    rng = np.random.default_rng(9)
    u, v = rnd(rng, n, -3, 3), rnd(rng, n, -5, 5)
    with F.FluidSolver(n) as s:
        s.upload(u=u, v=v)
        want = max(np.abs(u[1:-1, 1:-1]).max(), np.abs(v[1:-1, 1:-1]).max())
        assert s.absmax_velocity("u", "v") == want
        # residual is a diagnostic: compare against a float64 evaluation loosely,
        # and check it falls as the solve proceeds and leaves the fields alone
        x, x0 = rnd(rng, n), rnd(rng, n)
        s.upload(u=x, v=x0)
        r0 = s.residual("u", "v", 1.0, 4.0)
        nb = x[1:-1, :-2] + x[1:-1, 2:] + x[:-2, 1:-1] + x[2:, 1:-1]
        ref = np.abs(4.0 * x[1:-1, 1:-1].astype(np.float64) - nb - x0[1:-1, 1:-1]).max()
        assert abs(r0 - ref) <= 1e-5 * ref
        assert_bit_equal(s.download("u"), x, "residual must not alter x")
        s.diffuse(0, "u", "v", 1.0, 4.0, 40)
        assert s.residual("u", "v", 1.0, 4.0) < r0


# ---- temporally blocked kernel: window / strip / wall edge cases ---------------
# A window is one wave: 64 lanes of 2 or 4 columns, overlapping its neighbours by ceil(T/cols) lanes
# per side.  Owned columns per window: 4-column lanes 240 (T=8), 248 (T=4, 2); 2-column lanes 96
# (T=16), 112 (T=8), 120 (T=4), 124 (T=2) -> sizes straddling those multiples; strips of `rows`
# output rows overlap by T rows: tiny, ragged and huge strips.
TB_SIZES = [1, 2, 3, 4, 5, 7, 8, 9, 14, 61, 64, 95, 96, 97, 111, 112, 113, 119, 120, 121, 123, 124, 125, 191, 192, 193,
            223, 224, 225, 239, 240, 241, 247, 248, 249, 255, 256, 257, 480, 481, 1022]


@pytest.mark.parametrize("lane_cols,max_t,fast_div", [(2, 16, 1), (2, 16, 2), (2, 16, 3), (2, 16, 0), (2, 12, 2), (2, 12, 3), (2, 12, 0), (2, 8, 1), (2, 8, 2),
                                                      (2, 8, 3), (2, 4, 2), (2, 2, 1), (4, 8, 1), (4, 8, 2), (4, 4, 2), (4, 2, 2), (4, 2, 3)])
@pytest.mark.parametrize("n", TB_SIZES)
def test_temporal_blocking_matches_oracle(F, oracle, n, lane_cols, max_t, fast_div):
    """Every depth of the fused kernel, both forms (pressure: alpha 1, beta 4; general: the exact
    reciprocal divisions of fast_div = 1 / 2 / 3 -- two-term, scaled residual correction (the default), double-precision
    reciprocal -- or true division with fast_div = 0), on window / strip / wall edge sizes.
    The launch count is asserted so that a silently shallower schedule fails: 16-sweep launches are
    normally reserved for grids of 8 M cells and more (PARAM_TB_T16_MIN_CELLS forces them here)."""
    from fluidsimulationcuda_amd import capi
    rng = np.random.default_rng(500 + n)
    with F.FluidSolver(n, jacobi=capi.JACOBI_TB) as s:
        s.set_param(capi.PARAM_TB_MIN_CELLS, 0)              # fuse sweeps even on these small grids
        s.set_param(capi.PARAM_TB_T16_MIN_CELLS, 0)          # ... 16 at a time where max_t allows
        s.set_param(capi.PARAM_TB_LANE_COLUMNS, lane_cols)
        s.set_param(capi.PARAM_TB_MAX_SWEEPS, max_t)
        s.set_param(capi.PARAM_TB_FAST_DIVISION, fast_div)
        s.timing_enable(True)
        for rows in (0, 1, 3, 16, 5000):
            s.set_param(capi.PARAM_TB_ROWS, rows)
            for b, (alpha, beta), iters in ((0, (1.0, 4.0), 40), (1, F.coefficients(n, DT, VISC), 22),
                                            (2, F.coefficients(n, DT, DIFF), 6), (0, (0.7, 3.3), 8),
                                            (2, F.coefficients(n, DT, DIFF), 32)):
                x, x0 = rnd(rng, n), rnd(rng, n)
                s.upload(u=x, v=x0)
                s.timing_read(reset=True)
                s.diffuse(b, "u", "v", alpha, beta, iters)
                t = s.timing_read(reset=True)
                plan = tb_schedule(iters, max_t, deep=lane_cols == 2)
                assert t["jacobi_launches"] == len(plan) and t["sweeps"] == iters, \
                    "schedule %r expected for %d sweeps at max_t=%d, got %d launches" % (plan, iters, max_t, t["jacobi_launches"])
                want = x.copy()
                oracle.diffuse(b, want, x0, alpha, beta, iters)
                assert_bit_equal(s.download("u"), want,
                                 "TB n=%d cols=%d maxT=%d rows=%d b=%d iters=%d" % (n, lane_cols, max_t, rows, b, iters))
                assert_bit_equal(s.download("v"), x0, "x0 untouched")


def test_temporal_blocking_power_of_two_beta_paths(F, oracle):
    """beta = 2^k takes the multiply-by-reciprocal path; it must agree with true
    division on every input class, denormal results included."""
    from fluidsimulationcuda_amd import capi
    n = 61
    rng = np.random.default_rng(77)
    with F.FluidSolver(n, jacobi=capi.JACOBI_TB) as s:
        s.set_param(capi.PARAM_TB_MIN_CELLS, 0)
        for beta in (4.0, 0.5, 1.0, 1024.0, 2.0 ** -20, 2.0 ** 100):
            for scale in (1.0, 1e-38, 1e30):
                x, x0 = rnd(rng, n) * np.float32(scale), rnd(rng, n) * np.float32(scale)
                s.upload(u=x, v=x0)
                s.diffuse(0, "u", "v", 1.0, beta, 8)
                want = x.copy()
                oracle.diffuse(0, want, x0, 1.0, beta, 8)
                assert_bit_equal(s.download("u"), want, "beta=%g scale=%g" % (beta, scale))


@pytest.mark.parametrize("beta", [3.0, 6.0, 12.0, 10.0, 1.00016, 102.606407, 2682.734, 0.75, 3.3, 5e-5, 7e5])
def test_temporal_blocking_reciprocal_division_is_exact(F, oracle, beta):
    """The TB kernel may replace x/beta by an equivalent reciprocal form -- Markstein's residual correction with the
    residual scaled out of the underflow range (the default), (float)((double)x * (1/beta)), the two-term reciprocal --
    once the library has proven the two equal for all 2^32 inputs on the device (or fall
    back to dividing).  Either way the bits must match the oracle's true
    division, on ordinary data and on data full of zeros of both signs,
    denormals and values near overflow (3e38 is past the 2^104 the scaled form covers: those waves
    must notice the inf / NaN they produce and run again with the double form)."""
    from fluidsimulationcuda_amd import capi
    n = 61
    rng = np.random.default_rng(int(beta * 1000) % 2 ** 31)
    specials = np.array([0.0, -0.0, 1e-45, -1e-45, 1e-39, -3e-39, 1.17549435e-38, 3e38, -3e38, 1e-30, 6e-45, 9e-45],
                        dtype=np.float32)
    with F.FluidSolver(n, jacobi=capi.JACOBI_TB) as s:
        s.set_param(capi.PARAM_TB_MIN_CELLS, 0)
        for fast in (1, 2, 3, 0):    # two-term reciprocal where x0 allows / scaled residual correction / double reciprocal / true division
            s.set_param(capi.PARAM_TB_FAST_DIVISION, fast)
            for kind in ("uniform", "special", "tiny"):
                if kind == "uniform":
                    x, x0 = rnd(rng, n), rnd(rng, n)
                elif kind == "special":
                    x = rng.choice(specials, size=(n + 2, n + 2)).astype(np.float32)
                    x0 = rng.choice(specials, size=(n + 2, n + 2)).astype(np.float32)
                else:
                    x = (rnd(rng, n) * np.float32(1e-38)).astype(np.float32)
                    x0 = (rnd(rng, n) * np.float32(1e-41)).astype(np.float32)
                for alpha in (1.0, 0.37):
                    s.upload(u=x, v=x0)
                    s.diffuse(0, "u", "v", alpha, beta, 8)
                    want = x.copy()
                    with np.errstate(all="ignore"):
                        oracle.diffuse(0, want, x0, alpha, beta, 8)
                    got = s.download("u")
                    nan = np.isnan(want)
                    assert np.array_equal(np.isnan(got), nan)
                    assert_bit_equal(np.where(nan, 0, got), np.where(nan, 0, want),
                                     "beta=%g fast=%d %s alpha=%g" % (beta, fast, kind, alpha))


@pytest.mark.parametrize("fast", [1, 2])
@pytest.mark.parametrize("max_t", [16, 12, 8, 2])
@pytest.mark.parametrize("n", [97, 300, 1022])
def test_two_term_division_only_where_the_right_hand_side_allows_it(F, oracle, n, max_t, fast):
    """Division mode 3 (two packed float instructions) is exact for dividends that are zero or at least
    beta * 2^-98; the kernel uses it in a wave only if |x0| >= beta * 2^-72 on every tile the wave touches,
    which bounds every dividend of every sweep from below (fluid_kernels.hip, DIVMODE 3).  Ordinary values
    with islands of exact zeros (both signs), of tiny values (2^-149 .. 2^-60) and of huge ones, in x0 and in
    the first guess: every cell must still carry the oracle's bits.
    fast = 2: the same fields through division mode 5 (scaled residual correction), which needs no fact about the data
    except that a wave whose stored rows hold inf or NaN (the huge islands: 1e37 * beta-sized sums overflow, and past 2^104
    the scaled residual does) must run again in the double form; islands of inf and NaN themselves are added there."""
    from fluidsimulationcuda_amd import capi
    rng = np.random.default_rng(n + max_t)
    alpha, beta = F.coefficients(n, DT, VISC)

    def field(islands):
        f = rnd(rng, n)
        for _ in range(islands):
            i, j = rng.integers(1, n, 2)
            h, w = rng.integers(1, 40, 2)
            kind = rng.integers(0, 4 if fast == 1 else 6)
            blk = f[i:i + h, j:j + w]
            if kind == 0:
                blk[...] = np.where(rng.random(blk.shape) < 0.5, 0.0, -0.0)
            elif kind == 1:
                blk[...] = (blk * np.float32(2.0) ** rng.integers(-149, -60, blk.shape)).astype(np.float32)
            elif kind == 2:
                blk[...] = blk * np.float32(1e-37)
            elif kind == 3:
                blk[...] = blk * np.float32(1e37)
            elif kind == 4:
                blk[...] = blk * np.float32(2.0) ** rng.integers(100, 108, blk.shape)       # around the scaled form's 2^104
            else:
                blk[...] = rng.choice(np.array([np.inf, -np.inf, np.nan, 1.0], np.float32), blk.shape)
        return f

    with F.FluidSolver(n, params={capi.PARAM_TB_MIN_CELLS: 0, capi.PARAM_TB_T16_MIN_CELLS: 0,
                                  capi.PARAM_TB_MAX_SWEEPS: max_t, capi.PARAM_TB_FAST_DIVISION: fast}) as s:
        assert s.division_mode(alpha, beta) == (3 if fast == 1 else 5) and s.division_mode(1.0, 4.0) == 4
        for b, islands in ((0, 6), (1, 0), (2, 2)):
            x, x0 = field(6), field(islands)
            s.upload(u=x, v=x0)
            s.diffuse(b, "u", "v", alpha, beta, 16)
            want = x.copy()
            with np.errstate(all="ignore"):
                oracle.diffuse(b, want, x0, alpha, beta, 16)
            got, nan = s.download("u"), np.isnan(want)       # inf - inf next to the huge islands: any NaN is a NaN
            assert np.array_equal(np.isnan(got), nan)
            assert_bit_equal(np.where(nan, 0, got), np.where(nan, 0, want), "reciprocal division (fast=%d) n=%d maxT=%d b=%d" % (fast, n, max_t, b))


@pytest.mark.parametrize("n", [300, 1022])
def test_two_term_division_worst_case_cancellation(F, oracle, n):
    """The lower bound on the dividends is what makes mode 3 safe without a per-value check: x0 >= m does not keep
    x0 + alpha*nb away from zero, only away from (0, 2^-25 m).  First guesses built to cancel the right-hand
    side as closely as floats allow (x = -x0 / (4 alpha) and neighbours of it one ulp apart) drive the dividends
    of the first sweeps to their smallest possible nonzero magnitudes; x0 itself sits just above the tile
    threshold beta * 2^-72."""
    from fluidsimulationcuda_amd import capi
    rng = np.random.default_rng(n)
    alpha, beta = F.coefficients(n, DT, VISC)
    for scale in (np.float32(beta) * np.float32(2.0 ** -71), np.float32(1.0)):
        x0 = (rnd(rng, n, 1.0, 2.0) * scale * rng.choice(np.array([-1, 1], np.float32), (n + 2, n + 2))).astype(np.float32)
        x = (-x0 / (np.float32(4) * np.float32(alpha))).astype(np.float32)
        x = np.nextafter(x, np.float32(np.inf) * rng.choice(np.array([-1, 1], np.float32), x.shape)).astype(np.float32)
        # smooth x0 so that the four neighbours of a cell cancel it too
        x0[...] = x0[n // 2, n // 2]
        x[...] = np.where(rng.random(x.shape) < 0.5, x[n // 2, n // 2], np.nextafter(x[n // 2, n // 2], np.float32(0)))
        with F.FluidSolver(n, params={capi.PARAM_TB_MIN_CELLS: 0, capi.PARAM_TB_FAST_DIVISION: 1}) as s:
            assert s.division_mode(alpha, beta) == 3
            s.upload(u=x, v=x0)
            s.diffuse(0, "u", "v", alpha, beta, 8)
            want = x.copy()
            oracle.diffuse(0, want, x0, alpha, beta, 8)
            assert_bit_equal(s.download("u"), want, "cancellation, scale %g" % scale)


def test_exact_cancellations_and_signed_zeros(F, oracle):
    """Random floats almost never cancel exactly, so they never show whether
    -0 / +0 come out as the reference's expressions produce them (x - x = +0,
    negative * +0 = -0, -(+0) ghosts ...).  Fields drawn from a few dyadic values
    and both zeros make such cases the norm; every operator must still match
    bit for bit."""
    from fluidsimulationcuda_amd import capi
    n = 256
    rng = np.random.default_rng(0)
    vals = np.array([-1, -0.5, -0.25, 0.0, -0.0, 0.25, 0.5, 1], np.float32)

    def q():
        return rng.choice(vals, size=(n + 2, n + 2)).astype(np.float32)

    with F.FluidSolver(n, params={capi.PARAM_TB_MIN_CELLS: 0}) as s:
        u, v, p, d = q(), q(), q(), q()
        s.upload(u=u, v=v, u_prev=p, v_prev=d)
        s.computeDivergenceAndPressure("u", "v", "u_prev", "v_prev")
        oracle.divergence(u, v, p, d)
        assert_bit_equal(s.download("v_prev"), d, "divergence")
        p = q()
        s.upload(u_prev=p)
        s.lastProject("u", "v", "u_prev")
        oracle.subtract_gradient(u, v, p)
        assert_bit_equal(s.download("u"), u, "gradient u")
        assert_bit_equal(s.download("v"), v, "gradient v")
        x, src = q(), q()
        for dt in (0.5, -0.5, 0.0):
            s.upload(u=x, v=src)
            s.add_source("u", "v", dt)
            w = x.copy()
            oracle.add_source(w, src, dt)
            assert_bit_equal(s.download("u"), w, "add_source dt=%g" % dt)
        for variant in (0, 1, 2, 3):
            s.set_jacobi_variant(variant)
            for b in (0, 1, 2):
                for alpha, beta in ((1.0, 4.0), (0.5, 3.0), (0.25, 2.0)):
                    x, x0 = q(), q()
                    s.upload(u=x, v=x0)
                    s.diffuse(b, "u", "v", alpha, beta, 8)
                    w = x.copy()
                    oracle.diffuse(b, w, x0, alpha, beta, 8)
                    assert_bit_equal(s.download("u"), w, "diffuse variant %d b=%d beta=%g" % (variant, b, beta))
        uu, vv, d0 = q(), q(), q()
        s.upload(u=uu, v=vv, dens_prev=d0)
        for b in (0, 1, 2):
            s.advect(b, "dens", "dens_prev", "u", "v", DT)
            w = np.zeros_like(d0)
            oracle.advect(b, w, d0, uu, vv, DT)
            assert_bit_equal(s.download("dens"), w, "advect b=%d" % b)
        for b in (0, 1, 2):
            x = q()
            s.upload(u=x)
            s.set_bnd(b, "u")
            w = x.copy()
            oracle.set_bnd(b, w)
            assert_bit_equal(s.download("u"), w, "set_bnd b=%d" % b)
        # a whole step from such fields (zero sources: the x + dt*0 path keeps -0 where the reference does)
        f = dict(u=q(), v=q(), dens=q())
        s.upload(**f)
        s.step(2)
        z = np.zeros((n + 2, n + 2), np.float32)
        uu, vv, dd, a0, b0, c0 = f["u"].copy(), f["v"].copy(), f["dens"].copy(), z.copy(), z.copy(), z.copy()
        oracle.step(uu, vv, dd, a0, b0, c0)
        oracle.step(uu, vv, dd, a0, b0, c0)
        for name, want in (("u", uu), ("v", vv), ("dens", dd)):
            assert_bit_equal(s.download(name), want, "two steps from coarse fields: " + name)
