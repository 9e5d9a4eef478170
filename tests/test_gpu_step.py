"""GPU: whole vel_step + dens_step through the C ABI against the golden
snapshots from the compiled reference, the oracle at mid sizes, the reference's
checksums at 1022/4094, and size-independent properties at full size."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, assert_bit_equal, assert_close_fp32, load_golden, rnd, tb_schedule

pytestmark = pytest.mark.gpu
DT, VISC, DIFF = 0.016, 0.0025, 0.1


@pytest.fixture(scope="module")
def F():
    import fluidsimulationcuda_amd as F
    return F


@pytest.mark.parametrize("variant", [0, 1, 2, 3])
@pytest.mark.parametrize("n,iters,steps", [(30, 40, (1, 2, 5)), (61, 40, (1, 2, 5)), (126, 40, (1, 2, 5)),
                                           (126, 20, (1, 2))])
def test_golden_steps(F, n, iters, steps, variant):
    g = load_golden("step_n%d_k%d.npz" % (n, iters))
    z0 = np.zeros((n + 2, n + 2), np.float32)
    with F.FluidSolver(n, jacobi=variant, params={4: 0}) as s:
        s.upload(u=z0, v=z0, dens=z0, u_prev=g["init_u_prev"], v_prev=g["init_v_prev"], dens_prev=g["init_dens_prev"])
        for z in range(1, max(steps) + 1):
            s.step(1, use_sources=(z == 1), iters=iters)
            if z == 1:
                assert_bit_equal(s.download("u_prev"), g["s1_u_prev"], "pressure in u_prev")
                assert_bit_equal(s.download("v_prev"), g["s1_v_prev"], "divergence in v_prev")
                assert_bit_equal(s.download("dens_prev"), g["s1_dens_prev"], "diffused density in dens_prev")
            if z in steps:
                for name in ("u", "v", "dens"):
                    got, want = s.download(name), g["s%d_%s" % (z, name)]
                    assert_close_fp32(got, want, "%s step %d (contract: 1e-5 rel)" % (name, z))
                    assert_bit_equal(got, want, "%s after step %d, n=%d" % (name, z, n))


@pytest.mark.parametrize("max_t", [16, 12, 8])
@pytest.mark.parametrize("n", [1, 2, 3, 14, 61, 95, 96, 97, 126, 193, 257, 300, 1022])
def test_divergence_inside_the_pressure_solve_matches_oracle(F, oracle, n, max_t):
    """On one GPU a projection's divergence (FluidSequential.c:143-158) is computed by the first launch of the
    pressure solve that consumes it, row by row, and stored with its ghost cells (it is the step's v_prev).  Window
    and wall edge sizes, every first-launch depth, strip heights that put ghost rows in their own strips; u, v
    with both zeros and exact cancellations; against the oracle, all six fields."""
    from fluidsimulationcuda_amd import capi
    rng = np.random.default_rng(n * 31 + max_t)
    vals = np.array([-1, -0.5, -0.25, 0.0, -0.0, 0.25, 0.5, 1], np.float32)
    for rows, coarse in ((0, False), (3, True), (40, False)):
        fields = [rng.choice(vals, size=(n + 2, n + 2)).astype(np.float32) if coarse else rnd(rng, n) for _ in range(6)]
        u, v, dens, u0, v0, d0 = fields
        params = {capi.PARAM_TB_MIN_CELLS: 0, capi.PARAM_TB_T16_MIN_CELLS: 0, capi.PARAM_TB_MAX_SWEEPS: max_t,
                  capi.PARAM_TB_ROWS: rows}
        with F.FluidSolver(n, params=params) as s:
            s.upload(u=u, v=v, dens=dens, u_prev=u0, v_prev=v0, dens_prev=d0)
            s.timing_enable(True)
            s.step(1, use_sources=True)
            s.vel_step()
            t = s.timing_read()
            assert t["divergence_calls"] == 0, "the divergence was meant to ride in the pressure solves"
            oracle.step_src(u, v, dens, u0, v0, d0)
            oracle.vel_step(u, v, u0, v0)
            for name, want in (("u", u), ("v", v), ("dens", dens), ("u_prev", u0), ("v_prev", v0), ("dens_prev", d0)):
                assert_bit_equal(s.download(name), want, "%s n=%d maxT=%d rows=%d" % (name, n, max_t, rows))


def test_multi_step_call_equals_single_steps(F):
    g = load_golden("step_n61_k40.npz")
    z0 = np.zeros((63, 63), np.float32)
    with F.FluidSolver(61) as s:
        s.upload(u=z0, v=z0, dens=z0, u_prev=g["init_u_prev"], v_prev=g["init_v_prev"], dens_prev=g["init_dens_prev"])
        s.step(5, use_sources=True)
        for name in ("u", "v", "dens"):
            assert_bit_equal(s.download(name), g["s5_%s" % name], name)


def test_host_array_drop_in(F):
    """step()/step_src(): the C ABI on caller-owned host arrays
    (FluidSequential.c:298-306), including a change of N between calls."""
    for n in (30, 126):
        g = load_golden("step_n%d_k40.npz" % n)
        u, v, d = (np.zeros((n + 2, n + 2), np.float32) for _ in range(3))
        u0, v0, d0 = g["init_u_prev"].copy(), g["init_v_prev"].copy(), g["init_dens_prev"].copy()
        F.step_src(n, DT, DIFF, VISC, 40, u, v, d, u0, v0, d0)
        assert_bit_equal(u, g["s1_u"], "step_src u")
        assert_bit_equal(d, g["s1_dens"], "step_src dens")
        assert_bit_equal(u0, g["s1_u_prev"], "step_src leaves p in u_prev")
        F.step(n, DT, DIFF, VISC, u, v, d)
        assert_bit_equal(u, g["s2_u"], "step u")
        assert_bit_equal(v, g["s2_v"], "step v")
        assert_bit_equal(d, g["s2_dens"], "step dens")
    from fluidsimulationcuda_amd import capi
    with pytest.raises(capi.FluidError):
        F.step_src(30, DT, DIFF, VISC, 7, *(np.zeros((32, 32), np.float32) for _ in range(6)))
    capi.check(capi.lib().fluid_release_cached())


@pytest.mark.parametrize("n", [254, 1022])
def test_step_vs_oracle(F, oracle, n):
    dens, dens0, u, u0, v, v0 = oracle.initialize_portable(n, seed=n)
    with F.FluidSolver(n) as s:
        s.upload(u=u, v=v, dens=dens, u_prev=u0, v_prev=v0, dens_prev=dens0)
        s.step(1, use_sources=True)
        oracle.step_src(u, v, dens, u0, v0, dens0)
        s.step(1)
        oracle.step(u, v, dens, u0, v0, dens0)
        for name, want in (("u", u), ("v", v), ("dens", dens), ("u_prev", u0), ("v_prev", v0), ("dens_prev", dens0)):
            assert_bit_equal(s.download(name), want, "%s n=%d" % (name, n))


@pytest.mark.parametrize("fast_div", [2, 1, 3])
@pytest.mark.parametrize("lane_cols,max_t", [(2, 16), (2, 12), (2, 8), (4, 8), (4, 2)])
@pytest.mark.parametrize("n", [61, 126, 300])
def test_steps_through_the_fused_kernel_on_signed_zero_fields(F, oracle, n, lane_cols, max_t, fast_div):
    """Whole steps with the fused Jacobi kernel forced on (it is the default only on large grids), on
    fields drawn from a few dyadic values and both zeros: exact cancellations and -0 are the norm
    there, which is what shows whether the add_source of the zeroed sources (x + dt*0 turns -0 into
    +0; the library applies it inside the solve instead of in a pass of its own), the paired
    advection of u and v and the ghost-cell sign flips reproduce the reference's bits."""
    from fluidsimulationcuda_amd import capi
    rng = np.random.default_rng(n + lane_cols)
    vals = np.array([-1, -0.5, -0.25, 0.0, -0.0, 0.25, 0.5, 1], np.float32)
    u, v, dens, u0, v0, dens0 = (rng.choice(vals, size=(n + 2, n + 2)).astype(np.float32) for _ in range(6))
    params = {capi.PARAM_TB_MIN_CELLS: 0, capi.PARAM_TB_LANE_COLUMNS: lane_cols, capi.PARAM_TB_MAX_SWEEPS: max_t,
              capi.PARAM_TB_T16_MIN_CELLS: 0, capi.PARAM_TB_FAST_DIVISION: fast_div}
    per_solve = len(tb_schedule(40, max_t, deep=lane_cols == 2))
    with F.FluidSolver(n, params=params) as s:
        s.upload(u=u, v=v, dens=dens, u_prev=u0, v_prev=v0, dens_prev=dens0)
        s.timing_enable(True)
        s.timing_read(reset=True)
        s.step(1, use_sources=True)
        t = s.timing_read(reset=True)
        s.timing_enable(False)
        # the sourced step: add_source (FluidSequential.c:78-82) runs inside the first launch of the batched diffusion --
        # no k_add_source launch at all -- wherever that launch exists with the second store (2-column lanes, 8+ sweeps,
        # not the tile-guarded division); else three passes of their own.  The same launches and sweeps either way.
        fused = lane_cols == 2 and fast_div != 1
        assert t["source_calls"] == (0 if fused else 3), t
        assert t["jacobi_launches"] == 3 * per_solve and t["sweeps"] == 200, t
        oracle.step_src(u, v, dens, u0, v0, dens0)
        for name, want in (("u", u), ("v", v), ("dens", dens), ("u_prev", u0), ("v_prev", v0), ("dens_prev", dens0)):
            assert_bit_equal(s.download(name), want, "%s n=%d cols=%d maxT=%d sourced step" % (name, n, lane_cols, max_t))
        for z in range(2):
            s.timing_enable(True)
            s.timing_read(reset=True)
            s.step(1)
            t = s.timing_read(reset=True)
            s.timing_enable(False)
            # the three diffusions share their launches; each projection's solve has its own
            assert t["jacobi_launches"] == 3 * per_solve and t["sweeps"] == 200, t
            assert t["source_calls"] == 0, "the add_source of a zeroed source stays pending on the field: no kernel"
            oracle.step(u, v, dens, u0, v0, dens0)
            for name, want in (("u", u), ("v", v), ("dens", dens), ("u_prev", u0), ("v_prev", v0), ("dens_prev", dens0)):
                assert_bit_equal(s.download(name), want, "%s n=%d cols=%d maxT=%d step %d" % (name, n, lane_cols, max_t, z + 2))
        # single operators after a step see the settled fields too
        s.vel_step()
        oracle.vel_step(u, v, u0, v0)
        assert_bit_equal(s.download("u"), u, "vel_step u")
        assert_bit_equal(s.download("v_prev"), v0, "vel_step leaves the divergence in v_prev")


@pytest.mark.parametrize("storage", [0, 1])
@pytest.mark.parametrize("n", [97, 510, 1022])
def test_add_source_inside_the_first_diffusion_launch(F, oracle, n, storage):
    """FLUID_PARAM_FUSE_ADD_SOURCE: x + dt*s formed by the first launch of the diffusion (first guess = s, right-hand side
    = the sum; FluidSequential.c:78-82, :181, :201, :209) and stored out of place, against the same steps with k_add_source as a
    pass of its own -- all six fields bit for bit, three sourced steps in a row (the sum lands in a scratch buffer that
    trades places with the field, so the second and third step run on swapped buffers), fp32 and fp16 storage, and for
    fp32 against the oracle; then through fluid_vel_step / fluid_dens_step, and a sourced step whose first launch is a
    shallow one (6 sweeps per solve: no second store exists for it, so the source is settled by its own kernel)."""
    from fluidsimulationcuda_amd import capi
    rng = np.random.default_rng(n + storage)
    f0 = {k: rnd(rng, n) for k in ("u", "v", "dens")}
    srcs = [{k: rnd(rng, n) for k in ("u_prev", "v_prev", "dens_prev")} for _ in range(3)]
    names = ("u", "v", "dens", "u_prev", "v_prev", "dens_prev")
    got = {}
    for fuse in (1, 0):
        with F.FluidSolver(n, storage=storage, params={capi.PARAM_TB_MIN_CELLS: 0, capi.PARAM_TB_T16_MIN_CELLS: 0,
                                                       capi.PARAM_FUSE_ADD_SOURCE: fuse}) as s:
            s.upload(**f0)
            s.timing_enable(True)
            out = []
            for src in srcs:
                s.upload(**src)
                s.timing_read(reset=True)
                s.step(1, use_sources=True)
                t = s.timing_read(reset=True)
                assert t["source_calls"] == (0 if fuse else 3), t
                out.append({k: s.download(k) for k in names})
            s.upload(**srcs[0])
            s.vel_step()
            s.dens_step()
            out.append({k: s.download(k) for k in names})
            s.upload(**srcs[1])
            s.timing_read(reset=True)
            s.step(1, use_sources=True, iters=6)
            assert s.timing_read(reset=True)["source_calls"] == 3
            out.append({k: s.download(k) for k in names})
            got[fuse] = out
    for z, (a, b) in enumerate(zip(got[1], got[0])):
        for k in names:
            assert_bit_equal(a[k], b[k], "%s after call %d, n=%d storage=%d: fused add_source vs its own kernel" % (k, z, n, storage))
    if storage == 0:
        u, v, dens = (f0[k].copy() for k in ("u", "v", "dens"))
        for z, src in enumerate(srcs):
            u0, v0, d0 = (src[k].copy() for k in ("u_prev", "v_prev", "dens_prev"))
            oracle.step_src(u, v, dens, u0, v0, d0)
            for k, want in zip(names, (u, v, dens, u0, v0, d0)):
                assert_bit_equal(got[1][z][k], want, "%s after sourced step %d vs the oracle, n=%d" % (k, z, n))


@pytest.mark.parametrize("fast_div", [2, 1, 3])
@pytest.mark.parametrize("n", [254, 510])
def test_decay_through_the_denormal_range_matches_oracle(F, oracle, n, fast_div):
    """With the sources zeroed after step 0 (FluidSequential.c:298-302) every solve restarts from a zero
    first guess and the fields shrink by orders of magnitude per step: within a few steps they hold
    tiny (< 2^-100) and denormal values, and die out entirely after ~20.  Re-inject sources every
    four steps so the run keeps crossing that whole range, and demand the oracle's bits throughout
    (the fused kernel's division shortcuts are exact there only because they were chosen to be)."""
    from fluidsimulationcuda_amd import capi
    from fluidsimulationcuda_amd.harness import initialize_parameters
    z = np.zeros((n + 2, n + 2), np.float32)
    u, v, dens = z.copy(), z.copy(), z.copy()
    tiny_seen = denormal_seen = 0
    with F.FluidSolver(n, params={capi.PARAM_TB_MIN_CELLS: 0, capi.PARAM_TB_FAST_DIVISION: fast_div}) as s:
        s.upload(u=u, v=v, dens=dens)
        for k in range(3):
            f = initialize_parameters(n, seed=20 + k)
            u0, v0, d0 = f["u_prev"].copy(), f["v_prev"].copy(), f["dens_prev"].copy()
            s.upload(u_prev=u0, v_prev=v0, dens_prev=d0)
            s.step(1, use_sources=True)
            oracle.step_src(u, v, dens, u0, v0, d0)
            for _ in range(3):
                s.step(1)
                oracle.step(u, v, dens, u0, v0, d0)
            for name, want in (("u", u), ("v", v), ("dens", dens), ("u_prev", u0), ("v_prev", v0), ("dens_prev", d0)):
                assert_bit_equal(s.download(name), want, "%s after round %d, n=%d" % (name, k, n))
            a = np.abs(dens)
            tiny_seen += int(((a < 2.0 ** -100) & (a > 0)).sum())
            denormal_seen += int(((a < 2.0 ** -126) & (a > 0)).sum())
    assert tiny_seen > 0 and denormal_seen > 0, "the run was meant to cross the tiny and denormal ranges"


@pytest.mark.parametrize("n", [1022, 4094])
def test_reference_checksums(F, oracle, n):
    """Step 1 from the reference's own initializeParameters (glibc rand seed 1):
    sums, centre values and CRC-32 of the reference's output bytes (checksums.json;
    test_gpu_large.py repeats the CRC at 8190)."""
    import zlib
    row = [r for r in json.load(open(os.path.join(GOLDEN, "checksums.json"))) if r["n"] == n][0]
    dens, dens0, u, u0, v, v0 = oracle.initialize_glibc(n, seed=1)
    with F.FluidSolver(n) as s:
        s.upload(u=u, v=v, dens=dens, u_prev=u0, v_prev=v0, dens_prev=dens0)
        s.step(1, use_sources=True)
        gu, gv, gd = s.download("u"), s.download("v"), s.download("dens")
    c = (n + 2) // 2
    assert float(gu.sum(dtype=np.float64)) == row["sum_u"]
    assert float(gv.sum(dtype=np.float64)) == row["sum_v"]
    assert float(gd.sum(dtype=np.float64)) == row["sum_dens"]
    assert float(gu[c, c]) == row["u_c"] and float(gd[c, c]) == row["dens_c"]
    for name, a in (("u", gu), ("v", gv), ("dens", gd)):
        assert zlib.crc32(a.view(np.uint8).reshape(-1)) == row["crc_" + name], name


@pytest.mark.parametrize("variant", [0, 3])
@pytest.mark.parametrize("n", [30, 126, 257, 1022])
def test_vel_step_and_dens_step_separately_match_oracle(F, oracle, n, variant):
    """fluid_vel_step / fluid_dens_step (FluidSequential.c:189-241, :176-186) are paths of their own: no
    batched diffusion, k_advect instead of the fused k_gradient_advect.  With and without sources."""
    from fluidsimulationcuda_amd import capi
    rng = np.random.default_rng(n)
    u, v, dens, u0, v0, d0 = (rnd(rng, n) for _ in range(6))
    with F.FluidSolver(n, jacobi=variant, params={capi.PARAM_TB_T16_MIN_CELLS: 0}) as s:
        s.upload(u=u, v=v, dens=dens, u_prev=u0, v_prev=v0, dens_prev=d0)
        s.dens_step()
        oracle.dens_step(dens, d0, u, v)
        assert_bit_equal(s.download("dens"), dens, "dens_step: dens")
        assert_bit_equal(s.download("dens_prev"), d0, "dens_step leaves the diffused density in dens_prev")
        assert_bit_equal(s.download("u"), u, "dens_step leaves u alone")
        s.vel_step()
        oracle.vel_step(u, v, u0, v0)
        s.dens_step(diff=0.02, iters=20)
        oracle.dens_step(dens, d0, u, v, diff=0.02, iters=20)
        for name, want in (("u", u), ("v", v), ("dens", dens), ("u_prev", u0), ("v_prev", v0), ("dens_prev", d0)):
            assert_bit_equal(s.download(name), want, "vel_step + dens_step: %s n=%d" % (name, n))


def test_properties_at_full_size(F):
    """4096^2 (BASELINE config 3): properties that need no CPU run.
    - a uniform field is a fixed point of the diffusion solve when x0 = x
      (beta = 1+4*alpha) up to rounding;
    - Jacobi on the pressure system reduces the residual;
    - after projection the divergence of (u,v) is smaller than before;
    - advect with zero velocity is the identity on the interior."""
    n = 4094
    with F.FluidSolver(n) as s:
        c = np.full((n + 2, n + 2), 0.375, np.float32)
        s.upload(u=c, v=c)
        a, b = F.coefficients(n, DT, DIFF)
        s.diffuse(0, "u", "v", a, b, 40)
        got = s.download("u")
        assert np.abs(got - 0.375).max() <= 4e-7
        rng = np.random.default_rng(0)
        u, v = rnd(rng, n, 0, 1), rnd(rng, n, 0, 1)
        s.upload(u=u, v=v)
        s.set_bnd(1, "u")
        s.set_bnd(2, "v")
        s.computeDivergenceAndPressure("u", "v", "u_prev", "v_prev")
        div0 = np.abs(s.download("v_prev")[1:-1, 1:-1]).max()
        r0 = s.residual("u_prev", "v_prev", 1.0, 4.0)
        s.diffuse(0, "u_prev", "v_prev", 1.0, 4.0, 40)
        assert s.residual("u_prev", "v_prev", 1.0, 4.0) < r0
        s.lastProject("u", "v", "u_prev")
        s.computeDivergenceAndPressure("u", "v", "dens", "dens_prev")
        div1 = np.abs(s.download("dens_prev")[1:-1, 1:-1]).max()
        assert div1 < div0
        s.fill("u", 0.0)
        s.fill("v", 0.0)
        d0 = rnd(rng, n)
        s.upload(dens_prev=d0)
        s.advect(0, "dens", "dens_prev", "u", "v", DT)
        assert_bit_equal(s.download("dens")[1:-1, 1:-1], d0[1:-1, 1:-1], "zero-velocity advect")


def test_variants_agree_at_4096(F, oracle):
    """All three Jacobi kernels give the same bits on the bench workload, and
    they match the oracle on a band of rows (full-size CPU solve is too slow)."""
    n = 4094
    rng = np.random.default_rng(3)
    x, x0 = rnd(rng, n), rnd(rng, n)
    outs = []
    for variant in (0, 1, 2, 3):
        with F.FluidSolver(n, jacobi=variant, params={4: 0}) as s:
            s.upload(u=x, v=x0)
            s.diffuse(0, "u", "v", 1.0, 4.0, 4)
            outs.append(s.download("u"))
    assert_bit_equal(outs[1], outs[0], "LDS vs stream")
    assert_bit_equal(outs[2], outs[0], "naive vs stream")
    assert_bit_equal(outs[3], outs[0], "temporally blocked vs stream")
    want = x.copy()
    oracle.diffuse(0, want, x0, 1.0, 4.0, 4)
    assert_bit_equal(outs[0], want, "stream vs oracle, 4 sweeps at 4096^2")


def test_temporal_blocking_full_solve_at_4096(F):
    """40 sweeps, true-division coefficients, 4096^2: 5 fused launches of 8 ==
    40 single-sweep launches (which test_variants_agree_at_4096 pins to the oracle)."""
    n = 4094
    rng = np.random.default_rng(4)
    x, x0 = rnd(rng, n), rnd(rng, n)
    a, b = F.coefficients(n, DT, VISC)
    outs = []
    for variant in (0, 3):
        with F.FluidSolver(n, jacobi=variant, params={4: 0}) as s:
            s.upload(u=x, v=x0)
            s.diffuse(1, "u", "v", a, b, 40)
            outs.append(s.download("u"))
    assert_bit_equal(outs[1], outs[0], "TB vs stream, 40 sweeps at 4096^2")


def test_step_tb_equals_step_stream_at_4096(F):
    n = 4094
    from fluidsimulationcuda_amd.harness import initialize_parameters
    f = initialize_parameters(n)
    res = []
    for variant in (0, 3):
        with F.FluidSolver(n, jacobi=variant, params={4: 0}) as s:
            s.upload(**f)
            s.step(1, use_sources=True)
            s.step(1)
            res.append([s.download(k) for k in ("u", "v", "dens")])
    for a, b, k in zip(res[0], res[1], "uvd"):
        assert_bit_equal(b, a, "full steps, TB vs stream: " + k)


def test_small_grids_fused_and_single_sweep_launches_give_the_same_bits(F):
    """PARAM_TB_MIN_CELLS (default 0: always fuse) sends slabs below it to one-thread-per-cell
    sweeps; both choices give the same bits."""
    from fluidsimulationcuda_amd.harness import initialize_parameters
    n = 510
    f = initialize_parameters(n)
    res = []
    for params in (None, {4: 1 << 30}):
        with F.FluidSolver(n, params=params) as s:
            s.upload(**f)
            s.step(1, use_sources=True)
            s.step(1)
            res.append([s.download(k) for k in ("u", "v", "dens")])
    for a, b, k in zip(res[0], res[1], "uvd"):
        assert_bit_equal(b, a, "fused (default) vs single-sweep launches: " + k)
