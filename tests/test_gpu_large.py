"""GPU: BASELINE configs 3 and 4 at their own sizes -- 8192^2 (fp32, one context and row slabs) and
16384^2 (fp16 storage; fp32 against the reference's own checksums and a windowed oracle) -- against the oracle itself, not
against another HIP path.  These sizes select
kernels no smaller grid reaches by default: 16-sweep fused launches of the general (diffusion) form need a
field of more than 96 MiB.  Every test asserts the launch schedule it means to exercise.

CPU cost (one host core): a 40-sweep solve at 8192^2 ~2-3 s, a full step ~12 s; 16 sweeps at 16384^2 ~4 s."""
import json
import os
import zlib

import numpy as np
import pytest

from conftest import GOLDEN, assert_bit_equal, rnd

pytestmark = pytest.mark.gpu
DT, VISC, DIFF = 0.016, 0.0025, 0.1


@pytest.fixture(scope="module")
def F():
    import fluidsimulationcuda_amd as F
    return F


def crc(a):
    return zlib.crc32(np.ascontiguousarray(a, dtype=np.float32).view(np.uint8).reshape(-1))


@pytest.mark.parametrize("form", ["pressure", "viscosity", "density"])
def test_40_sweep_solve_at_8192_matches_oracle(F, oracle, form):
    """FluidSequential.c:85-104 at N = 8190: the default schedule is 16 + 12 + 12 sweeps per launch for both
    forms (k_jacobi_tb<16,4,2> / <16,2,2> and <12,4,2> / <12,2,2>)."""
    n = 8190
    rng = np.random.default_rng({"pressure": 1, "viscosity": 2, "density": 3}[form])
    b, (alpha, beta) = {"pressure": (0, (1.0, 4.0)), "viscosity": (1, F.coefficients(n, DT, VISC)),
                        "density": (0, F.coefficients(n, DT, DIFF))}[form]
    x, x0 = rnd(rng, n), rnd(rng, n)
    with F.FluidSolver(n) as s:
        s.upload(u=x, v=x0)
        s.timing_enable(True)
        s.timing_read(reset=True)
        s.diffuse(b, "u", "v", alpha, beta, 40)
        t = s.timing_read(reset=True)
        got = s.download("u")
    assert t["jacobi_launches"] == 3 and t["sweeps"] == 40, "expected 16 + 12 + 12 sweeps, got %r" % (t,)
    oracle.diffuse(b, x, x0, alpha, beta, 40)
    assert_bit_equal(got, x, "40-sweep %s solve at 8192^2" % form)


def test_true_division_16_sweep_kernel_at_8192_matches_oracle(F, oracle):
    """k_jacobi_tb<16,0,2>: the general form with FAST_DIVISION off (the fallback every beta that fails the
    on-device proof takes)."""
    from fluidsimulationcuda_amd import capi
    n = 8190
    rng = np.random.default_rng(4)
    alpha, beta = F.coefficients(n, DT, VISC)
    x, x0 = rnd(rng, n), rnd(rng, n)
    with F.FluidSolver(n, params={capi.PARAM_TB_FAST_DIVISION: 0}) as s:
        s.upload(u=x, v=x0)
        s.timing_enable(True)
        s.timing_read(reset=True)
        s.diffuse(2, "u", "v", alpha, beta, 16)
        t = s.timing_read(reset=True)
        got = s.download("u")
    assert t["jacobi_launches"] == 1
    oracle.diffuse(2, x, x0, alpha, beta, 16)
    assert_bit_equal(got, x, "16 sweeps, true division, 8192^2")


def test_two_steps_at_8192_match_oracle_and_two_slabs(F, oracle):
    """BASELINE config 3's grid: step_src + step (FluidSequential.c:298-306) in one context against the
    oracle, all six fields; then the same two steps on two row slabs (in-process fabric, one GPU) against
    that -- the slab kernels at the size they are meant for."""
    from test_gpu_slab import run_ranks
    from fluidsimulationcuda_amd.harness import initialize_parameters
    n = 8190
    f = initialize_parameters(n, seed=3)

    def body(s):
        s.step(1, use_sources=True)
        s.step(1)

    with F.FluidSolver(n) as s:
        s.upload(**f)
        s.timing_enable(True)
        body(s)
        t = s.timing_read()
        one = {k: s.download(k) for k in ("u", "v", "dens", "u_prev", "v_prev", "dens_prev")}
    # per step: the three diffusions share launches (16+12+12), each projection has its own 16+12+12
    assert t["jacobi_launches"] == 2 * 9 and t["sweeps"] == 2 * 200, t
    got, fab = run_ranks(n, 2, 0, f, body, jacobi=3)
    for k in ("u", "v", "dens"):
        assert_bit_equal(got[k], one[k], "%s: 2 slabs vs one context at 8192^2" % k)
    assert fab.log[1] == fab.log[0]
    del got
    w = {k: v.copy() for k, v in f.items()}
    oracle.step_src(w["u"], w["v"], w["dens"], w["u_prev"], w["v_prev"], w["dens_prev"])
    oracle.step(w["u"], w["v"], w["dens"], w["u_prev"], w["v_prev"], w["dens_prev"])
    for k in ("u", "v", "dens", "u_prev", "v_prev", "dens_prev"):
        assert_bit_equal(one[k], w[k], "%s after two steps at 8192^2" % k)


@pytest.mark.parametrize("n", [1022, 4094, 8190, 16382])
def test_reference_crc_at_full_size(F, n):
    """Step 1 from the reference's own initializeParameters (glibc rand, seed 1): CRC-32 of the bytes of u, v
    and dens as the compiled reference left them (tests/golden/checksums.json, make_golden.py)."""
    from oracle.oracle import Oracle
    row = [r for r in json.load(open(os.path.join(GOLDEN, "checksums.json"))) if r["n"] == n][0]
    dens, dens0, u, u0, v, v0 = Oracle().initialize_glibc(n, seed=1)
    with F.FluidSolver(n) as s:
        s.upload(u=u, v=v, dens=dens, u_prev=u0, v_prev=v0, dens_prev=dens0)
        s.step(1, use_sources=True)
        gu, gv, gd = s.download("u"), s.download("v"), s.download("dens")
    assert (crc(gu), crc(gv), crc(gd)) == (row["crc_u"], row["crc_v"], row["crc_dens"])


def test_fp16_storage_against_fp32_at_16384(F):
    """BASELINE config 5 at its own size: the reference's loop (FluidSequential.c:289-312; its own initializeParameters,
    glibc rand seed 1) for one sourced and two plain steps at N = 16382 with fp16 fields, against the fp32 fields of the same
    steps on the GPU -- which test_reference_crc_at_full_size / test_reference_trajectory_crc tie to the compiled
    reference's bytes (the reference allocates float, :277-282: it has no fp16 mode, so the distance from the fp32 result is
    this mode's only tie to it).  Asserted: every field stays within 2^-8 of the fp32 field's largest magnitude (fp16
    carries 11 bits; ~40 roundings per step accumulate a few ulps).  Reported (gpurun_out/f16_vs_f32_16384.json, quoted in
    DESIGN.md): the largest relative error over the cells that carry at least 2^-6 of the field's magnitude, and the step
    of the decaying loop at which the fp16 velocities have flushed to exact zeros (fp16's smallest subnormal is 6e-8;
    the reference's fields shrink by orders of magnitude per step)."""
    from oracle.oracle import Oracle
    from fluidsimulationcuda_amd import capi
    n = 16382
    # Bounds (of each fp32 field's largest magnitude; measured values in the report): velocities 2^-9 + 4e-6 N = 6.7e-2 -- the
    # projection subtracts the GRADIENT of a smooth pressure, a small difference of neighbouring fp16 values, so its error grows
    # with the grid (measured 3.3e-2 at a few cells, 1.5e-3 rms); the density, advected ~150 cells along those velocities,
    # is off by a cell or two at its sharp front -- up to 0.2 of its magnitude there, 2e-3 rms (asserted: rms <= 2^-6).
    # From the third step of the reference's decaying loop on the velocities themselves (5e-7) are fp16 subnormals: the
    # floor of a few fp16 quanta (6e-8) applies.
    BOUND = {"u": 2.0 ** -9 + 4e-6 * n, "v": 2.0 ** -9 + 4e-6 * n, "dens": 2.0 ** -1}
    FLOOR = 8 * 2.0 ** -24
    dens, dens0, u, u0, v, v0 = Oracle().initialize_glibc(n, seed=1)
    report = {"n": n, "asserted_bound_of_scale": BOUND, "steps": []}
    with F.FluidSolver(n) as s32, F.FluidSolver(n, storage=capi.STORAGE_F16) as s16:
        for s in (s32, s16):
            s.upload(u=u, v=v, dens=dens, u_prev=u0, v_prev=v0, dens_prev=dens0)
        del dens0, u0, v0
        a32, a16 = np.empty_like(u), np.empty_like(u)
        for z in (1, 2, 3):
            for s in (s32, s16):
                s.step(1, use_sources=(z == 1))
            row = {"step": z}
            for k in ("u", "v", "dens"):
                s32.download(k, out=a32)
                s16.download(k, out=a16)
                scale = float(np.abs(a32).max())
                np.subtract(a16, a32, out=a16)
                np.abs(a16, out=a16)
                err = float(a16.max())
                big = np.abs(a32) >= scale * 2.0 ** -6
                rel = float((a16[big] / np.abs(a32[big])).max()) if big.any() else 0.0
                rms = float(np.sqrt(np.mean(np.square(a16, dtype=np.float64))))
                row[k] = {"max_abs_fp32": scale, "max_abs_err": err, "err_over_scale": err / scale if scale else 0.0,
                          "rms_err_over_scale": rms / scale if scale else 0.0, "max_rel_err_cells_above_scale_2^-6": rel}
                assert np.isfinite(err) and err <= max(scale * BOUND[k], FLOOR), "step %d %s: err %.3g vs scale %.3g" % (z, k, err, scale)
                assert rms <= max(scale * 2.0 ** -6, FLOOR), "step %d %s: rms err %.3g vs scale %.3g" % (z, k, rms, scale)
            report["steps"].append(row)
        # how long until the fp16 velocities are exact zeros (fp32 keeps shrinking through its denormals much longer)
        flushed = None
        for z in range(4, 41):
            s16.step(1)
            if s16.absmax_velocity("u", "v") == 0.0:
                flushed = z
                break
        report["fp16_velocities_exactly_zero_after_step"] = flushed
        s32.step(max((flushed or 40) - 3, 0))
        report["fp32_max_velocity_at_that_step"] = s32.absmax_velocity("u", "v")
    assert flushed is not None, "the decaying loop must flush fp16 velocities to zero within 40 steps"
    out = os.path.join(os.path.dirname(GOLDEN), "..", "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "f16_vs_f32_16384.json"), "w") as f:
        json.dump(report, f, indent=1)
    print("fp16 vs fp32 at 16384^2:", json.dumps(report))


def test_fp16_fused_launches_at_16384_match_rounded_oracle(F, oracle):
    """BASELINE config 4's grid (fp16 fields, fp32 accumulate): two fused launches of 8 sweeps of the
    density-diffusion form at N = 16382 against the oracle's fp32 sweeps with one numpy.float16 rounding
    per launch (the parity recipe of test_gpu_f16.py)."""
    from fluidsimulationcuda_amd import capi
    from test_gpu_f16 import emu_solve, h
    n = 16382
    rng = np.random.default_rng(16)
    alpha, beta = F.coefficients(n, DT, DIFF)
    x = rng.random((n + 2, n + 2), dtype=np.float32) - np.float32(0.5)
    x0 = rng.random((n + 2, n + 2), dtype=np.float32) - np.float32(0.5)
    with F.FluidSolver(n, storage=capi.STORAGE_F16) as s:
        s.upload(u=x, v=x0)
        s.timing_enable(True)
        s.timing_read(reset=True)
        s.diffuse(0, "u", "v", alpha, beta, 16)
        t = s.timing_read(reset=True)
        got = s.download("u")
    assert t["jacobi_launches"] == 2
    want = emu_solve(oracle, 0, x, x0, alpha, beta, [8, 8])
    del x, x0
    assert_bit_equal(got, want, "fp16 storage, 8 + 8 sweeps at 16384^2")
    assert_bit_equal(got, h(got), "stored values are fp16")


def test_fp16_two_slabs_at_16384_match_one_context(F):
    """BASELINE config 4 as it is meant to run -- 16384^2, fp16 storage, row slabs: two slabs (in-process fabric, one
    GPU) against one context, two steps, every bit of u, v and dens.  (With fp16 storage every launch rounds once, so
    this also pins that slabs and one GPU take the same launch schedule at this size; the one-context arithmetic is
    pinned to the rounded oracle above and in test_gpu_f16.py.)"""
    from test_gpu_slab import run_ranks
    from fluidsimulationcuda_amd import capi
    from fluidsimulationcuda_amd.harness import initialize_parameters
    n = 16382
    f = initialize_parameters(n, seed=4)

    def body(s):
        s.step(1, use_sources=True)
        s.step(1)

    with F.FluidSolver(n, storage=capi.STORAGE_F16) as s:
        s.upload(**f)
        body(s)
        one = {k: s.download(k) for k in ("u", "v", "dens")}
    got, fab = run_ranks(n, 2, 0, f, body, jacobi=3, storage=capi.STORAGE_F16)
    assert fab.log[1] == fab.log[0]
    for k in ("u", "v", "dens"):
        assert_bit_equal(got[k], one[k], "%s: 2 slabs vs one context at 16384^2, fp16 storage" % k)


@pytest.mark.parametrize("form", ["pressure", "viscosity"])
def test_40_sweep_solve_at_16384_fp32_matches_windowed_oracle(F, oracle, form):
    """16384^2 in fp32 (1 GiB per field; bench.py --grid 16384): a full-grid oracle solve takes minutes, but a cell after
    k sweeps depends only on the (2k+1)^2 cells around it, so the oracle runs on WINDOWS: 83 x 83 cells cut out around a
    sampled cell (41 > 40 sweeps from the cut, so what the oracle's set_bnd does to the cut edge never reaches the centre),
    or cut flush with the domain's walls and corners, where the window's own ghost ring IS the domain's and set_bnd
    (FluidSequential.c:62-75) is replayed for real.  Bit equality at every cell the cut cannot have touched."""
    n, iters = 16382, 40
    r = iters + 1
    rng = np.random.default_rng(160 + len(form))
    b, (alpha, beta) = {"pressure": (0, (1.0, 4.0)), "viscosity": (1, F.coefficients(n, DT, VISC))}[form]
    x, x0 = rnd(rng, n), rnd(rng, n)
    with F.FluidSolver(n) as s:
        s.upload(u=x, v=x0)
        s.timing_enable(True)
        s.timing_read(reset=True)
        s.diffuse(b, "u", "v", alpha, beta, iters)
        t = s.timing_read(reset=True)
        got = s.download("u")
    assert t["jacobi_launches"] == 3 and t["sweeps"] == iters, t          # 16 + 12 + 12
    w = 2 * r + 1
    # window origins (row, col of the window's ghost ring): flush with each wall / corner, and random interior ones
    lo, hi = 0, n + 2 - w
    origins = [(lo, lo), (lo, hi), (hi, lo), (hi, hi), (lo, 5000), (hi, 7001), (4000, lo), (9000, hi)]
    origins += [(int(a), int(c)) for a, c in zip(rng.integers(1, hi, 24), rng.integers(1, hi, 24))]
    origins += [(8191 - r, 8191 - r), (16382 - w, 3), (3, 16382 - w), (128 * 40 - r, 96 * 50 - r)]   # strip / window seams
    for (i0, j0) in origins:
        xw = np.ascontiguousarray(x[i0:i0 + w, j0:j0 + w])
        x0w = np.ascontiguousarray(x0[i0:i0 + w, j0:j0 + w])
        oracle.diffuse(b, xw, x0w, alpha, beta, iters)
        # cells the cut can have reached: within `iters` of a cut edge (a side that lies on the domain's wall is no cut)
        a0 = 0 if i0 == 0 else iters + 1
        a1 = w if i0 + w == n + 2 else w - iters - 1
        c0 = 0 if j0 == 0 else iters + 1
        c1 = w if j0 + w == n + 2 else w - iters - 1
        assert a1 > a0 and c1 > c0
        assert_bit_equal(got[i0 + a0:i0 + a1, j0 + c0:j0 + c1], xw[a0:a1, c0:c1],
                         "%s solve at 16384^2, window at (%d, %d)" % (form, i0, j0))


def test_two_steps_at_8192_on_eight_slabs_match_one_context(F):
    """BASELINE config 3 as the driver's scaling run splits it -- 8192^2 over EIGHT row slabs (1023 or 1024 rows each,
    default ghost zones, density diffusion on the second stream, advections started on the previous bound) -- here as
    eight contexts on this one GPU behind the in-process fabric, against one context: every bit of u, v and dens after
    the sourced step and a plain one, and the same exchange sequence on every rank."""
    from test_gpu_slab import run_ranks
    from fluidsimulationcuda_amd.harness import initialize_parameters
    n = 8190
    f = initialize_parameters(n, seed=8)

    def body(s):
        s.step(1, use_sources=True)
        s.step(1)

    with F.FluidSolver(n) as s:
        s.upload(**f)
        body(s)
        one = {k: s.download(k) for k in ("u", "v", "dens")}
    got, fab = run_ranks(n, 8, 0, f, body, jacobi=3)
    for r in range(1, 8):
        assert fab.log[r] == fab.log[0]
    for k in ("u", "v", "dens"):
        assert_bit_equal(got[k], one[k], "%s: 8 slabs vs one context at 8192^2" % k)


def test_two_steps_at_16384_fp16_on_eight_slabs_match_one_context(F):
    """BASELINE config 4 with its own split: 16384^2, fp16 storage, EIGHT row slabs (in-process fabric, one GPU) against
    one context -- with fp16 storage every launch rounds once, so this also pins that eight slabs and one GPU take the
    same launch schedule."""
    from test_gpu_slab import run_ranks
    from fluidsimulationcuda_amd import capi
    from fluidsimulationcuda_amd.harness import initialize_parameters
    n = 16382
    f = initialize_parameters(n, seed=9)

    def body(s):
        s.step(1, use_sources=True)
        s.step(1)

    with F.FluidSolver(n, storage=capi.STORAGE_F16) as s:
        s.upload(**f)
        body(s)
        one = {k: s.download(k) for k in ("u", "v", "dens")}
    got, fab = run_ranks(n, 8, 0, f, body, jacobi=3, storage=capi.STORAGE_F16)
    for r in range(1, 8):
        assert fab.log[r] == fab.log[0]
    for k in ("u", "v", "dens"):
        assert_bit_equal(got[k], one[k], "%s: 8 slabs vs one context at 16384^2, fp16 storage" % k)


@pytest.mark.parametrize("n", [1022, 4094, 8190, 16382])
def test_reference_trajectory_crc(F, n):
    """The reference's own loop over 10 (N = 1022) / 5 (N = 4094) / 3 (N = 8190) / 2 (N = 16382) steps -- sources at step 0 only, fields decaying by one
    to two orders of magnitude per step -- as the compiled reference ran it (tests/golden/trajectory_checksums.json,
    make_golden.py trajectory): CRC-32 of u, v and dens after every single step, device-resident throughout."""
    from oracle.oracle import Oracle
    rows = [r for r in json.load(open(os.path.join(GOLDEN, "trajectory_checksums.json"))) if r["n"] == n]
    assert len(rows) >= 2
    dens, dens0, u, u0, v, v0 = Oracle().initialize_glibc(n, seed=1)
    with F.FluidSolver(n) as s:
        s.upload(u=u, v=v, dens=dens, u_prev=u0, v_prev=v0, dens_prev=dens0)
        for row in rows:
            s.step(1, use_sources=row["step"] == 1)
            got = tuple(crc(s.download(k)) for k in ("u", "v", "dens"))
            assert got == (row["crc_u"], row["crc_v"], row["crc_dens"]), "N=%d, step %d" % (n, row["step"])
