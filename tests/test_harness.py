"""CPU parts of the harness (input recipe, report format) + GPU: state round trip,
per-operator timing, tolerance-terminated solve."""
import io
import os

import numpy as np
import pytest

from conftest import assert_bit_equal, rnd


def test_initialize_parameters_recipe():
    """FluidSequential.c:244-271: density k/1000 only inside the centred square of
    half-width (N+2)/8, velocity sources k/100 everywhere, current fields zero."""
    from fluidsimulationcuda_amd.harness import initialize_parameters
    n = 62
    f = initialize_parameters(n, seed=3)
    w, c, r = n + 2, (n + 2) // 2, (n + 2) // 8
    mask = np.zeros((w, w), bool)
    mask[c - r:c + r, c - r:c + r] = True
    assert (f["dens_prev"][~mask] == 0).all() and 0 < f["dens_prev"][mask].max() < 0.1
    assert not f["u"].any() and not f["v"].any() and not f["dens"].any()
    for name in ("u_prev", "v_prev"):
        k = np.rint(f[name] * 100)
        assert np.array_equal((k / 100.0).astype(np.float32), f[name]) and k.min() >= 0 and k.max() <= 99
    g = initialize_parameters(n, seed=3)
    assert all(np.array_equal(f[k], g[k]) for k in f)
    assert not np.array_equal(initialize_parameters(n, seed=4)["u_prev"], f["u_prev"])


def test_report_format_matches_reference_printf():
    from fluidsimulationcuda_amd.harness import format_report
    txt = format_report(dict(Tot=1.5, Source=0.25, Diffusion=0.125, Divergence=2, Advection=3, Projection=4))
    assert txt == "Tot 1.500000\nSource 0.250000\nDiffusion 0.125000\nDivergence 2.000000\nAdvection 3.000000\nProjection 4.000000\n"


def test_print_state_grid_layout():
    from fluidsimulationcuda_amd.harness import print_state_grid
    d = np.array([[1, 2], [3, 4]], np.float32)
    buf = io.StringIO()
    print_state_grid(d, d * 10, d * 100, out=buf)
    lines = buf.getvalue().splitlines()
    assert lines[1] == "DENSITY" and lines[2] == "[1.000000] [2.000000] "
    assert "VELOCITY" in lines and lines[-1] == "[30.000000, 300.000000] [40.000000, 400.000000] "


def test_print_state_grid_matches_the_references_own_output():
    """tests/golden/state_grid_n14.txt is stdout of the compiled reference's printStateGrid
    (FluidSequential.c:32-52) for the three fields in state_grid_n14.npz (make_golden.py: state_grid)."""
    import io
    from conftest import GOLDEN, load_golden
    from fluidsimulationcuda_amd.harness import print_state_grid
    g = load_golden("state_grid_n14.npz")
    buf = io.StringIO()
    print_state_grid(g["dens"], g["u"], g["v"], out=buf)
    want = open(os.path.join(GOLDEN, "state_grid_n14.txt")).read()
    assert buf.getvalue() == want
    assert "[-0." in want and want.count("\n") == 2 * 16 + 5        # negative velocities, both blocks present


@pytest.mark.gpu
def test_state_dump_of_a_gpu_step_matches_the_references_printout(oracle):
    """The same text from the GPU: one step from the reference's initializeParameters at N = 14, downloaded and
    dumped, against what the reference itself printed for that state."""
    import io
    import fluidsimulationcuda_amd as F
    from conftest import GOLDEN
    from fluidsimulationcuda_amd.harness import print_state_grid
    dens, dens0, u, u0, v, v0 = oracle.initialize_glibc(14, seed=1)
    with F.FluidSolver(14) as s:
        s.upload(u=u, v=v, dens=dens, u_prev=u0, v_prev=v0, dens_prev=dens0)
        s.step(1, use_sources=True)
        buf = io.StringIO()
        print_state_grid(s.download("dens"), s.download("u"), s.download("v"), out=buf)
    assert buf.getvalue() == open(os.path.join(GOLDEN, "state_grid_n14.txt")).read()


def test_threaded_oracle_sweeps_match(oracle):
    rng = np.random.default_rng(8)
    x, x0 = rnd(rng, 97), rnd(rng, 97)
    a, b = x.copy(), x.copy()
    oracle.diffuse(2, a, x0, 0.4, 2.6, 6)
    oracle.diffuse_threaded(2, b, x0, 0.4, 2.6, 6, threads=5)
    assert_bit_equal(b, a, "row-band threaded sweeps")


@pytest.mark.gpu
def test_state_round_trip(tmp_path):
    import fluidsimulationcuda_amd as F
    from fluidsimulationcuda_amd.harness import initialize_parameters, load_state, save_state
    n = 126
    path = os.path.join(tmp_path, "state.f32")
    with F.FluidSolver(n) as s:
        s.upload(**initialize_parameters(n))
        s.step(1, use_sources=True)
        s.step(1)
        save_state(s, path, step_index=2)
        s.step(2)
        want = {k: s.download(k) for k in ("u", "v", "dens")}
    assert os.path.getsize(path) == 8 + 8 + 6 * 128 * 128 * 4
    with F.FluidSolver(n) as s:
        assert load_state(s, path) == 2
        s.step(2)
        for k in want:
            assert_bit_equal(s.download(k), want[k], "resume from dump: " + k)
    with F.FluidSolver(30) as s, pytest.raises(ValueError):
        load_state(s, path)


@pytest.mark.gpu
def test_timing_categories_count_every_operator():
    import fluidsimulationcuda_amd as F
    from fluidsimulationcuda_amd.harness import initialize_parameters, run_steps
    n = 254
    from fluidsimulationcuda_amd import capi
    with F.FluidSolver(n, params={capi.PARAM_FUSE_DIVERGENCE: 0, capi.PARAM_FUSE_ADD_SOURCE: 0}) as s:     # every operator as a launch of its own
        s.upload(**initialize_parameters(n))
        s.timing_enable(True)
        s.step(1, use_sources=True)
        s.step(2)
        t = s.timing_read()
        # per step: 5 solves = 200 sweeps (the 3 diffusions run as one batch call, the 2 pressure solves
        # on their own), 2 divergence, 2 gradient, 3 advections in 1 launch of their own (u and v share
        # it; the density's rides in the second gradient launch).  add_source: 3 kernels
        # in the step that has sources; once the sources are zero (FluidSequential.c:298-302) x + dt*0
        # rides in the solve's load of its right-hand side.
        assert (t["source_calls"], t["solves"], t["sweeps"]) == (3, 9, 600)
        assert (t["divergence_calls"], t["projection_calls"], t["advection_calls"]) == (6, 6, 3)
        assert all(t[k + "_ms"] > 0 for k in ("source", "diffusion", "divergence", "projection", "advection"))
        s.timing_enable(False)
        r = run_steps(s, 3, first_uses_sources=False)
        assert r["sweeps"] == 600 and 0 < r["Diffusion"] < r["Tot"]
    with F.FluidSolver(n) as s:          # default: the divergence is computed inside the pressure solve's first launch
        s.upload(**initialize_parameters(n))
        s.timing_enable(True)
        s.step(1, use_sources=True)
        s.step(2)
        t = s.timing_read()
        assert (t["divergence_calls"], t["solves"], t["sweeps"], t["projection_calls"]) == (0, 9, 600, 6)
        assert t["source_calls"] == 0      # ... and the sources are added by the first launch of the diffusion that consumes the sums


@pytest.mark.gpu
@pytest.mark.parametrize("n,check_every", [(126, 8), (61, 2), (300, 16)])
def test_tolerance_terminated_solve_is_opt_in_extension(oracle, n, check_every):
    """fluid_op_diffuse_tol (not the reference's behaviour: it always runs 40 sweeps, FluidSequential.c:91): the field
    it returns is the oracle's after exactly the sweeps it reports, and the residual it reports is the max-norm
    residual of that field (float64 evaluation)."""
    import fluidsimulationcuda_amd as F
    rng = np.random.default_rng(n)
    x0 = rnd(rng, n)
    z = np.zeros_like(x0)
    with F.FluidSolver(n) as s:
        s.upload(u=z, v=x0)
        a, b = F.coefficients(n, 0.016, 0.0025)
        r0 = s.residual("u", "v", a, b)
        its, res = s.diffuse_tol(0, "u", "v", a, b, tol=r0 * 1e-4, max_iters=4000, check_every=check_every)
        assert 0 < its < 4000 and its % check_every == 0 and res <= r0 * 1e-4
        got = s.download("u")
        want = z.copy()
        oracle.diffuse(0, want, x0, a, b, its)
        assert_bit_equal(got, want, "diffuse_tol == oracle.diffuse(%d sweeps)" % its)
        # its own residual: max |beta*x - alpha*(L+R+U+D) - x0| over the interior, here in float64
        w = want.astype(np.float64)
        nb = w[1:-1, :-2] + w[1:-1, 2:] + w[:-2, 1:-1] + w[2:, 1:-1]
        ref = np.abs(np.float64(np.float32(b)) * w[1:-1, 1:-1] - np.float64(np.float32(a)) * nb - x0[1:-1, 1:-1]).max()
        assert abs(res - ref) <= 1e-4 * ref + 1e-7 * np.abs(x0).max(), (res, ref)
        # one block of sweeps earlier the tolerance was not met yet
        if its > check_every:
            prev = z.copy()
            oracle.diffuse(0, prev, x0, a, b, its - check_every)
            p = prev.astype(np.float64)
            nbp = p[1:-1, :-2] + p[1:-1, 2:] + p[:-2, 1:-1] + p[2:, 1:-1]
            assert np.abs(np.float64(np.float32(b)) * p[1:-1, 1:-1] - np.float64(np.float32(a)) * nbp - x0[1:-1, 1:-1]).max() > r0 * 1e-4 * 0.999
        its2, _ = s.diffuse_tol(0, "u", "v", a, b, tol=1e30)
        assert its2 == 0                                   # already converged: no sweeps
        its3, res3 = s.diffuse_tol(0, "u", "v", a, b, tol=0.0, max_iters=10, check_every=4)
        assert its3 == 10 and res3 > 0.0                   # the cap: blocks of check_every, the last one shortened to an even rest
