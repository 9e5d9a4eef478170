"""GPU: grids whose fields outgrow 32-bit offsets.  From N ~ 23170 a fp32 field passes 2 GiB (the fused Jacobi kernel
addresses fields through 32-bit buffer offsets and hands over to the single-sweep kernels), from N = 32767 it passes
4 GiB (advect and the gradient / divergence kernels switch from 32-bit to 64-bit byte offsets).  No oracle finishes a
grid of 10^9 cells in test time, and none is needed: every operator here is local, so the reference's expression
(FluidSequential.c, cited per check) is evaluated in numpy float32 at a few thousand SAMPLED cells -- the four corners'
neighbourhoods, the walls, the last rows (the highest addresses), random interior cells -- and compared bit for bit.
Fields are a 1024 x 1024 random block tiled over the grid (cheap to make, still different at every sampled stencil)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
DT = 0.016
f32 = np.float32


def tiled(n, seed, lo=-1.0, hi=1.0):
    rng = np.random.default_rng(seed)
    blk = rng.uniform(lo, hi, size=(1024, 1000)).astype(np.float32)     # 1000 columns: rows of tiles do not line up
    w = n + 2
    reps = (w + 1023) // 1024, (w + 999) // 1000
    return np.ascontiguousarray(np.tile(blk, reps)[:w, :w])


def samples(n, count, seed):
    """(i, j) interior cells: a band along each wall, the corners, and random ones; row-major, rows i, columns j."""
    rng = np.random.default_rng(seed)
    edge = np.concatenate([np.arange(1, 6), np.arange(n - 4, n + 1)])
    ii = np.concatenate([rng.integers(1, n + 1, count), np.repeat(edge, len(edge)), rng.choice(edge, count), rng.integers(1, n + 1, count)])
    jj = np.concatenate([rng.integers(1, n + 1, count), np.tile(edge, len(edge)), rng.integers(1, n + 1, count), rng.choice(edge, count)])
    return ii.astype(np.int64), jj.astype(np.int64)


def bits_equal(got, want, what):
    got, want = np.asarray(got, np.float32), np.asarray(want, np.float32)
    bad = got.view(np.uint32) != want.view(np.uint32)
    assert not bad.any(), "%s: %d of %d sampled cells differ; first got %r want %r" % (what, bad.sum(), bad.size, got[bad][0], want[bad][0])


def ghost_checks(x, b, n, what):
    """set_bnd (FluidSequential.c:62-75) on whole ghost rows / columns and the corners."""
    sx, sy = (f32(-1) if b == 1 else f32(1)), (f32(-1) if b == 2 else f32(1))
    bits_equal(x[1:n + 1, 0], sx * x[1:n + 1, 1], what + ": ghost column 0")
    bits_equal(x[1:n + 1, n + 1], sx * x[1:n + 1, n], what + ": ghost column n+1")
    bits_equal(x[0, 1:n + 1], sy * x[1, 1:n + 1], what + ": ghost row 0")
    bits_equal(x[n + 1, 1:n + 1], sy * x[n, 1:n + 1], what + ": ghost row n+1")
    for (gi, gj), (ai, aj), (bi, bj) in (((0, 0), (0, 1), (1, 0)), ((0, n + 1), (0, n), (1, n + 1)),
                                          ((n + 1, 0), (n + 1, 1), (n, 0)), ((n + 1, n + 1), (n + 1, n), (n, n + 1))):
        bits_equal(x[gi, gj], f32(0.5) * (x[ai, aj] + x[bi, bj]), what + ": corner")


# FLUID_HUGE_N=65533 adds the largest grid the library takes (17 GiB per field; ~150 GiB of host memory for this test)
SIZES = [24000, 32800] + ([int(os.environ["FLUID_HUGE_N"])] if os.environ.get("FLUID_HUGE_N") else [])


@pytest.mark.parametrize("n", SIZES)
def test_operators_at_sampled_cells_of_a_huge_grid(n):
    import fluidsimulationcuda_amd as F
    u, v, d = tiled(n, 1), tiled(n, 2), tiled(n, 3, 0.0, 1.0)
    ii, jj = samples(n, 2000, 4)
    with F.FluidSolver(n) as s:
        s.upload(u=u, v=v, dens=d)
        # --- one Jacobi sweep + set_bnd, general coefficients (FluidSequential.c:93-96): out = (x0 + a*(((L+R)+U)+D))/c
        a, c = F.coefficients(n, DT, 1e-9)                    # a tiny viscosity keeps a of order 1 at this N
        a, c = f32(a), f32(c)
        s.jacobi_sweep(1, "u", "v", "u_prev", a, c)
        got = s.download("u_prev")
        nb = ((u[ii, jj - 1] + u[ii, jj + 1]) + u[ii - 1, jj]) + u[ii + 1, jj]
        bits_equal(got[ii, jj], (v[ii, jj] + a * nb) / c, "jacobi sweep, N=%d" % n)
        ghost_checks(got, 1, n, "jacobi sweep b=1, N=%d" % n)
        # --- two sweeps through diffuse() (past 2 GiB: single-sweep launches, ping-pong, result back in x)
        s.upload(u_prev=u)
        s.diffuse(2, "u_prev", "v", a, c, 2)
        got2 = s.download("u_prev")
        inner = (ii > 2) & (ii < n - 1) & (jj > 2) & (jj < n - 1)          # cells whose two-sweep cone stays off the walls
        i2, j2 = ii[inner], jj[inner]

        def sweep_at(src, i, j):
            return (v[i, j] + a * (((src(i, j - 1) + src(i, j + 1)) + src(i - 1, j)) + src(i + 1, j))) / c
        first = lambda i, j: sweep_at(lambda p, q: u[p, q], i, j)           # noqa: E731
        bits_equal(got2[i2, j2], sweep_at(first, i2, j2), "diffuse, 2 sweeps, N=%d" % n)
        ghost_checks(got2, 2, n, "diffuse b=2, N=%d" % n)
        del got, got2
        # --- divergence (FluidSequential.c:151-152) and gradient subtraction (:167-168)
        s.computeDivergenceAndPressure("u", "v", "u_prev", "v_prev")
        div = s.download("v_prev")
        h = f32(1.0) / f32(n)
        bits_equal(div[ii, jj], (f32(-0.5) * h) * (((u[ii, jj + 1] - u[ii, jj - 1]) + v[ii + 1, jj]) - v[ii - 1, jj]), "divergence, N=%d" % n)
        ghost_checks(div, 0, n, "divergence, N=%d" % n)
        assert not s.download("u_prev").any(), "p = 0"
        del div
        s.upload(dens_prev=d)                                   # stands in for p
        s.lastProject("u", "v", "dens_prev")
        gu, gv = s.download("u"), s.download("v")
        bits_equal(gu[ii, jj], u[ii, jj] - (f32(0.5) * (d[ii, jj + 1] - d[ii, jj - 1])) / h, "gradient u, N=%d" % n)
        bits_equal(gv[ii, jj], v[ii, jj] - (f32(0.5) * (d[ii + 1, jj] - d[ii - 1, jj])) / h, "gradient v, N=%d" % n)
        ghost_checks(gu, 1, n, "gradient u, N=%d" % n)
        ghost_checks(gv, 2, n, "gradient v, N=%d" % n)
        assert s.absmax_velocity("u", "v") == max(np.abs(gu[1:n + 1, 1:n + 1]).max(), np.abs(gv[1:n + 1, 1:n + 1]).max())
        del gu, gv
        # --- advect (FluidSequential.c:107-141) along a velocity small enough to stay a few cells from home, and along
        # the full-size one (every back-trace clamps to a wall)
        for scale, tag in ((f32(2e-3), "short"), (f32(1.0), "clamped")):
            uu, vv = (u * scale).astype(np.float32), (v * scale).astype(np.float32)
            s.upload(u=uu, v=vv, dens_prev=d)
            s.advect(0, "dens", "dens_prev", "u", "v")
            got = s.download("dens")
            dt0 = f32(DT) * f32(n)
            x = jj.astype(np.float32) - dt0 * uu[ii, jj]
            y = ii.astype(np.float32) - dt0 * vv[ii, jj]
            x = np.minimum(np.maximum(x, f32(0.5)), f32(n) + f32(0.5))
            y = np.minimum(np.maximum(y, f32(0.5)), f32(n) + f32(0.5))
            j0, i0 = x.astype(np.int64), y.astype(np.int64)
            s1 = x - j0.astype(np.float32)
            s0 = f32(1) - s1
            t1 = y - i0.astype(np.float32)
            t0 = f32(1) - t1
            want = s0 * (t0 * d[i0, j0] + t1 * d[i0 + 1, j0]) + s1 * (t0 * d[i0, j0 + 1] + t1 * d[i0 + 1, j0 + 1])
            bits_equal(got[ii, jj], want, "advect (%s), N=%d" % (tag, n))
            ghost_checks(got, 0, n, "advect (%s), N=%d" % (tag, n))
            del got, uu, vv
        # --- add_source (FluidSequential.c:78-82), every cell incl. ghosts: the last rows are the highest addresses
        s.upload(u=u, u_prev=v)
        s.add_source("u", "u_prev")
        got = s.download("u")
        bits_equal(got[-3:], u[-3:] + f32(DT) * v[-3:], "add_source, last rows, N=%d" % n)
        bits_equal(got[ii, jj], u[ii, jj] + f32(DT) * v[ii, jj], "add_source, N=%d" % n)
        del got
        # --- a whole step (4 sweeps per solve) runs, and leaves fields whose ghost cells obey set_bnd
        s.upload(u=(u * f32(1e-3)).astype(np.float32), v=(v * f32(1e-3)).astype(np.float32), dens=d, u_prev=u, v_prev=v, dens_prev=d)
        s.step(1, use_sources=True, iters=4)
        for name, b in (("u", 1), ("v", 2), ("dens", 0)):
            got = s.download(name)
            assert np.isfinite(got).all()
            ghost_checks(got, b, n, "%s after a step, N=%d" % (name, n))
            del got
