"""The reference's benchmark harness, restated for this package: synthetic
input (initializeParameters, FluidSequential.c:244-271) and the Z-step mean-time
loop with per-solve breakdown (:289-324).

The reference draws from unseeded glibc rand(); the GPU box must not depend on
a libc, so draws come from numpy's PCG64 with an explicit seed.  The recipe is
the reference's: density source k/1000 (k uniform in 0..99) inside the centred
square of half-width (N+2)/8 and 0 outside; velocity sources k/100 on all
(N+2)^2 cells; current fields zero.
"""
import time

import numpy as np

from .solver import DIFF, DT, ITERS, VIS


def initialize_parameters(n, seed=1):
    """-> dict(dens, dens_prev, u, u_prev, v, v_prev), each (n+2, n+2) float32."""
    w = n + 2
    c, r = w // 2, w // 8
    rng = np.random.default_rng(seed)
    dens_prev = np.zeros((w, w), dtype=np.float32)
    k = rng.integers(0, 100, size=(2 * r, 2 * r), dtype=np.int32)
    dens_prev[c - r:c + r, c - r:c + r] = k.astype(np.float32) / np.float32(1000.0)
    u_prev = rng.integers(0, 100, size=(w, w), dtype=np.int32).astype(np.float32) / np.float32(100.0)
    v_prev = rng.integers(0, 100, size=(w, w), dtype=np.int32).astype(np.float32) / np.float32(100.0)
    z = np.zeros((w, w), dtype=np.float32)
    return dict(dens=z, dens_prev=dens_prev, u=z.copy(), u_prev=u_prev, v=z.copy(), v_prev=v_prev)


def run_steps(solver, steps, dt=DT, diff=DIFF, visc=VIS, iters=ITERS, first_uses_sources=True):
    """Z steps as the reference's main does, returning what it prints
    (FluidSequential.c:323-324): mean seconds per step and per Jacobi sweep."""
    solver.synchronize()
    solver.timing_enable(True)
    solver.timing_read(reset=True)
    t0 = time.perf_counter()
    for z in range(steps):
        solver.step(1, use_sources=(first_uses_sources and z == 0), dt=dt, diff=diff, visc=visc, iters=iters)
    solver.synchronize()
    wall = time.perf_counter() - t0
    t = solver.timing_read(reset=True)
    solver.timing_enable(False)
    return {"Tot": wall / steps, "Diffusion": t["jacobi_ms"] * 1e-3 / max(t["sweeps"], 1), "sweeps": t["sweeps"]}
