#!/usr/bin/env python3
"""Host-array drop-in step()/step_src() (include/fluid_amd.h): wall time per call at a few grids, next to the
device-resident step and to what the bytes alone cost over PCIe (pinned-memory copies of the same size, measured here).
    python tools/step_timing.py [grid ...]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fluidsimulationcuda_amd as F  # noqa: E402
import torch  # noqa: E402

for grid in [int(g) for g in sys.argv[1:]] or [1024, 4096, 8192]:
    n = grid - 2
    rng = np.random.default_rng(0)
    u, v, d, u0, v0, d0 = (rng.random((grid, grid), dtype=np.float32) for _ in range(6))
    F.step_src(n, 0.016, 0.1, 0.0025, 40, u, v, d, u0, v0, d0)          # creates the cached context, proves the betas
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        F.step(n, 0.016, 0.1, 0.0025, u, v, d)
    t_step = (time.perf_counter() - t0) / reps
    t0 = time.perf_counter()
    for _ in range(reps):
        F.step_src(n, 0.016, 0.1, 0.0025, 40, u, v, d, u0, v0, d0)
    t_src = (time.perf_counter() - t0) / reps
    # PCIe reference: 3 fields up + 3 down through pinned buffers
    pin = [torch.empty((grid, grid), dtype=torch.float32).pin_memory() for _ in range(3)]
    dev = [torch.empty((grid, grid), dtype=torch.float32, device="cuda") for _ in range(3)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        for a, b in zip(pin, dev):
            b.copy_(a, non_blocking=True)
        for a, b in zip(pin, dev):
            a.copy_(b, non_blocking=True)
        torch.cuda.synchronize()
    t_pcie = (time.perf_counter() - t0) / reps
    with F.FluidSolver(n) as s:
        s.upload(u=u, v=v, dens=d)
        s.step(2)
        s.synchronize()
        t0 = time.perf_counter()
        s.step(10)
        s.synchronize()
        t_dev = (time.perf_counter() - t0) / 10
    mib = 6 * grid * grid * 4 / 2 ** 20
    print("%5d^2: step() %7.2f ms   step_src() %7.2f ms   resident step %6.2f ms   pinned PCIe for step()'s %d MiB: %6.2f ms (%.1f GB/s)  -> step() = %.2f x (PCIe + compute)"
          % (grid, t_step * 1e3, t_src * 1e3, t_dev * 1e3, mib, t_pcie * 1e3, mib * 2 ** 20 / t_pcie / 1e9,
             t_step / (t_pcie + t_dev)), flush=True)
F.capi.check(F.capi.lib().fluid_release_cached())
