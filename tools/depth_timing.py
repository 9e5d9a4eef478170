#!/usr/bin/env python3
"""Tuning aid: us per sweep of stand-alone solves at one launch depth, per grid size and form.
    python tools/depth_timing.py T grid [grid ...]     (T = sweeps per launch; 48 sweeps per solve)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fluidsimulationcuda_amd as F  # noqa: E402
from fluidsimulationcuda_amd import capi  # noqa: E402

T = int(sys.argv[1])
for grid in [int(g) for g in sys.argv[2:]]:
    n = grid - 2
    x = np.random.default_rng(0).uniform(-1, 1, (n + 2, n + 2)).astype(np.float32)
    for form, (al, be) in (("pressure", (1.0, 4.0)), ("viscosity", F.coefficients(n, 0.016, 0.0025))):
        with F.FluidSolver(n, params={capi.PARAM_TB_MAX_SWEEPS: T, capi.PARAM_TB_T16_MIN_CELLS: 0}) as s:
            s.upload(u=x, v=x)
            iters = 48
            for _ in range(14):
                s.diffuse(0, "u", "v", al, be, iters)
                s.synchronize()
                if s.autotune_pending() == 0:
                    break
            s.timing_enable(True)
            s.timing_read(reset=True)
            for _ in range(10):
                s.diffuse(0, "u", "v", al, be, iters)
            t = s.timing_read(reset=True)
            us = t["jacobi_ms"] * 1e3 / t["sweeps"]
            print("%5d^2 %-9s T=%2d: %7.3f us/sweep, %.3f ns per 1000 cells, %d launches/solve" %
                  (grid, form, T, us, us * 1e6 / (grid * grid), t["jacobi_launches"] // 10), flush=True)
