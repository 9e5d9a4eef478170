#!/usr/bin/env python3
"""General-form (diffusion) Jacobi solve: us per field-sweep by division mode and launch depth, on ordinary
(random) data and on all-zero data.  FAST_DIVISION 2 = scaled residual correction (mode 5, the default), 3 = double
reciprocal (mode 2), 1 = two-term reciprocal where x0 allows (mode 3), 0 = true division.
    python tools/div_timing.py [grid ...]      (default 4096 8192)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fluidsimulationcuda_amd as F  # noqa: E402
from fluidsimulationcuda_amd import capi  # noqa: E402

for grid in [int(g) for g in sys.argv[1:]] or [4096, 8192]:
    n = grid - 2
    rng = np.random.default_rng(grid)
    x = rng.random((n + 2, n + 2), dtype=np.float32) - np.float32(0.5)
    x0 = rng.random((n + 2, n + 2), dtype=np.float32) - np.float32(0.5)
    alpha, beta = F.coefficients(n, 0.016, 0.0025)
    for data in ("random", "zero"):
        for max_t in (8, 12, 16):
            for fast in (2, 3, 1, 0):
                with F.FluidSolver(n, params={capi.PARAM_TB_FAST_DIVISION: fast, capi.PARAM_TB_MAX_SWEEPS: max_t,
                                              capi.PARAM_TB_T16_MIN_CELLS: 0}) as s:
                    if data == "random":
                        s.upload(u=x, v=x0)
                    for _ in range(4):
                        s.diffuse(1, "u", "v", alpha, beta, 48)      # warm-up: the division proof, the strip-height tuner
                    if data == "random":
                        s.upload(u=x, v=x0)
                    s.timing_enable(True)
                    s.timing_read(reset=True)
                    for _ in range(3):
                        s.diffuse(1, "u", "v", alpha, beta, 48)
                    t = s.timing_read()
                    print("%d^2 %-6s T<=%-2d fast_div=%d: %.2f us/sweep, %d launches" % (
                        grid, data, max_t, fast, t["jacobi_ms"] * 1e3 / t["sweeps"], t["jacobi_launches"]), flush=True)
