#!/usr/bin/env python3
"""One configuration of the fused Jacobi solve for profiling under rocprofv3 (kernel trace / PMC):
    python3 tools/prof_solve.py GRID FORM MAX_T FAST_DIV [DATA] [REPS]
FORM: general | pressure; DATA: random | zero.  Runs REPS solves of 48 sweeps on one field."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fluidsimulationcuda_amd as F  # noqa: E402
from fluidsimulationcuda_amd import capi  # noqa: E402

grid, form, max_t, fast = int(sys.argv[1]), sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
data = sys.argv[5] if len(sys.argv) > 5 else "random"
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 6
n = grid - 2
rng = np.random.default_rng(grid)
alpha, beta = (1.0, 4.0) if form == "pressure" else F.coefficients(n, 0.016, 0.0025)
with F.FluidSolver(n, params={capi.PARAM_TB_FAST_DIVISION: fast, capi.PARAM_TB_MAX_SWEEPS: max_t,
                              capi.PARAM_TB_T16_MIN_CELLS: 0}) as s:
    if data == "random":
        s.upload(u=rng.random((n + 2, n + 2), dtype=np.float32) - np.float32(0.5),
                 v=rng.random((n + 2, n + 2), dtype=np.float32) - np.float32(0.5))
    s.timing_enable(True)
    for _ in range(reps):
        s.diffuse(1 if form == "general" else 0, "u", "v", alpha, beta, 48)
    t = s.timing_read()
    print("%d^2 %s %s T<=%d fast_div=%d: %.2f us/sweep, %d launches" % (
        grid, form, data, max_t, fast, t["jacobi_ms"] * 1e3 / t["sweeps"], t["jacobi_launches"]), flush=True)
