#!/usr/bin/env python3
"""Large-grid cross-check: a 40-sweep solve through the fused kernel (16 + 16 + 8 sweeps per launch) must equal 40
single-sweep launches bit for bit, for the pressure form (alpha 1, beta 4) and the general form.
    python tools/check_large.py [grid ...]      (default 8192 16384)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fluidsimulationcuda_amd as F  # noqa: E402

for grid in [int(g) for g in sys.argv[1:]] or [8192, 16384]:
    n = grid - 2
    rng = np.random.default_rng(grid)
    x = rng.random((n + 2, n + 2), dtype=np.float32) - np.float32(0.5)
    x0 = rng.random((n + 2, n + 2), dtype=np.float32) - np.float32(0.5)
    for b, (alpha, beta) in ((0, (1.0, 4.0)), (1, F.coefficients(n, 0.016, 0.0025))):
        outs = []
        for variant in (0, 3):
            with F.FluidSolver(n, jacobi=variant) as s:
                s.upload(u=x, v=x0)
                s.timing_enable(True)
                s.diffuse(b, "u", "v", alpha, beta, 40)
                t = s.timing_read()
                outs.append(s.download("u"))
                print("%d^2 b=%d beta=%g variant %d: %.2f us/sweep, %d launches" % (
                    grid, b, beta, variant, t["jacobi_ms"] * 1e3 / t["sweeps"], t["jacobi_launches"]), flush=True)
        same = np.array_equal(outs[0].view(np.uint32), outs[1].view(np.uint32))
        print("%d^2 b=%d: fused == single-sweep launches: %s" % (grid, b, same), flush=True)
        assert same
