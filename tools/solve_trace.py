#!/usr/bin/env python3
"""Profiling aid: repeated stand-alone solves (meant to run under `rocprofv3 --kernel-trace`), to compare the launches of
one solve with one another.   python tools/solve_trace.py [grid] [sweeps] [max sweeps per launch] [form: p|g]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fluidsimulationcuda_amd as F  # noqa: E402
from fluidsimulationcuda_amd import capi  # noqa: E402

grid = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
sweeps = int(sys.argv[2]) if len(sys.argv) > 2 else 48
max_t = int(sys.argv[3]) if len(sys.argv) > 3 else 12
form = sys.argv[4] if len(sys.argv) > 4 else "p"
n = grid - 2
al, be = (1.0, 4.0) if form == "p" else F.coefficients(n, 0.016, 0.0025)
rng = np.random.default_rng(0)
x = rng.uniform(-1, 1, (n + 2, n + 2)).astype(np.float32)
with F.FluidSolver(n, params={capi.PARAM_TB_MAX_SWEEPS: max_t}) as s:
    s.upload(u=x, v=x)
    for _ in range(12):
        s.diffuse(0, "u", "v", al, be, sweeps)
        s.synchronize()
        if s.autotune_pending() == 0:
            break
    for _ in range(5):
        s.diffuse(0, "u", "v", al, be, sweeps)
    s.synchronize()
