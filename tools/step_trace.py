#!/usr/bin/env python3
"""Profiling aid: a few device-resident steps on one GPU, meant to run under `rocprofv3 --kernel-trace` so that the
kernel timeline of a step (durations and the idle time between dependent launches) can be read off the trace.
    python tools/step_trace.py [grid] [steps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fluidsimulationcuda_amd as F  # noqa: E402
from fluidsimulationcuda_amd.harness import initialize_parameters  # noqa: E402

grid = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
n = grid - 2
with F.FluidSolver(n) as s:
    s.upload(**initialize_parameters(n, seed=1))
    s.step(1, use_sources=True)
    for _ in range(6):                      # let the strip-height tuner settle
        s.step(5)
        s.synchronize()
        if s.autotune_pending() == 0:
            break
    s.synchronize()
    s.step(steps)
    s.synchronize()
