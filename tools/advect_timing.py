#!/usr/bin/env python3
"""Tuning aid: the velocity self-advection (two fields per launch) and the density advection against the size of the
back-trace, i.e. how far the gathers land from home.   python tools/advect_timing.py [grid]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fluidsimulationcuda_amd as F  # noqa: E402
from fluidsimulationcuda_amd import capi  # noqa: E402

grid = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
n = grid - 2
rng = np.random.default_rng(0)
blk = rng.uniform(-1, 1, (1024, 1000)).astype(np.float32)
base = np.ascontiguousarray(np.tile(blk, ((n + 2 + 1023) // 1024, (n + 2 + 999) // 1000))[:n + 2, :n + 2])
# a smooth field as well: what velocities look like after 40 sweeps of viscosity
yy, xx = np.meshgrid(np.arange(n + 2, dtype=np.float32), np.arange(n + 2, dtype=np.float32), indexing="ij")
smooth = (np.sin(xx / 97.0) * np.cos(yy / 131.0)).astype(np.float32)
with F.FluidSolver(n) as s:
    for name, field in (("noise", base), ("smooth", smooth)):
        for scale in (0.0, 1e-4, 1e-2, 0.1, 0.6):
            u = (field * np.float32(scale)).astype(np.float32)
            s.upload(u_prev=u, v_prev=u.T.copy(), dens_prev=base)
            s.timing_enable(True)
            for _ in range(2):
                s.timing_read(reset=True)
                for _ in range(5):
                    capi.check(capi.lib().fluid_op_advect(s._h, 1, capi.U, capi.U_PREV, capi.U_PREV, capi.V_PREV, 0.016))
                    capi.check(capi.lib().fluid_op_advect(s._h, 0, capi.DENS, capi.DENS_PREV, capi.U_PREV, capi.V_PREV, 0.016))
                t = s.timing_read(reset=True)
            print("%d^2 %-6s velocity x %-6g (back-trace up to %6.1f cells): %.1f us per single-field advect" %
                  (grid, name, scale, 0.016 * n * scale, t["advection_ms"] * 1e3 / 10), flush=True)
