#!/usr/bin/env python3
"""Tuning aid: time 40-sweep solves of the temporally blocked Jacobi kernel over
strip heights (FLUID_PARAM_TB_ROWS) and sweeps per launch, one process, HIP
events via the library's timing API.   python tools/tb_sweep.py [grid] [T...]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fluidsimulationcuda_amd as F  # noqa: E402
from fluidsimulationcuda_amd import capi  # noqa: E402

grid = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
Ts = [int(x) for x in sys.argv[2:]] or [4, 8]
n = grid - 2
rng = np.random.default_rng(0)
x = rng.random((n + 2, n + 2), dtype=np.float32)
with F.FluidSolver(n, jacobi=capi.JACOBI_TB) as s:
    s.upload(u=x, v=x)
    a, b = F.coefficients(n, 0.016, 0.0025)
    if os.environ.get('TB_NV'):
        s.set_param(capi.PARAM_TB_LANE_COLUMNS, int(os.environ['TB_NV']))
    nv = int(os.environ.get('TB_NV', '2'))
    if os.environ.get('TB_EDGE'):
        s.set_param(capi.PARAM_TB_EDGE_ROWS_PCT, int(os.environ['TB_EDGE']))
    for T in Ts:
        s.set_param(capi.PARAM_TB_MAX_SWEEPS, T)
        HL = (T + nv - 1) // nv
        nwin = -(-((n + nv - 1) // nv) // (64 - 2 * HL))
        for rows in ([int(r) for r in os.environ['TB_ROWS'].split(',')] if os.environ.get('TB_ROWS') else list(range(8, 41, 2)) + [44, 48, 56, 64, 96, 128]):
            s.set_param(capi.PARAM_TB_ROWS, rows)
            out = []
            for alpha, beta in ((a, b), (1.0, 4.0)):
                s.diffuse(0, "u", "v", alpha, beta, 40)      # warm
                s.timing_enable(True)
                s.timing_read(reset=True)
                for _ in range(3):
                    s.diffuse(0, "u", "v", alpha, beta, 40)
                t = s.timing_read(reset=True)
                s.timing_enable(False)
                out.append(t["jacobi_ms"] * 1e3 / t["sweeps"])
            strips = -(-n // rows) if rows else 0      # rows == 0: the library's own choice
            blocks = nwin * (-(-strips // 4))
            print("T=%d rows=%3d  waves=%5d blocks=%4d  us/sweep: div %.2f  mul %.2f" % (
                T, rows, strips * nwin, blocks, out[0], out[1]), flush=True)
