#!/usr/bin/env python3
"""Strip-height sweep of the fused Jacobi kernel on ONE slab of a P-way split (no-op exchange: the numbers the kernels
produce are meaningless, their timing is what one rank of a real run sees).  For each strip height: us per sweep of a
40-sweep solve, pressure form (16 sweeps per launch where allowed) and general form (8).
    python tools/slab_rb_sweep.py [grid] [P ...]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fluidsimulationcuda_amd as F  # noqa: E402
from fluidsimulationcuda_amd import capi  # noqa: E402

grid = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
n = grid - 2
for P in [int(x) for x in sys.argv[2:]] or [8, 4]:
    with F.FluidSolver(n, rank=P // 2 - 1 if P > 1 else 0, nranks=P) as s:
        if P > 1:
            s.set_exchange(lambda kind, ids, depth, scalar: 0.6 if kind in (capi.XCHG_MAX, capi.XCHG_MAX_END) else None)
        lo, hi = s.owned_rows
        x = np.random.default_rng(0).random((n + 2, n + 2), dtype=np.float32)
        s.upload_rows("u", x, max(lo - 64, 0), min(hi + 64, n + 2))
        s.upload_rows("v", x, max(lo - 64, 0), min(hi + 64, n + 2))
        if os.environ.get("TB_T"):
            s.set_param(capi.PARAM_TB_MAX_SWEEPS, int(os.environ["TB_T"]))
            s.set_param(capi.PARAM_TB_T16_MIN_CELLS, 0)
        av, bv = F.coefficients(n, 0.016, 0.0025)
        sweeps = int(os.environ.get("SWEEPS", "40"))
        for form, b, alpha, beta in (("pressure", 0, 1.0, 4.0), ("general", 1, av, bv)):
            out = []
            for rb in (0, 32, 48, 56, 64, 72, 80, 88, 96, 104, 112, 128, 144, 160, 192, 256):
                s.set_param(capi.PARAM_TB_ROWS, rb)
                s.diffuse(b, "u", "v", alpha, beta, sweeps)
                s.timing_enable(True)
                s.timing_read(reset=True)
                for _ in range(4):
                    s.diffuse(b, "u", "v", alpha, beta, sweeps)
                t = s.timing_read(reset=True)
                s.timing_enable(False)
                out.append("%d:%.2f" % (rb, t["jacobi_ms"] * 1e3 / t["sweeps"]))
            print("grid %d, slab of %d (%d rows), %s form, %d launches per solve: us/sweep by strip rows: %s"
                  % (grid, P, hi - lo, form, t["jacobi_launches"] // 4, "  ".join(out)), flush=True)
