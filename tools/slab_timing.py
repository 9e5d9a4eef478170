#!/usr/bin/env python3
"""Tuning aid: what ONE rank of an N-GPU run computes, timed on one GPU.
A context configured as rank r of P with a no-op exchange callback executes the
same kernels on the same row ranges as in a real run (the numbers it produces
are meaningless: halos are never filled).   python tools/slab_timing.py [grid] [P] [rows...]
environment: HALO (ghost-zone depth), STORAGE (1: fp16 fields), T16MIN (FLUID_PARAM_TB_T16_MIN_CELLS), TB_T (most sweeps per launch),
OVERLAP (FLUID_PARAM_XCHG_OVERLAP)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fluidsimulationcuda_amd as F  # noqa: E402
from fluidsimulationcuda_amd import capi  # noqa: E402

grid = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
P = int(sys.argv[2]) if len(sys.argv) > 2 else 8
rows_list = [int(x) for x in sys.argv[3:]] or [0]
n = grid - 2
calls = {"halo": 0, "max": 0}


def xchg(kind, ids, depth, scalar):
    if kind in (capi.XCHG_MAX, capi.XCHG_MAX_END):
        calls["max"] += 1
        return 0.6          # a typical max |velocity| of the synthetic workload: ~80-row advect halo at 8192
    if kind == capi.XCHG_HALO:
        calls["halo"] += 1
    return None


with F.FluidSolver(n, rank=P // 2 - 1 if P > 1 else 0, nranks=P, halo=int(os.environ.get("HALO", "0")),
                   storage=int(os.environ.get("STORAGE", "0"))) as s:
    if P > 1:
        s.set_exchange(xchg)
    rng = np.random.default_rng(0)
    lo, hi = s.owned_rows
    x = np.zeros((n + 2, n + 2), np.float32)
    x[max(lo - 64, 0):hi + 64] = rng.random((min(hi + 64, n + 2) - max(lo - 64, 0), n + 2), dtype=np.float32)
    for name in ("u", "v", "dens"):
        s.upload_rows(name, x, max(lo - 64, 0), min(hi + 64, n + 2))
    if os.environ.get("T16MIN"):
        s.set_param(capi.PARAM_TB_T16_MIN_CELLS, int(os.environ["T16MIN"]))
    if os.environ.get("TB_T"):
        s.set_param(capi.PARAM_TB_MAX_SWEEPS, int(os.environ["TB_T"]))
    if os.environ.get("OVERLAP"):
        s.set_param(capi.PARAM_XCHG_OVERLAP, int(os.environ["OVERLAP"]))
    for rows in rows_list:
        s.set_param(capi.PARAM_TB_ROWS, rows)
        s.step(2)
        for _ in range(6):                  # let the library's strip-height tuner finish (rows = 0)
            if rows or s.autotune_pending() == 0:
                break
            s.step(5)
            s.synchronize()
        s.synchronize()
        s.timing_enable(True)
        s.timing_read(reset=True)
        calls["halo"] = calls["max"] = 0
        t0 = time.perf_counter()
        K = 10
        s.step(K)
        s.synchronize()
        wall = (time.perf_counter() - t0) / K
        t = s.timing_read(reset=True)
        s.timing_enable(False)
        gpu = sum(t[k + "_ms"] for k in capi.TIMING_CATEGORIES) / K
        print("grid %d, rank of %d (rows %d..%d), strip rows %3d: %.3f ms/step wall (timing on), %.3f ms/step in kernels "
              "(%s), %.2f us per Jacobi sweep, %d halo + %d max exchanges per step"
              % (grid, P, lo, hi, rows, wall * 1e3, gpu, ", ".join("%s %.3f" % (k, t[k + "_ms"] / K) for k in capi.TIMING_CATEGORIES),
                 t["jacobi_ms"] * 1e3 / t["sweeps"], calls["halo"] // K, calls["max"] // K), flush=True)
        # the same without the timing events (they add host work and stream markers)
        s.step(2)
        s.synchronize()
        t0 = time.perf_counter()
        s.step(K)
        t_enq = time.perf_counter() - t0
        s.synchronize()
        print("   timing off: %.3f ms/step wall, host enqueue alone %.3f ms/step" % ((time.perf_counter() - t0) / K * 1e3,
                                                                                    t_enq / K * 1e3), flush=True)
