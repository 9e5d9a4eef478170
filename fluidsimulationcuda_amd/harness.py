"""The reference's benchmark harness, restated for this package: synthetic
input (initializeParameters, FluidSequential.c:244-271) and the Z-step mean-time
loop with per-solve breakdown (:289-324).

The reference draws from unseeded glibc rand(); the GPU box must not depend on
a libc, so draws come from numpy's PCG64 with an explicit seed.  The recipe is
the reference's: density source k/1000 (k uniform in 0..99) inside the centred
square of half-width (N+2)/8 and 0 outside; velocity sources k/100 on all
(N+2)^2 cells; current fields zero.
"""
import time

import numpy as np

from .solver import DIFF, DT, ITERS, VIS


def initialize_parameters(n, seed=1):
    """-> dict(dens, dens_prev, u, u_prev, v, v_prev), each (n+2, n+2) float32."""
    w = n + 2
    c, r = w // 2, w // 8
    rng = np.random.default_rng(seed)
    dens_prev = np.zeros((w, w), dtype=np.float32)
    k = rng.integers(0, 100, size=(2 * r, 2 * r), dtype=np.int32)
    dens_prev[c - r:c + r, c - r:c + r] = k.astype(np.float32) / np.float32(1000.0)
    u_prev = rng.integers(0, 100, size=(w, w), dtype=np.int32).astype(np.float32) / np.float32(100.0)
    v_prev = rng.integers(0, 100, size=(w, w), dtype=np.int32).astype(np.float32) / np.float32(100.0)
    z = np.zeros((w, w), dtype=np.float32)
    return dict(dens=z, dens_prev=dens_prev, u=z.copy(), u_prev=u_prev, v=z.copy(), v_prev=v_prev)


def run_steps(solver, steps, dt=DT, diff=DIFF, visc=VIS, iters=ITERS, first_uses_sources=True):
    """Z steps as the reference's main does (FluidSequential.c:289-312), returning
    the means it prints (:323-324), in seconds: Tot per step; Source, Divergence,
    Advection, Projection per call; Diffusion per Jacobi SWEEP (the reference
    divides its diffusion time by 40).  Unlike the reference, which times only
    the first call of each kind per step, every call is timed (HIP events)."""
    solver.synchronize()
    solver.timing_enable(True)
    solver.timing_read(reset=True)
    t0 = time.perf_counter()
    for z in range(steps):
        solver.step(1, use_sources=(first_uses_sources and z == 0), dt=dt, diff=diff, visc=visc, iters=iters)
    solver.synchronize()
    wall = time.perf_counter() - t0
    t = solver.timing_read(reset=True)
    solver.timing_enable(False)

    def per_call(name):
        return t[name + "_ms"] * 1e-3 / max(t[name + "_calls"], 1)

    return {"Tot": wall / steps, "Source": per_call("source"),
            "Diffusion": t["jacobi_ms"] * 1e-3 / max(t["sweeps"], 1), "Divergence": per_call("divergence"),
            "Advection": per_call("advection"), "Projection": per_call("projection"), "sweeps": t["sweeps"]}


def format_report(r):
    """The reference's closing printf (FluidSequential.c:323-324), same labels and order."""
    return "Tot %f\nSource %f\nDiffusion %f\nDivergence %f\nAdvection %f\nProjection %f\n" % (
        r["Tot"], r["Source"], r["Diffusion"], r["Divergence"], r["Advection"], r["Projection"])


# ---- state dump / load (the reference keeps state only in memory) ----
STATE_FIELDS = ("u", "v", "dens", "u_prev", "v_prev", "dens_prev")
_MAGIC = b"FLUIDF32"


def save_state(solver, path, step_index=0):
    """Raw little-endian dump: magic, int32 N, int32 step index, then the six
    (N+2)^2 float32 fields in the order of the reference's main (:277-282)."""
    with open(path, "wb") as f:
        f.write(_MAGIC)
        np.array([solver.n, step_index], dtype="<i4").tofile(f)
        for name in STATE_FIELDS:
            solver.download(name).astype("<f4").tofile(f)


def load_state(solver, path):
    """Inverse of save_state; returns the stored step index."""
    with open(path, "rb") as f:
        if f.read(8) != _MAGIC:
            raise ValueError("%s is not a fluid state file" % path)
        n, step_index = np.fromfile(f, dtype="<i4", count=2)
        if n != solver.n:
            raise ValueError("state is for N=%d, solver has N=%d" % (n, solver.n))
        w = n + 2
        for name in STATE_FIELDS:
            a = np.fromfile(f, dtype="<f4", count=w * w)
            if a.size != w * w:
                raise ValueError("truncated state file")
            solver.upload(**{name: a.reshape(w, w)})
    return int(step_index)


def print_state_grid(dens, u, v, out=None):
    """Text dump in the layout of the reference's printStateGrid (FluidSequential.c:32-52)."""
    import sys
    out = out or sys.stdout
    out.write("---------------------------------------\nDENSITY\n")
    for row in dens:
        out.write("".join("[%f] " % x for x in row) + "\n")
    out.write("\n\nVELOCITY\n")
    for ru, rv in zip(u, v):
        out.write("".join("[%f, %f] " % (a, b) for a, b in zip(ru, rv)) + "\n")


def main(argv=None):
    """python -m fluidsimulationcuda_amd.harness [--grid W] [--steps Z]: the reference's
    main() on the MI355X -- synthetic input, Z steps, its closing report."""
    import argparse
    from .solver import FluidSolver
    ap = argparse.ArgumentParser()
    ap.add_argument("--grid", type=int, default=4096)
    ap.add_argument("--steps", type=int, default=50)        # Z, FluidSequential.c:10
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args(argv)
    n = a.grid - 2
    with FluidSolver(n) as s:
        s.upload(**initialize_parameters(n, a.seed))
        s.step(1, use_sources=True)                          # warm-up incl. one-time division proofs
        s.upload(**initialize_parameters(n, a.seed))
        print(format_report(run_steps(s, a.steps)), end="")


if __name__ == "__main__":
    main()
