#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REFERENCE ITSELF.

Runs only in the build container: it drives project/sequential/FluidSequential.c
compiled by oracle/build_ref.sh (oracle/_ref/*.so).  The reference holds no
tests or fixtures of its own, so these vectors are what
pins parity.  Fixtures are data only: inputs drawn from numpy's PCG64 with the
seeds below (or from the reference's own initializeParameters), and the arrays
the reference produced from them.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.oracle import DIFF, DT, VISC, Reference, build_ref  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def coeff(n, coef):
    """alpha, beta as the reference evaluates them (FluidSequential.c:179-180)."""
    f = np.float32
    a = f(f(f(f(DT) * f(coef)) * f(n)) * f(n))
    return a, f(f(1) + f(f(4) * a))


def rnd(rng, n, lo=-1.0, hi=1.0):
    return rng.uniform(lo, hi, size=(n + 2, n + 2)).astype(np.float32)


def fnv1a(a):
    """64-bit FNV-1a over 32-bit words of the array (vectorised per word)."""
    h = np.uint64(0xCBF29CE484222325)
    p = np.uint64(0x100000001B3)
    w = np.ascontiguousarray(a).view(np.uint32).ravel().astype(np.uint64)
    # fold in chunks: hash of per-chunk xor-multiply chain is order dependent,
    # so do it in pure integer arithmetic on python ints for exactness
    hv = int(h)
    pv = int(p)
    mask = (1 << 64) - 1
    for x in w.tolist():
        hv = ((hv ^ x) * pv) & mask
    return hv


def operators(n):
    build_ref(["%d:40" % n, "%d:2" % n])
    r40, r2 = Reference(n, 40), Reference(n, 2)
    rng = np.random.default_rng(1000 + n)
    g = {}
    # --- set_bnd, b = 0, 1, 2 (bit-exact contract)
    x = rnd(rng, n)
    g["bnd_in"] = x
    for b in (0, 1, 2):
        y = x.copy()
        r40.set_bnd(b, y)
        g["bnd_out_b%d" % b] = y
    # --- add_source
    x, s = rnd(rng, n), rnd(rng, n)
    g["src_x"], g["src_s"] = x, s
    y = x.copy()
    r40.add_source(y, s)
    g["src_out"] = y
    # --- diffuse: the four (b, alpha, beta) uses of the step, 2 and 40 sweeps
    av, bv = coeff(n, VISC)
    ad, bd = coeff(n, DIFF)
    cases = [(1, av, bv), (2, av, bv), (0, ad, bd), (0, np.float32(1), np.float32(4))]
    g["dif_params"] = np.array([[b, a, be] for b, a, be in cases], dtype=np.float64)
    for k, (b, a, be) in enumerate(cases):
        x, x0 = rnd(rng, n), rnd(rng, n)
        g["dif%d_x" % k], g["dif%d_x0" % k] = x, x0
        for ref, tag in ((r2, 2), (r40, 40)):
            y = x.copy()
            ref.diffuse(b, y, x0.copy(), float(a), float(be))
            g["dif%d_out%d" % (k, tag)] = y
    # --- divergence + pressure clear
    u, v = rnd(rng, n), rnd(rng, n)
    p, div = rnd(rng, n), rnd(rng, n)
    g["div_u"], g["div_v"] = u, v
    g["div_p_in"], g["div_div_in"] = p.copy(), div.copy()
    r40.divergence(u, v, p, div)
    g["div_p"], g["div_div"] = p, div
    # --- gradient subtraction
    u, v, p = rnd(rng, n), rnd(rng, n), rnd(rng, n)
    g["grad_u"], g["grad_v"], g["grad_p"] = u.copy(), v.copy(), p
    r40.subtract_gradient(u, v, p)
    g["grad_u_out"], g["grad_v_out"] = u, v
    # --- advect: small velocities, and velocities big enough to hit all four
    #     clamps (dt0 = DT*n, so |vel| up to 3/DT/... covers the whole grid)
    for tag, amp in (("small", 0.05), ("clamp", 4.0 / DT)):
        u, v = rnd(rng, n, -amp, amp), rnd(rng, n, -amp, amp)
        d0 = rnd(rng, n)
        g["adv_%s_u" % tag], g["adv_%s_v" % tag], g["adv_%s_d0" % tag] = u, v, d0
        for b in (0, 1, 2):
            d = rnd(rng, n)          # stale contents must not matter
            r40.advect(b, d, d0, u, v)
            g["adv_%s_out_b%d" % (tag, b)] = d
        # self-advection aliasing of the velocity step: d0 is u (b=1), v (b=2)
        d = rnd(rng, n)
        r40.advect(1, d, u, u, v)
        g["adv_%s_self_u" % tag] = d
        d = rnd(rng, n)
        r40.advect(2, d, v, u, v)
        g["adv_%s_self_v" % tag] = d
    np.savez_compressed(os.path.join(OUT, "ops_n%d.npz" % n), **g)


def full_steps(n, iters=40, steps=(1, 2, 5)):
    build_ref(["%d:%d" % (n, iters)])
    r = Reference(n, iters)
    dens, dens0, u, u0, v, v0 = r.initialize(seed=1)
    g = {"init_u_prev": u0.copy(), "init_v_prev": v0.copy(), "init_dens_prev": dens0.copy()}
    for z in range(1, max(steps) + 1):
        if z == 1:
            r.step_src(u, v, dens, u0, v0, dens0)
            # the buffers the reference leaves in *_prev: p, div, diffused dens
            g["s1_u_prev"], g["s1_v_prev"], g["s1_dens_prev"] = u0.copy(), v0.copy(), dens0.copy()
        else:
            r.step(u, v, dens, u0, v0, dens0)
        if z in steps:
            g["s%d_u" % z], g["s%d_v" % z], g["s%d_dens" % z] = u.copy(), v.copy(), dens.copy()
    np.savez_compressed(os.path.join(OUT, "step_n%d_k%d.npz" % (n, iters)), **g)


def state_grid(n=14):
    """The reference's own printStateGrid (FluidSequential.c:32-52) on its state after one step from
    initializeParameters: stdout of the compiled reference captured at file-descriptor level, plus the three
    fields it printed.  Pins harness.print_state_grid's text layout."""
    import ctypes
    import tempfile
    build_ref(["%d:40" % n])
    r = Reference(n, 40)
    r.lib.printStateGrid.restype, r.lib.printStateGrid.argtypes = None, [np.ctypeslib.ndpointer(np.float32, flags="C")] * 3
    dens, dens0, u, u0, v, v0 = r.initialize(seed=1)
    r.step_src(u, v, dens, u0, v0, dens0)
    libc = ctypes.CDLL(None)
    sys.stdout.flush()
    libc.fflush(None)
    with tempfile.TemporaryFile() as tmp:
        saved = os.dup(1)
        os.dup2(tmp.fileno(), 1)
        try:
            r.lib.printStateGrid(dens, u, v)
            libc.fflush(None)
        finally:
            os.dup2(saved, 1)
            os.close(saved)
        tmp.seek(0)
        text = tmp.read().decode()
    with open(os.path.join(OUT, "state_grid_n%d.txt" % n), "w") as f:
        f.write(text)
    np.savez_compressed(os.path.join(OUT, "state_grid_n%d.npz" % n), dens=dens, u=u, v=v)
    print("printStateGrid: %d bytes captured" % len(text))


def checksums():
    """Too big to commit as arrays: CRC-32 of the bytes + float64 sums of step 1 from the
    reference's own initializeParameters (glibc rand, seed 1); FNV-1a too where the
    pure-python hash is affordable."""
    import zlib
    rows = []
    for n in (254, 1022, 4094, 8190, 16382):
        r = Reference(n, 40)
        dens, dens0, u, u0, v, v0 = r.initialize(seed=1)
        r.step_src(u, v, dens, u0, v0, dens0)
        c = (n + 2) // 2
        row = dict(n=n, crc_u=zlib.crc32(u.view(np.uint8).reshape(-1)), crc_v=zlib.crc32(v.view(np.uint8).reshape(-1)),
                   crc_dens=zlib.crc32(dens.view(np.uint8).reshape(-1)),
                   sum_u=float(u.sum(dtype=np.float64)), sum_v=float(v.sum(dtype=np.float64)),
                   sum_dens=float(dens.sum(dtype=np.float64)),
                   u_c=float(u[c, c]), dens_c=float(dens[c, c]))
        if n <= 1022:
            row.update(fnv_u=fnv1a(u), fnv_v=fnv1a(v), fnv_dens=fnv1a(dens))
        rows.append(row)
        print(rows[-1])
    import json
    with open(os.path.join(OUT, "checksums.json"), "w") as f:
        json.dump(rows, f, indent=1)


TRAJECTORIES = ((1022, 10), (4094, 5), (8190, 3), (16382, 2))


def trajectory_checksums(only=None):
    """The reference's own loop over several steps (FluidSequential.c:289-324: sources at step 0 only, zeroed before every
    later step) at the grids where arrays are too big to commit: CRC-32 of u, v and dens after EVERY step, so that the
    decay of the fields towards zero is pinned to the reference itself, not only its first step.  `only` = (n, steps):
    that grid alone, merged into the file (16382^2 takes minutes per step on one core)."""
    import json
    import zlib
    rows = []
    path = os.path.join(OUT, "trajectory_checksums.json")
    if only is not None:
        rows = [r for r in json.load(open(path)) if r["n"] != only[0]]
    for n, steps in ((only,) if only is not None else TRAJECTORIES):
        r = Reference(n, 40)
        dens, dens0, u, u0, v, v0 = r.initialize(seed=1)
        for z in range(1, steps + 1):
            if z == 1:
                r.step_src(u, v, dens, u0, v0, dens0)
            else:
                r.step(u, v, dens, u0, v0, dens0)
            rows.append(dict(n=n, step=z, crc_u=zlib.crc32(u.view(np.uint8).reshape(-1)), crc_v=zlib.crc32(v.view(np.uint8).reshape(-1)),
                             crc_dens=zlib.crc32(dens.view(np.uint8).reshape(-1)), max_u=float(np.abs(u).max())))
            print(rows[-1])
    rows.sort(key=lambda r: (r["n"], r["step"]))
    with open(path, "w") as f:
        json.dump(rows, f, indent=1)


if __name__ == "__main__":
    if sys.argv[1:] == ["checksums"]:
        checksums()
        sys.exit(0)
    if sys.argv[1:2] == ["trajectory"]:
        trajectory_checksums(tuple(int(v) for v in sys.argv[2:4]) if len(sys.argv) >= 4 else None)
        sys.exit(0)
    if sys.argv[1:] == ["state_grid"]:
        state_grid()
        sys.exit(0)
    for n in (14, 30, 61):
        operators(n)
    for n in (30, 61, 126):
        full_steps(n)
    full_steps(126, iters=20, steps=(1, 2))
    state_grid()
    checksums()
    trajectory_checksums()
    print("golden vectors written to", OUT)
