#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on MI355X: Mcells/s per Jacobi iteration
and ms per simulation step of the Stable-Fluids vel_step + dens_step.

  python bench.py [--gpus N --steps K --warmup W]

With N > 1 and no WORLD_SIZE in the environment the command starts its own N ranks as child processes
(python -m torch.distributed.run ... bench.py, before anything here touches the GPU) and relays rank 0's line;
started under torch.distributed.run (as the driver does) it is one of those ranks.

`value` follows SURVEY.md 8(d): W^2 / t_sweep with t_sweep the mean time of one sweep (set_bnd included) of the
40-sweep pressure solves (alpha 1, beta 4, b 0) inside the timed steps, HIP events on the solver's stream;
`all_solves` is the same rate over all 200 sweeps of a step, and the `roofline` object prices ALL Jacobi launches.

A "step" is one loop body of the reference's main (FluidSequential.c:298-306):
sources zeroed, vel_step, dens_step, 40 Jacobi sweeps per solve = 200 sweeps,
3 advects, 2 projections.  Fields are resident in HBM when the timed region
starts.  One JSON line on rank 0.

Workload: N=1 -> 4096^2 (the grid BASELINE.json's metric is quoted on, config
2); N>1 -> 8192^2 split into row slabs (config 3), strong scaling, with the
4096^2 grid on the same slabs beside it ("grid_4096": north_star asks for both
grids at 1/2/4/8 GPUs).  The N=1 line carries the 1-GPU 8192^2 measurement
("scaling_base"), the N>1 line rank 0's own 1-GPU run of both grids
("single_gpu"), so either speed-up can be formed from one line.

Two data regimes are reported.  The headline follows the reference's loop: sources only at step 0, zeroed
afterwards (FluidSequential.c:298-302), so every solve restarts from a zero first guess and the fields decay
by 1-2 orders of magnitude per step -- the timed steps run on small, then tiny values.  `value_ordinary_data`
repeats the measurement with the synthetic sources re-injected before every step (device-to-device copies,
timed apart), i.e. on fields of ordinary magnitude: the chip holds a lower clock on such data and the
diffusion solves can use their cheaper exact division there.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# multi-process GPU work on this platform needs dmabuf IPC (RCCL / tensor sharing fail otherwise)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

KERNELS = ["stream", "lds", "naive", "tb"]
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.3 TB/s achievable)
# what one SIMD sustains of the fused kernel's instruction mix (packed f32, DPP adds, f64 / conversions: 4 cycles
# each), in wave-instructions per microsecond with >= 4 waves resident: tools/ubench/valu_peak.hip on MI355X
VALU_PEAK_PER_SIMD_US = 570.0
SIMDS = 1024
BYTES_PER_CELL_SWEEP = 12      # read x + read x0 + write x_new, fp32 (6 with fp16 storage)
BYTES_PER_CELL_STEP = 2548     # SURVEY.md 8(d), 40 sweeps/solve


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--grid", type=int, default=0, help="grid width W=N+2 (default 4096 on 1 GPU, 8192 on several)")
    ap.add_argument("--iters", type=int, default=40)
    ap.add_argument("--variant", type=int, default=3, help="Jacobi kernel: 0 stream, 1 LDS-tiled, 2 naive, 3 temporally blocked")
    ap.add_argument("--tb-sweeps", type=int, default=0, help="temporal blocking: most sweeps per launch (16, 8, 4, 2; 0 = default)")
    ap.add_argument("--tb-rows", type=int, default=0, help="temporal blocking: rows per wave strip (0 = auto)")
    ap.add_argument("--fast-division", type=int, default=-1, help="FLUID_PARAM_TB_FAST_DIVISION (default: library's)")
    ap.add_argument("--t16-min-cells", type=int, default=-1, help="FLUID_PARAM_TB_T16_MIN_CELLS (default: library's rule)")
    ap.add_argument("--halo", type=int, default=0, help="multi-GPU ghost-zone depth (0 = library default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-scaling-base", action="store_true", help="skip the other side of the speed-up (N=1: the 8192^2 run; "
                                                                   "N>1: rank 0's single-GPU runs)")
    ap.add_argument("--no-second-grid", action="store_true", help="N>1: skip the 4096^2 leg")
    ap.add_argument("--overlap", type=int, default=-1, choices=[-1, 0, 1],
                    help="N>1: FLUID_PARAM_XCHG_OVERLAP (halo exchange beside the interior strips of the solve it feeds): 1 on, 0 off, "
                         "-1 (default) a few untimed steps each way on every grid, the faster one kept for the timed steps")
    ap.add_argument("--no-ordinary", action="store_true", help="skip the value_ordinary_data leg")
    ap.add_argument("--only-ordinary", action="store_true", help="profiling aid: run only the ordinary-data leg")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--dtype", default="f32", choices=["f32", "f16"],
                    help="field storage: f32 (reference arithmetic, the measured configuration) or f16 "
                         "(BASELINE config 4: fp16 fields, fp32 arithmetic)")
    ap.add_argument("--backend", default="nccl", help="nccl (= RCCL; the measured path) or gloo (host-staged rehearsal "
                                                      "of the multi-process path when ranks outnumber GPUs)")
    ap.add_argument("--exchange", default="auto", choices=["auto", "rccl", "torch", "try-rccl"],
                    help="multi-GPU halo exchange: the library's own RCCL exchange (C++, no host wait) or "
                         "torch.distributed (auto: rccl with --backend nccl)")
    ap.add_argument("--check", action="store_true", help="multi-GPU: also verify bit equality with a 1-context run (small grids)")
    return ap.parse_args()


def timed_steps(solver, dist, world, steps, iters, before_step=None):
    """Exactly `steps` steps between barrier+synchronize pairs; returns the max-over-ranks seconds, Jacobi ms,
    pressure-solve ms and the timing record."""
    import torch
    solver.timing_enable(True)
    solver.timing_read(reset=True)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if before_step is None:
        solver.step(steps, iters=iters)
    else:
        for _ in range(steps):
            before_step()
            solver.step(1, use_sources=True, iters=iters)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    t = solver.timing_read(reset=True)
    solver.timing_enable(False)
    jac_ms, prs_ms = t["jacobi_ms"], t["pressure_ms"]
    if world > 1:
        buf = torch.tensor([elapsed, jac_ms, prs_ms], dtype=torch.float64,
                           device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(buf, op=dist.ReduceOp.MAX)
        elapsed, jac_ms, prs_ms = float(buf[0]), float(buf[1]), float(buf[2])
    return elapsed, jac_ms, prs_ms, t


def measure(solver, dist, world, steps, warmup, iters):
    """The reference's loop: sources at step 0 only.  W untimed steps, then exactly K timed ones."""
    solver.step(1, use_sources=True, iters=iters)          # z == 0 consumes the synthetic sources
    for _ in range(max(warmup - 1, 0)):
        solver.step(1, iters=iters)
    return timed_steps(solver, dist, world, steps, iters)


def pressure_solve_rate(solver, iters, reps):
    """SURVEY.md 8(d)(i) on its own: the mean time of one sweep (set_bnd included) of the pressure solve -- alpha 1,
    beta 4, b 0, first guess p = 0 -- measured over `reps` solves of `iters` sweeps on the fields the timed steps left
    behind: computeDivergenceAndPressure(u, v, p, div) as its own kernel (which also marks p zero), then the solve, HIP
    events around the solve only.  (Inside a step the divergence rides in the solve's first launch since round 2, so the
    in-step solve time is no longer sweeps alone; it is reported beside this as in_step_pressure_solve.)
    Returns (ms in solves, sweeps)."""
    solver.timing_enable(True)
    solver.timing_read(reset=True)
    for _ in range(reps):
        solver.computeDivergenceAndPressure("u", "v", "u_prev", "v_prev")
        solver.diffuse(0, "u_prev", "v_prev", 1.0, 4.0, iters)
    t = solver.timing_read(reset=True)
    solver.timing_enable(False)
    return t["jacobi_ms"], t["sweeps"]


def measure_ordinary(solver, fields, steps, warmup, iters):
    """Sources re-injected before every step (one GPU): device copies of the three source fields are written into
    u_prev / v_prev / dens_prev, then one step consumes them.  Returns timed_steps()'s tuple plus the time the
    copies alone take per step."""
    import torch
    src = {k: torch.from_numpy(fields[k]).to(solver.device) for k in ("u_prev", "v_prev", "dens_prev")}

    def inject():
        with torch.cuda.stream(solver.torch_stream):
            for k, t in src.items():
                solver.interior(k).copy_(t, non_blocking=True)

    for _ in range(max(warmup, 1)):
        inject()
        solver.step(1, use_sources=True, iters=iters)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        inject()
    torch.cuda.synchronize()
    copy_ms = (time.perf_counter() - t0) * 1e3 / steps
    return timed_steps(solver, None, 1, steps, iters, before_step=inject) + (copy_ms,)


def pmc_summary(grid):
    """The newest committed rocprofv3 PMC summary of this bench command at this grid
    (profiles/r*_<grid>_pmc.json, tools/profile_bench.sh + tools/summarize_profiles.py): separate
    --pmc passes for FETCH_SIZE, WRITE_SIZE (gfx950 corrections applied) and SQ_INSTS_VALU."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_%d_pmc.json" % grid))):
        try:
            best = (json.load(open(f)), os.path.basename(f))
        except (OSError, ValueError):
            continue
    return best


def pmc_per_launch(summary, kernel, key):
    """Mean of `key` per launch over all instantiations of `kernel`, weighted by how often each ran."""
    tot = cnt = 0.0
    for name, v in summary.items():
        if not isinstance(v, dict) or name.split("<")[0] != kernel or not v.get(key):
            continue
        tot += v[key] * v["launches_sampled"]
        cnt += v["launches_sampled"]
    return tot / cnt if cnt else None


def cpu_baseline(n, fields, iters):
    """The oracle leg (checker only): one full step and one 40-sweep pressure
    solve on ONE host core at the bench workload's N -- the reference itself
    (oracle/_ref, kind "reference") when its build for this N travelled here
    (it is git-ignored: a fresh clone reports the restatement, kind "port").  Plus the restatement's
    solve on all host cores (POSIX threads over row bands, oracle/fluid_oracle.c: fo_diffuse_mt)."""
    import numpy as np
    from oracle.oracle import Oracle, Reference, have_ref
    f = {k: v.copy() for k, v in fields.items()}
    use_ref = have_ref(n, iters)
    eng = Reference(n, iters) if use_ref else Oracle()
    w = n + 2
    t0 = time.perf_counter()
    if use_ref:
        eng.step_src(f["u"], f["v"], f["dens"], f["u_prev"], f["v_prev"], f["dens_prev"])
    else:
        eng.step_src(f["u"], f["v"], f["dens"], f["u_prev"], f["v_prev"], f["dens_prev"], iters=iters)
    t_step = time.perf_counter() - t0
    p, div = np.zeros((w, w), np.float32), f["v_prev"]
    t0 = time.perf_counter()
    if use_ref:
        eng.diffuse(0, p, div, 1.0, 4.0)
    else:
        eng.diffuse(0, p, div, 1.0, 4.0, iters)
    t_solve = time.perf_counter() - t0
    out = {"value": w * w / (t_solve / iters) / 1e6, "unit": "Mcells/s per Jacobi iter", "cores": 1,
           "kind": "reference" if use_ref else "port", "ms_per_step": t_step * 1e3,
           "sample": "1 full step (200 sweeps) + one %d-sweep pressure solve at %dx%d, gcc -O2, 1 thread"
                     % (iters, w, w),
           "note": "kind 'reference' = project/sequential itself from oracle/_ref (built in the dev container, "
                   "git-ignored, shipped with the snapshot); without it the bit-identical restatement runs (kind 'port')"}
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    if cores > 1:
        threads = min(cores, 64)
        p2 = np.zeros((w, w), np.float32)
        Oracle().diffuse_threaded(0, p2, div, 1.0, 4.0, 2, threads)          # spin the threads up once
        sweeps = 16
        t0 = time.perf_counter()
        Oracle().diffuse_threaded(0, p2, div, 1.0, 4.0, sweeps, threads)
        t_mt = (time.perf_counter() - t0) / sweeps
        out["all_cores"] = {"value": w * w / t_mt / 1e6, "unit": "Mcells/s per Jacobi iter", "cores": threads,
                            "kind": "port", "sample": "%d sweeps of the pressure solve, POSIX threads over row bands "
                                                      "(fo_diffuse_mt), %d threads of %d host cores" % (sweeps, threads, cores)}
    return out


def self_launch(a):
    """`python bench.py --gpus N` as typed: N ranks as CHILD processes (python -m torch.distributed.run, one per GPU, RCCL
    rendezvous on 127.0.0.1), started before this process has imported torch or touched the GPU -- never a re-exec of a
    process that has.  The children inherit stdout / stderr, so rank 0's JSON line is this command's output; the exit
    status is the launcher's."""
    import socket
    import subprocess
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))


def library_source_hash():
    """sha256 over the sources libfluid_amd.so is built from (csrc/*.hip, *.h, Makefile, include/fluid_amd.h): what ties a
    committed PMC summary to the kernels it measured (tools/summarize_profiles.py stores the same hash)."""
    import hashlib
    h = hashlib.sha256()
    src = os.path.join(ROOT, "fluidsimulationcuda_amd", "csrc")
    names = sorted(f for f in os.listdir(src) if f.endswith((".hip", ".h")) or f == "Makefile")
    for f in [os.path.join(src, x) for x in names] + [os.path.join(ROOT, "include", "fluid_amd.h")]:
        h.update(os.path.basename(f).encode() + b"\0")
        h.update(open(f, "rb").read())
    return h.hexdigest()


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        sys.exit(self_launch(a))
    # stdout carries ONE line, the JSON: libraries that write to file descriptor 1 themselves (RCCL prints its host name
    # and library path there when a communicator comes up) go to stderr until that line is due
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    def emit(obj):
        sys.stdout.flush()
        os.dup2(json_fd, 1)
        print(json.dumps(obj), flush=True)
        os.dup2(2, 1)

    import numpy as np
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    a.gpus = world                                          # under a launcher the launcher's world size is the truth
    local = local % max(torch.cuda.device_count(), 1)       # rehearsal: several ranks may share a GPU
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(a.backend)
        # the first operation on a NCCL group must be one every rank takes part in (batched send/recv as
        # the first call is undefined otherwise); it also gets RCCL's lazy set-up out of the way
        hello = torch.ones(1, device="cuda" if a.backend == "nccl" else "cpu")
        dist.all_reduce(hello)
        assert int(hello.item()) == world
    from fluidsimulationcuda_amd.harness import initialize_parameters
    from fluidsimulationcuda_amd.slab import SlabSolver

    grid = a.grid or (4096 if world == 1 else 8192)
    n = grid - 2
    cells = grid * grid
    # auto: the library's own RCCL exchange when every rank brings it up, else torch.distributed (SlabSolver agrees on it)
    exchange = {"auto": "auto" if a.backend == "nccl" else "torch", "try-rccl": "auto"}.get(a.exchange, a.exchange)

    def make(n_, fuse_divergence=True):
        s = SlabSolver(n_, rank=rank, nranks=world, halo=a.halo, jacobi=a.variant,
                       storage=1 if a.dtype == "f16" else 0, exchange=exchange)
        if not fuse_divergence:
            s.set_param(9, 0)
        if a.tb_sweeps:
            s.set_param(0, a.tb_sweeps)
        if a.tb_rows:
            s.set_param(1, a.tb_rows)
        if a.fast_division >= 0:
            s.set_param(3, a.fast_division)
        if a.t16_min_cells >= 0:
            s.set_param(7, a.t16_min_cells)
        if world > 1 and overlap_choice.get(n_) is not None:
            s.set_param(13, overlap_choice[n_])
        return s

    tuning = {}
    native = [False]
    overlap_choice = {}          # grid size n -> FLUID_PARAM_XCHG_OVERLAP for this job's contexts (None: the library's default)
    overlap_trials = {}

    def tune(n_):
        """The library measures the fused kernel's strip heights during the first launches of each launch shape
        (FLUID_PARAM_TB_AUTOTUNE) and keeps the result for the process: let it finish in a throw-away context so that
        neither the warm-up nor the timed steps contain trial launches.  Not counted as steps; the same number of
        steps on every rank (a step holds collectives)."""
        if a.variant != 3 or a.tb_rows or (n_, world, overlap_choice.get(n_)) in tuning:
            return
        s = make(n_)
        s.set_param(12, 0)      # FUSE_ADD_SOURCE off here: the sourced step's own launch shape is tune_ordinary()'s business
        s.load_global(**initialize_parameters(n_, seed=a.seed))
        s.step(1, use_sources=True, iters=a.iters)
        steps = 1
        for _ in range(4):
            s.step(6, iters=a.iters)
            steps += 6
            if world == 1:
                pressure_solve_rate(s, a.iters, 8)          # the stand-alone solve's launch shapes too
            s.synchronize()
            if world == 1 and s.autotune_pending() == 0:
                break
        tuning[(n_, world, overlap_choice.get(n_))] = tuning[(n_, world)] = {"untimed_steps": steps, "shapes_still_open": s.autotune_pending()}
        s.close()

    def choose_overlap(n_):
        """Row slabs: whether the halo exchange that feeds a solve runs beside that solve's interior strips (one more, small,
        launch and two stream hops per solve; pays when the exchange takes longer than those) is a speed-only switch whose
        verdict depends on the fabric -- so it is measured: a few untimed steps each way (tuner settled for each), max over ranks,
        the faster one kept for this grid's timed steps.  Results are identical either way."""
        if world == 1 or n_ in overlap_choice:
            return
        if a.overlap >= 0:
            overlap_choice[n_] = a.overlap
            return
        ms = {}
        for ov in (1, 0):
            overlap_choice[n_] = ov
            tune(n_)
            s = make(n_)
            s.load_global(**initialize_parameters(n_, seed=a.seed))
            e, _j, _p, _t = measure(s, dist, world, 4, 2, a.iters)
            s.close()
            ms[ov] = e * 1e3 / 4
        overlap_choice[n_] = 1 if ms[1] <= ms[0] else 0
        overlap_trials[n_] = {"overlap_ms_per_step": ms[1], "in_line_ms_per_step": ms[0], "chosen": overlap_choice[n_],
                              "note": "FLUID_PARAM_XCHG_OVERLAP measured both ways over 4 untimed steps before the warm-up"}

    def run(n_, steps, warmup, fuse_divergence=True):
        choose_overlap(n_)
        tune(n_)
        fields = initialize_parameters(n_, seed=a.seed)     # same seed on every rank
        s = make(n_, fuse_divergence)
        s.load_global(**fields)
        out = measure(s, dist, world, steps, warmup, a.iters)
        solve = pressure_solve_rate(s, a.iters, max(steps // 2, 4)) if world == 1 else None
        calls = s.exchange_calls()
        native[0] = s.native_exchange
        s.close()
        return out + (solve,), fields, calls

    def tune_ordinary(n_):
        """The sourced step has a launch shape of its own (the first diffusion launch also adds the sources and stores the
        sums): one launch per step, so its strip heights are measured here, in a throw-away context stepping on
        re-injected sources, until the tuner has nothing open."""
        if a.variant != 3 or a.tb_rows or (n_, "ordinary") in tuning:
            return
        import torch
        fields = initialize_parameters(n_, seed=a.seed)
        s = make(n_)
        s.load_global(**fields)
        src = {k: torch.from_numpy(fields[k]).to(s.device) for k in ("u_prev", "v_prev", "dens_prev")}
        steps = 0
        for _ in range(48):
            with torch.cuda.stream(s.torch_stream):
                for k, t in src.items():
                    s.interior(k).copy_(t, non_blocking=True)
            s.step(1, use_sources=True, iters=a.iters)
            steps += 1
            if steps % 8 == 0:
                s.synchronize()
                if s.autotune_pending() == 0:
                    break
        tuning[(n_, "ordinary")] = {"untimed_steps": steps, "shapes_still_open": s.autotune_pending()}
        s.close()

    def run_ordinary(n_, steps, warmup):
        tune(n_)
        tune_ordinary(n_)
        fields = initialize_parameters(n_, seed=a.seed)
        s = make(n_)
        s.load_global(**fields)
        out = measure_ordinary(s, fields, steps, warmup, a.iters)
        solve = pressure_solve_rate(s, a.iters, max(steps // 2, 4))
        s.close()
        return out + (solve,)

    def rates(elapsed, jac_ms, prs_ms, t, steps, cells_, solve=None):
        t_sweep = jac_ms * 1e-3 / max(t["sweeps"], 1)
        t_instep = prs_ms * 1e-3 / max(t["pressure_sweeps"], 1)      # in-step pressure solves (with the fused divergence)
        t_psweep = solve[0] * 1e-3 / max(solve[1], 1) if solve else t_instep
        cats = {k: t[k + "_ms"] / steps for k in ("source", "diffusion", "divergence", "projection", "advection")}
        return {"value": cells_ / t_psweep / 1e6, "ms_per_step": elapsed * 1e3 / steps, "kernel_ms_per_step": cats,
                "us_per_jacobi_sweep": t_psweep * 1e6, "all_solves_value": cells_ / t_sweep / 1e6,
                "all_solves_us_per_jacobi_sweep": t_sweep * 1e6, "t_sweep": t_sweep,
                "in_step_pressure_us_per_sweep": t_instep * 1e6}

    if a.only_ordinary:
        e, j, p, t, copy_ms, solve = run_ordinary(n, a.steps, a.warmup)
        r = rates(e, j, p, t, a.steps, cells, solve)
        emit({"value_ordinary_data": r, "copy_ms_per_step": copy_ms})
        return

    if a.check and world > 1:
        # rehearsal aid: W+K steps on slabs vs the same steps in one context on this rank's GPU
        import fluidsimulationcuda_amd as F
        fields = initialize_parameters(n, seed=a.seed)
        s = make(n)
        s.load_global(**fields)
        s.step(1, use_sources=True, iters=a.iters)
        s.step(2, iters=a.iters)
        got = {k: s.gather_global(k) for k in ("u", "v", "dens")}
        s.close()
        with F.FluidSolver(n) as one:
            one.upload(**fields)
            one.step(1, use_sources=True, iters=a.iters)
            one.step(2, iters=a.iters)
            for k in got:
                same = np.array_equal(one.download(k).view(np.uint32), got[k].view(np.uint32))
                if not same:
                    sys.exit("rank %d: %s differs between %d slabs and one context" % (rank, k, world))
        if rank == 0:
            print("check ok: %d slabs bit-identical to one context at %dx%d" % (world, grid, grid), file=sys.stderr)
    def measure_grid(grid_, steps, warmup):
        """One grid on this job's ranks: the timed steps, the rates and the roofline objects."""
        n_, cells_ = grid_ - 2, grid_ * grid_
        (elapsed, jac_ms, prs_ms, t, solve), fields, calls = run(n_, steps, warmup)
        r = rates(elapsed, jac_ms, prs_ms, t, steps, cells_, solve)
        ms_step, t_sweep = r["ms_per_step"], r["t_sweep"]
        sweeps, field_launches = t["sweeps"], t["jacobi_field_launches"]
        bpc = BYTES_PER_CELL_SWEEP // (2 if a.dtype == "f16" else 1)
        achieved = bpc * cells_ / t_sweep / 1e9
        launches = max(t["jacobi_launches"], 1)
        per_launch = sweeps / launches                       # field-sweeps per launch (a batched launch sweeps 3 fields)
        kernel_name = ("k_jacobi_tb (8, 12 or 16 sweeps + set_bnd per launch, up to 3 fields per launch; %.1f launches/step)"
                       % (launches / steps)) if a.variant == 3 else "k_jacobi_%s (one sweep + fused set_bnd)" % KERNELS[a.variant]
        out = {
            "value": r["value"], "unit": "Mcells/s", "ms_per_step": ms_step, "steps": steps, "warmup": warmup,
            "kernel_ms_per_step": dict(r["kernel_ms_per_step"], note="HIP events per operator category (the reference's timers, "
                                       "FluidSequential.c:192-234); 'projection' holds the gradient subtractions, the second "
                                       "one fused with the density advection"),
            "us_per_jacobi_sweep": r["us_per_jacobi_sweep"],
            "value_definition": ("SURVEY.md 8(d)(i): W^2 / t_sweep over the 40-sweep pressure solve (alpha 1, beta 4, b 0, first guess 0), "
                                 "%d solves on the fields the timed steps left behind, HIP events around each solve; the divergence "
                                 "before it is a kernel of its own there" % (solve[1] // a.iters)) if solve else
                                "W^2 / t_sweep over the pressure solves inside the timed steps (HIP events, max over ranks)",
            "in_step_pressure_solve": {"us_per_jacobi_sweep": r["in_step_pressure_us_per_sweep"],
                                       "value": cells_ / (r["in_step_pressure_us_per_sweep"] * 1e-6) / 1e6,
                                       "note": "the 80 pressure sweeps of each timed step; their first launch also computes "
                                               "and stores the projection's divergence (no separate k_divergence pass), so this is "
                                               "sweeps + divergence"},
            "all_solves": {"value": r["all_solves_value"], "unit": "Mcells/s", "us_per_jacobi_sweep": t_sweep * 1e6,
                           "note": "the same rate over all 200 sweeps of the step (3 diffusions, whose exact division "
                                   "by 1+4a costs more than the pressure solve's multiply by 1/4, + 2 pressure solves)"},
            "step_algorithmic_GBps": BYTES_PER_CELL_STEP // (2 if a.dtype == "f16" else 1) * cells_ / (ms_step * 1e-3) / 1e9,
            "roofline": {"bound": "hbm", "kernel": kernel_name,
                         "achieved": achieved * (1.0 / world), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / world / HBM_PEAK_GBS, "traffic": None,
                         "launches": launches, "mean_launch_us": jac_ms * 1e3 / launches,
                         "algorithmic_bytes_per_launch": bpc * (cells_ // world) * per_launch,
                         "note": ("SURVEY.md 8(d)'s definition -- per GPU: algorithmic 12 B/cell/sweep x %d cells x %.1f "
                                  "field-sweeps per launch / mean launch time, HIP events on the solver's stream over %d "
                                  "timed launches" % (cells_ // world, per_launch, launches)) + (
                                     ".  Temporal blocking keeps the intermediate sweeps on chip, so this ALGORITHMIC rate "
                                     "exceeds the HBM peak and `frac` is not a fraction of anything the hardware does: "
                                     "`frac_compulsory` prices each launch at its own compulsory traffic (read x, x0, "
                                     "write x once per field = 12 B/cell), and `roofline_actual` states what the launches "
                                     "are bound by and how close they come (<= 1)" if a.variant == 3 else "")},
        }
        out["roofline"]["frac_compulsory"] = bpc * (cells_ / world) * field_launches / (jac_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
        pmc = pmc_summary(grid_) if world == 1 else None
        if pmc:
            summary, src = pmc
            stale = summary.get("_source_sha256") != library_source_hash()
            kern = "k_jacobi_%s" % KERNELS[a.variant]
            traffic = pmc_per_launch(summary, kern, "hbm_bytes_per_launch")
            valu = pmc_per_launch(summary, kern, "valu_insts_per_launch")
            mean_us = jac_ms * 1e3 / launches
            if stale:
                out["roofline"]["stale_profile"] = True
                out["roofline"]["traffic_note"] = ("profiles/%s was taken from other kernel sources than the library loaded "
                                                   "here (its _source_sha256 differs): counters not used" % src)
            elif traffic:
                out["roofline"]["traffic"] = traffic
                out["roofline"]["traffic_source"] = "profiles/" + src
            if traffic and valu and not stale:
                hbm_frac = traffic / (mean_us * 1e-6) / 1e9 / HBM_PEAK_GBS
                valu_rate = valu / SIMDS / mean_us
                valu_frac = valu_rate / VALU_PEAK_PER_SIMD_US
                out["roofline_actual"] = {
                    "bound": "valu_issue" if valu_frac >= hbm_frac else "hbm",
                    "achieved": valu_rate if valu_frac >= hbm_frac else traffic / (mean_us * 1e-6) / 1e9,
                    "peak": VALU_PEAK_PER_SIMD_US if valu_frac >= hbm_frac else HBM_PEAK_GBS,
                    "unit": "wave-instructions/us/SIMD" if valu_frac >= hbm_frac else "GB/s",
                    "frac": max(valu_frac, hbm_frac),
                    "valu_issue": {"achieved": valu_rate, "peak": VALU_PEAK_PER_SIMD_US, "frac": valu_frac,
                                   "valu_insts_per_launch": valu},
                    "hbm": {"achieved": traffic / (mean_us * 1e-6) / 1e9, "peak": HBM_PEAK_GBS, "frac": hbm_frac,
                            "bytes_per_launch": traffic},
                    "note": "mean Jacobi launch of the timed steps (HIP events) against both ceilings: vector instructions per "
                            "launch (SQ_INSTS_VALU) over the 1024 SIMDs vs what one SIMD sustains of this instruction mix "
                            "(tools/ubench/valu_peak.hip), and HBM bytes per launch (FETCH_SIZE x2 + WRITE_SIZE) vs the 8 TB/s "
                            "peak; counters from %s (the same command under rocprofv3, one --pmc pass per counter group; "
                            "its source hash matches the library loaded here)" % ("profiles/" + src)}
        elif world > 1:
            out["roofline"]["traffic_note"] = ("no PMC summary exists for slab launches (this pool hands out one GPU per call): "
                                               "frac_compulsory is the measured figure, roofline_actual is given for 1 GPU only")
        if (n_, world) in tuning:
            out["autotune"] = dict(tuning[(n_, world)], note="strip heights of the fused Jacobi kernel measured by the library in a throw-away "
                                   "context before the warm-up (FLUID_PARAM_TB_AUTOTUNE); results do not depend on them")
        if calls:
            out["exchanges_per_rank"] = {"halo": calls[0], "gather": calls[1], "max": calls[2],
                                         "per_step": {"halo": calls[0] / (steps + warmup), "max": calls[2] / (steps + warmup)}}
        if world > 1:
            out["native_exchange"] = bool(native[0])
            out["exchange_overlap"] = overlap_trials.get(n_, {"chosen": overlap_choice.get(n_), "note": "--overlap given"})
        return out, fields, (elapsed, jac_ms, prs_ms, t)

    def single_gpu(grid_, steps, warmup):
        """rank 0 of a multi-GPU job, alone: the same grid in one context on its GPU (no collectives inside)."""
        nonlocal world
        keep, world = world, 1
        try:
            tune(grid_ - 2)
            s = make(grid_ - 2)
            s.load_global(**initialize_parameters(grid_ - 2, seed=a.seed))
            e, j, p, t = measure(s, None, 1, steps, warmup, a.iters)
            s.close()
            r = rates(e, j, p, t, steps, grid_ * grid_)
            return {"ms_per_step": r["ms_per_step"], "in_step_pressure_value": r["value"], "all_solves_value": r["all_solves_value"],
                    "steps": steps, "warmup": warmup}
        except Exception as exc:                         # a courtesy figure: never fails the line
            return {"error": repr(exc)}
        finally:
            world = keep

    m, fields, _raw = measure_grid(grid, a.steps, a.warmup)
    arith = "fp32" if a.dtype == "f32" else "fp16 storage, fp32 arithmetic"
    native_exchange = native[0]
    line = {
        "metric": "Mcells/s per Jacobi iter", "value": m["value"], "unit": "Mcells/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": m["ms_per_step"],
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f32" if a.dtype == "f32" else "f16 storage, f32 arithmetic",
        "data": "synthetic (initializeParameters recipe, PCG64 seed %d; sources at step 0 only, as the reference's loop: "
                "the timed steps run on decaying fields -- see value_ordinary_data)" % a.seed,
        "config": {"workload": "%dx%d grid, full vel_step+dens_step, %d Jacobi sweeps/solve (200/step), %s"
                               % (grid, grid, a.iters, arith),
                   "grid": grid, "iters": a.iters, "jacobi_kernel": KERNELS[a.variant],
                   "parallelism": "1 GPU" if world == 1 else "row slabs x%d, halo rows by %s" % (
                       world, "RCCL, library-native exchange (csrc/fluid_exchange_rccl.hip)" if native_exchange else
                       "RCCL via torch.distributed" if a.backend == "nccl" else a.backend + " (host-staged rehearsal)")},
        "ms_per_sim_step": m["ms_per_step"],
    }
    for k, v in m.items():
        if k not in ("value", "unit", "ms_per_step", "steps", "warmup"):
            line[k] = v
    cells = grid * grid
    bpc = BYTES_PER_CELL_SWEEP // (2 if a.dtype == "f16" else 1)
    if world > 1 and grid != 4096 and not a.no_second_grid:
        # north_star: "throughput on synthetic 4096^2 and 8192^2 grids is reported at 1/2/4/8 GPUs"; BASELINE.json's metric is
        # quoted on 4096^2.  Same ranks, same code path, the other grid (strong scaling: 4096 / N rows per rank).
        g2, _f2, _r2 = measure_grid(4096, a.steps, a.warmup)
        line["grid_4096"] = dict(g2, workload="4096x4096 grid on the same %d row slabs" % world)
    if world > 1 and not a.no_scaling_base:
        if rank == 0:
            line["single_gpu"] = {"note": "rank 0 alone, one context on its GPU, the reference's loop (in-step figures), while the "
                                          "other ranks wait at the barrier below: the 1-GPU side of this line's speed-ups",
                                  str(grid): single_gpu(grid, max(a.steps // 4, 3), 2)}
            if "grid_4096" in line:
                line["single_gpu"]["4096"] = single_gpu(4096, max(a.steps // 2, 3), 2)
            for g_, sub in ((grid, line), (4096, line.get("grid_4096"))):
                base = line["single_gpu"].get(str(g_), {})
                if sub is not None and base.get("ms_per_step"):
                    sub["speedup_vs_single_gpu"] = base["ms_per_step"] / sub["ms_per_step"]
        dist.barrier()
    if world == 1 and a.variant == 3:
        # the round-1 form of the step, for comparison across rounds: every operator a launch of its own
        (e3, j3, p3, t3, _s3), _, _ = run(n, a.steps, a.warmup, fuse_divergence=False)
        r3 = rates(e3, j3, p3, t3, a.steps, cells)
        line["divergence_as_its_own_kernel"] = {
            "ms_per_step": r3["ms_per_step"], "in_step_pressure_us_per_sweep": r3["in_step_pressure_us_per_sweep"],
            "in_step_pressure_value": cells / (r3["in_step_pressure_us_per_sweep"] * 1e-6) / 1e6,
            "all_solves_value": r3["all_solves_value"], "kernel_ms_per_step": r3["kernel_ms_per_step"],
            "note": "FLUID_PARAM_FUSE_DIVERGENCE = 0: the same K steps with k_divergence as a separate pass, so that the "
                    "pressure solves inside the steps are sweeps alone (how round 1 measured `value`)"}
    if world == 1 and not a.no_ordinary:
        e2, j2, p2, t2, copy_ms, solve2 = run_ordinary(n, a.steps, a.warmup)
        r2 = rates(e2, j2, p2, t2, a.steps, cells, solve2)
        line["value_ordinary_data"] = {
            "value": r2["value"], "unit": "Mcells/s", "us_per_jacobi_sweep": r2["us_per_jacobi_sweep"],
            "all_solves_value": r2["all_solves_value"], "all_solves_us_per_jacobi_sweep": r2["all_solves_us_per_jacobi_sweep"],
            "ms_per_step": r2["ms_per_step"], "source_copies_ms_per_step": copy_ms, "kernel_ms_per_step": r2["kernel_ms_per_step"],
            "frac_compulsory": bpc * cells * t2["jacobi_field_launches"] / (j2 * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "note": "same grid and kernels; the three source fields are re-injected before every step (device-to-device "
                    "copies, included in ms_per_step, %.3f ms of it) and consumed by add_source, so all fields keep "
                    "ordinary magnitudes instead of decaying towards zero" % copy_ms}
    if world == 1 and not a.no_scaling_base and grid != 8192:
        steps2 = max(a.steps // 4, 3)
        (e2, j2, p2, t2, solve2), _, _ = run(8190, steps2, 2)
        r2 = rates(e2, j2, p2, t2, steps2, 8192 * 8192, solve2)
        line["scaling_base"] = {"workload": "8192x8192 on 1 GPU", "value": r2["value"], "unit": "Mcells/s",
                                "ms_per_step": r2["ms_per_step"], "all_solves_value": r2["all_solves_value"],
                                "in_step_pressure_value": 8192 * 8192 / (r2["in_step_pressure_us_per_sweep"] * 1e-6) / 1e6,
                                "roofline_frac": BYTES_PER_CELL_SWEEP * 8192 * 8192 / r2["t_sweep"] / 1e9 / HBM_PEAK_GBS,
                                "frac_compulsory": BYTES_PER_CELL_SWEEP * 8192 * 8192 * t2["jacobi_field_launches"]
                                / (j2 * 1e-3) / 1e9 / HBM_PEAK_GBS}
        if not a.no_ordinary:
            eo, jo, po, to, copy_o, solve_o = run_ordinary(8190, steps2, 2)
            ro = rates(eo, jo, po, to, steps2, 8192 * 8192, solve_o)
            line["scaling_base"]["ordinary_data"] = {"ms_per_step": ro["ms_per_step"], "value": ro["value"],
                                                     "all_solves_value": ro["all_solves_value"],
                                                     "source_copies_ms_per_step": copy_o}
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(n, fields, a.iters)
    if rank == 0:
        emit(line)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
