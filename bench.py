#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on MI355X: Mcells/s per Jacobi iteration
and ms per simulation step of the Stable-Fluids vel_step + dens_step.

  python bench.py [--gpus N --steps K --warmup W]      (N>1: under torch.distributed.run)

`value` follows SURVEY.md 8(d): W^2 / t_sweep with t_sweep the mean time of one sweep (set_bnd included) of the
40-sweep pressure solves (alpha 1, beta 4, b 0) inside the timed steps, HIP events on the solver's stream;
`all_solves` is the same rate over all 200 sweeps of a step, and the `roofline` object prices ALL Jacobi launches.

A "step" is one loop body of the reference's main (FluidSequential.c:298-306):
sources zeroed, vel_step, dens_step, 40 Jacobi sweeps per solve = 200 sweeps,
3 advects, 2 projections.  Fields are resident in HBM when the timed region
starts.  One JSON line on rank 0.

Workload: N=1 -> 4096^2 (the grid BASELINE.json's metric is quoted on, config
2); N>1 -> 8192^2 split into row slabs (config 3), strong scaling.  The N=1
line also carries the 1-GPU 8192^2 measurement ("scaling_base") so the 8192^2
speed-up can be formed from the driver's own runs.
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
    ap.add_argument("--halo", type=int, default=0, help="multi-GPU ghost-zone depth (0 = library default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-scaling-base", action="store_true")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--dtype", default="f32", choices=["f32", "f16"],
                    help="field storage: f32 (reference arithmetic, the measured configuration) or f16 "
                         "(BASELINE config 4: fp16 fields, fp32 arithmetic)")
    ap.add_argument("--backend", default="nccl", help="nccl (= RCCL; the measured path) or gloo (host-staged rehearsal "
                                                      "of the multi-process path when ranks outnumber GPUs)")
    ap.add_argument("--check", action="store_true", help="multi-GPU: also verify bit equality with a 1-context run (small grids)")
    return ap.parse_args()


def measure(solver, dist, world, steps, warmup, iters, cells):
    """W untimed steps, then exactly K timed steps between barrier+synchronize
    pairs; returns (max-over-ranks seconds, max-over-ranks Jacobi ms, sweeps, Jacobi launches,
    launches counted once per field swept, max-over-ranks ms in the pressure solves, their sweeps)."""
    import torch
    solver.step(1, use_sources=True, iters=iters)          # z == 0 consumes the synthetic sources
    for _ in range(max(warmup - 1, 0)):
        solver.step(1, iters=iters)
    solver.timing_enable(True)
    solver.timing_read(reset=True)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    solver.step(steps, iters=iters)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    t = solver.timing_read(reset=True)
    solver.timing_enable(False)
    jac_ms, sweeps, prs_ms = t["jacobi_ms"], t["sweeps"], t["pressure_ms"]
    if world > 1:
        buf = torch.tensor([elapsed, jac_ms, prs_ms], dtype=torch.float64,
                           device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(buf, op=dist.ReduceOp.MAX)
        elapsed, jac_ms, prs_ms = float(buf[0]), float(buf[1]), float(buf[2])
    return elapsed, jac_ms, sweeps, t["jacobi_launches"], t["jacobi_field_launches"], prs_ms, t["pressure_sweeps"]


def pmc_traffic(kernel, grid):
    """Mean HBM bytes per launch of the dominant kernel (all its instantiations, weighted by how often
    each ran) from the committed rocprofv3 PMC summary (profiles/r*_<grid>_pmc.json, written by
    tools/summarize_profiles.py from separate --pmc FETCH_SIZE / --pmc WRITE_SIZE passes of this same
    command, gfx950 x2 read correction applied).  The newest summary wins; None when none matches."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_%d_pmc.json" % grid))):
        try:
            d = json.load(open(f))
        except (OSError, ValueError):
            continue
        tot = cnt = 0.0
        for name, v in d.items():
            if name.split("<")[0] != kernel or not v.get("hbm_bytes_per_launch"):
                continue
            tot += v["hbm_bytes_per_launch"] * v["launches_sampled"]
            cnt += v["launches_sampled"]
        if cnt:
            best = (tot / cnt, os.path.basename(f))
    return best


def cpu_baseline(n, fields, iters):
    """The oracle leg (checker only): one full step and one 40-sweep pressure
    solve on ONE host core at the bench workload's N -- the reference itself
    (oracle/_ref, kind "reference") when its build for this N travelled here,
    else the restatement (kind "port")."""
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
                     % (iters, w, w)}
    # secondary line: the restatement's sweep split into row bands over all host cores
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = min(cores, 16)          # one GPU's share of the host; more Python threads only add hand-off cost
    if cores > 1:
        p2 = np.zeros((w, w), np.float32)
        t0 = time.perf_counter()
        Oracle().diffuse_threaded(0, p2, div, 1.0, 4.0, 8, cores)
        t_mt = (time.perf_counter() - t0) / 8
        out["all_cores"] = {"value": w * w / t_mt / 1e6, "unit": "Mcells/s per Jacobi iter", "cores": cores,
                            "kind": "port", "sample": "8 sweeps of the pressure solve, row bands over %d threads" % cores}
    return out


def main():
    a = parse()
    import numpy as np
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d" % (a.gpus, a.gpus))
        a.gpus = world
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

    def run(n_, steps, warmup):
        fields = initialize_parameters(n_, seed=a.seed)     # same seed on every rank
        s = SlabSolver(n_, rank=rank, nranks=world, halo=a.halo, jacobi=a.variant,
                       storage=1 if a.dtype == "f16" else 0)
        if a.tb_sweeps:
            s.set_param(0, a.tb_sweeps)
        if a.tb_rows:
            s.set_param(1, a.tb_rows)
        s.load_global(**fields)
        out = measure(s, dist, world, steps, warmup, a.iters, (n_ + 2) ** 2)
        calls = dict(s.exchange.calls) if s.exchange else None
        s.close()
        return out, fields, calls

    if a.check and world > 1:
        # rehearsal aid: W+K steps on slabs vs the same steps in one context on this rank's GPU
        import fluidsimulationcuda_amd as F
        fields = initialize_parameters(n, seed=a.seed)
        s = SlabSolver(n, rank=rank, nranks=world, halo=a.halo, jacobi=a.variant)
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
    (elapsed, jac_ms, sweeps, launches, field_launches, prs_ms, prs_sweeps), fields, calls = run(n, a.steps, a.warmup)
    ms_step = elapsed * 1e3 / a.steps
    t_sweep = jac_ms * 1e-3 / max(sweeps, 1)             # all 200 sweeps of the step
    t_psweep = prs_ms * 1e-3 / max(prs_sweeps, 1)        # the pressure solves: SURVEY.md 8(d)'s definition of the metric
    mcells = cells / t_psweep / 1e6
    bpc = BYTES_PER_CELL_SWEEP // (2 if a.dtype == "f16" else 1)
    achieved = bpc * cells / t_sweep / 1e9
    launches = max(launches, 1)
    per_launch = sweeps / launches                       # field-sweeps per launch (a batched launch sweeps 3 fields)
    kernel_name = ("k_jacobi_tb (8 or 16 sweeps + set_bnd per launch, up to 3 fields per launch; %.1f launches/step)"
                   % (launches / a.steps)) if a.variant == 3 else "k_jacobi_%s (one sweep + fused set_bnd)" % KERNELS[a.variant]
    line = {
        "metric": "Mcells/s per Jacobi iter", "value": mcells, "unit": "Mcells/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms_step,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f32" if a.dtype == "f32" else "f16 storage, f32 arithmetic",
        "data": "synthetic (initializeParameters recipe, PCG64 seed %d)" % a.seed,
        "config": {"workload": "%dx%d grid, full vel_step+dens_step, %d Jacobi sweeps/solve (200/step), fp32"
                               % (grid, grid, a.iters),
                   "grid": grid, "iters": a.iters, "jacobi_kernel": KERNELS[a.variant],
                   "parallelism": "1 GPU" if world == 1 else "row slabs x%d, %s halo rows" % (
                       world, "RCCL" if a.backend == "nccl" else a.backend + " (host-staged rehearsal)")},
        "ms_per_sim_step": ms_step,
        "us_per_jacobi_sweep": t_psweep * 1e6,
        "all_solves": {"value": cells / t_sweep / 1e6, "unit": "Mcells/s", "us_per_jacobi_sweep": t_sweep * 1e6,
                       "note": "the same rate over all 200 sweeps of the step (3 diffusions, whose exact division "
                               "by 1+4a costs more than the pressure solve's multiply by 1/4, + 2 pressure solves)"},
        "step_algorithmic_GBps": BYTES_PER_CELL_STEP // (2 if a.dtype == "f16" else 1) * cells / (ms_step * 1e-3) / 1e9,
        "roofline": {"bound": "hbm", "kernel": kernel_name,
                     "achieved": achieved * (1.0 / world), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / world / HBM_PEAK_GBS, "traffic": None,
                     "launches": launches, "mean_launch_us": jac_ms * 1e3 / launches,
                     "algorithmic_bytes_per_launch": bpc * (cells // world) * per_launch,
                     "note": ("per GPU: algorithmic 12 B/cell/sweep x %d cells x %.1f field-sweeps per launch / mean "
                              "launch time, HIP events on the solver's stream over %d timed launches"
                              % (cells // world, per_launch, launches)) + (
                                 "; temporal blocking keeps the intermediate sweeps on chip, so the algorithmic "
                                 "rate may exceed the HBM peak -- frac_compulsory prices each launch at its own "
                                 "compulsory traffic (read x, x0, write x once per field = 12 B/cell)" if a.variant == 3 else "")},
    }
    line["roofline"]["frac_compulsory"] = bpc * (cells / world) * field_launches / (jac_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
    tr = pmc_traffic("k_jacobi_%s" % KERNELS[a.variant], grid) if world == 1 else None
    if tr:
        line["roofline"]["traffic"] = tr[0]
        line["roofline"]["traffic_source"] = "profiles/" + tr[1]
    if calls:
        line["exchanges_per_rank"] = {"halo": calls[0], "gather": calls[1], "max": calls[2]}
    if world == 1 and not a.no_scaling_base and grid != 8192:
        (e2, j2, s2, _l2, _f2, p2, ps2), _, _ = run(8190, max(a.steps // 4, 3), 2)
        ts2 = p2 * 1e-3 / max(ps2, 1)
        line["scaling_base"] = {"workload": "8192x8192 on 1 GPU", "value": 8192 * 8192 / ts2 / 1e6, "unit": "Mcells/s",
                                "ms_per_step": e2 * 1e3 / max(a.steps // 4, 3),
                                "all_solves_value": 8192 * 8192 / (j2 * 1e-3 / max(s2, 1)) / 1e6,
                                "roofline_frac": BYTES_PER_CELL_SWEEP * 8192 * 8192 / (j2 * 1e-3 / max(s2, 1)) / 1e9 / HBM_PEAK_GBS}
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(n, fields, a.iters)
    if rank == 0:
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
