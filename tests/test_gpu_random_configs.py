"""GPU: seeded random configurations.  The hand-picked cases elsewhere fix most knobs at a time; here they vary
together -- grid size, sweep count, launch depth, lanes, strip height, division form, the fused divergence, the tuner --
so that interactions between the round-2 and round-3 paths (planned depths, 12-sweep launches, divergence inside the pressure
solve, second stream on slabs; the scaled-residual division, add_source inside the first diffusion launch, launches split around
an exchange in flight) are exercised.  One GPU: two steps against the oracle, all six fields.  Slabs (in-process
fabric): two steps against one context, plus the requirement that every rank issues the same exchange sequence."""
import os

import numpy as np
import pytest

from conftest import assert_bit_equal, rnd

pytestmark = pytest.mark.gpu
# more seeds for a one-off soak: FLUID_FUZZ_SINGLE=2000 FLUID_FUZZ_SLABS=500 python -m pytest tests/test_gpu_random_configs.py
N_SINGLE = int(os.environ.get("FLUID_FUZZ_SINGLE", "160"))
N_SLABS = int(os.environ.get("FLUID_FUZZ_SLABS", "64"))


def random_params(rng, capi):
    p = {capi.PARAM_TB_MIN_CELLS: int(rng.choice([0, 0, 0, 1 << 30])),
         capi.PARAM_TB_T16_MIN_CELLS: int(rng.choice([0, 0, -1])),
         capi.PARAM_TB_MAX_SWEEPS: int(rng.choice([16, 12, 8, 4, 2])),
         capi.PARAM_TB_LANE_COLUMNS: int(rng.choice([2, 2, 4])),
         capi.PARAM_TB_ROWS: int(rng.choice([0, 0, 1, 2, 5, 17, 64, 1000])),
         capi.PARAM_TB_FAST_DIVISION: int(rng.choice([0, 1, 2, 2, 3])),
         capi.PARAM_FUSE_DIVERGENCE: int(rng.choice([0, 1, 1])),
         capi.PARAM_FUSE_ADD_SOURCE: int(rng.choice([0, 1, 1])),
         capi.PARAM_TB_AUTOTUNE: int(rng.choice([0, 1])),
         capi.PARAM_TB_EDGE_ROWS_PCT: int(rng.choice([40, 0, 100]))}
    return p


@pytest.mark.parametrize("seed", range(N_SINGLE))
def test_random_single_gpu_configuration_matches_oracle(oracle, seed):
    import fluidsimulationcuda_amd as F
    from fluidsimulationcuda_amd import capi
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.choice([1, 2, 3, 5, 8, 13, 31, 47, 48, 49, 63, 64, 65, 95, 96, 97, 127, 128, 129, 200, 255, 256, 257, 333, 511]))
    iters = int(rng.choice([0, 2, 4, 6, 8, 10, 12, 14, 16, 18, 20, 22, 24, 28, 32, 36, 40, 44]))
    variant = int(rng.choice([3, 3, 3, 3, 0, 1, 2]))
    params = random_params(rng, capi)
    coarse = bool(rng.integers(0, 2))
    vals = np.array([-1, -0.5, -0.25, 0.0, -0.0, 0.25, 0.5, 1], np.float32)
    fields = [rng.choice(vals, size=(n + 2, n + 2)).astype(np.float32) if coarse else rnd(rng, n) for _ in range(6)]
    u, v, dens, u0, v0, d0 = fields
    dt, diff, visc = float(rng.choice([0.016, 0.1])), float(rng.choice([0.1, 0.0, 1e-4])), float(rng.choice([0.0025, 0.0, 0.3]))
    what = "seed %d: n=%d iters=%d variant=%d %r" % (seed, n, iters, variant, params)
    with F.FluidSolver(n, jacobi=variant, params=params) as s:
        s.upload(u=u, v=v, dens=dens, u_prev=u0, v_prev=v0, dens_prev=d0)
        s.step(1, use_sources=True, dt=dt, diff=diff, visc=visc, iters=iters)
        s.step(1, dt=dt, diff=diff, visc=visc, iters=iters)
        oracle.step_src(u, v, dens, u0, v0, d0, dt=dt, diff=diff, visc=visc, iters=iters)
        oracle.step(u, v, dens, u0, v0, d0, dt=dt, diff=diff, visc=visc, iters=iters)
        for name, want in (("u", u), ("v", v), ("dens", dens), ("u_prev", u0), ("v_prev", v0), ("dens_prev", d0)):
            assert_bit_equal(s.download(name), want, name + " -- " + what)


@pytest.mark.parametrize("seed", range(N_SLABS))
def test_random_slab_configuration_matches_one_context(seed):
    from test_gpu_slab import run_ranks, single
    from fluidsimulationcuda_amd import capi
    from fluidsimulationcuda_amd.harness import initialize_parameters
    rng = np.random.default_rng(2000 + seed)
    nranks = int(rng.choice([2, 2, 3, 4, 5]))
    n = int(rng.choice([61, 100, 126, 200, 254, 257, 400, 510]))
    while n // nranks < 10:
        nranks -= 1
    storage = int(rng.choice([0, 0, 1]))
    halo = int(rng.choice([0, 1, 3, 8, 9, 16, 40, 41, 42, 60]))
    iters = int(rng.choice([2, 8, 12, 20, 28, 40, 40]))
    jacobi = int(rng.choice([3, 3, 3, 0]))
    params = random_params(rng, capi)
    params[capi.PARAM_SLAB_OVERLAP] = int(rng.choice([0, 1, 1]))
    params[capi.PARAM_XCHG_OVERLAP] = int(rng.choice([0, 1, 1]))
    params.pop(capi.PARAM_TB_MIN_CELLS)          # (the fake ranks force it to 0, as the single-context reference does)
    if storage == 1:
        # fp16 results depend on the launch schedule: keep the knobs that change it at their defaults on both sides
        jacobi = 3
        for k in (capi.PARAM_TB_MAX_SWEEPS, capi.PARAM_TB_T16_MIN_CELLS):
            params.pop(k)
    fields = initialize_parameters(n, seed=seed)
    what = "seed %d: n=%d ranks=%d halo=%d iters=%d storage=%d jacobi=%d %r" % (seed, n, nranks, halo, iters, storage, jacobi, params)

    def body(s):
        s.step(1, use_sources=True, iters=iters)
        s.step(1, iters=iters)

    want = single(n, fields, body, storage=storage)
    got, fab = run_ranks(n, nranks, halo, fields, body, jacobi=jacobi, storage=storage, params=params)
    for r in range(1, nranks):
        assert fab.log[r] == fab.log[0], "rank %d issued a different exchange sequence -- %s" % (r, what)
    for k in ("u", "v", "dens", "u_prev", "v_prev", "dens_prev"):
        assert_bit_equal(got[k], want[k], k + " -- " + what)
