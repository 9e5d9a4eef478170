"""GPU: the row-slab (multi-GPU) orchestration on ONE device.  P contexts, one
thread each, stand in for P ranks; the exchange callback is an in-process fake
(device-to-device row copies between the contexts' arenas behind a thread
barrier).  The bar is the multi-GPU contract of SURVEY.md 8(e): results
bit-identical to the single-context run -- which test_gpu_step pins to the
reference."""
import threading

import numpy as np
import pytest

from conftest import assert_bit_equal, rnd

pytestmark = pytest.mark.gpu
DT, VISC, DIFF = 0.016, 0.0025, 0.1


@pytest.fixture(scope="module")
def F():
    import fluidsimulationcuda_amd as F
    return F


class FakeFabric:
    """Shared state of the P fake ranks: a barrier and everyone's field tensors."""

    def __init__(self, nranks):
        self.nranks = nranks
        self.barrier = threading.Barrier(nranks)
        self.solvers = [None] * nranks
        self.scalars = [0.0] * nranks
        self.log = [[] for _ in range(nranks)]
        self.maxima = [[] for _ in range(nranks)]      # the local bound each rank brought to every MAX exchange

    def make_callback(self, rank):
        import torch
        from fluidsimulationcuda_amd import capi
        from fluidsimulationcuda_amd.slab import slab_rows

        def cb(kind, ids, depth, scalar):
            me = self.solvers[rank]
            self.log[rank].append((kind, tuple(ids), depth))
            if kind == capi.XCHG_MAX_BEGIN:          # this fabric reduces on the host, at END
                return None
            if kind in (capi.XCHG_MAX, capi.XCHG_MAX_END):
                self.scalars[rank] = scalar
                self.maxima[rank].append(scalar)
                self.barrier.wait()
                out = max(self.scalars)
                self.barrier.wait()
                return out
            me.synchronize()                         # this rank's producers are done
            self.barrier.wait()                      # ... and so are everyone else's
            lo, hi = me.owned_rows
            for fid in ids:
                mine = me.field_tensor(fid)
                if kind == capi.XCHG_HALO:
                    if rank > 0:                     # pull the rows above my slab from the rank above
                        mine[lo - depth:lo].copy_(self.solvers[rank - 1].field_tensor(fid)[lo - depth:lo])
                    if rank < self.nranks - 1:
                        mine[hi:hi + depth].copy_(self.solvers[rank + 1].field_tensor(fid)[hi:hi + depth])
                else:                                # gather: pull every other slab (+ wall rows)
                    for r in range(self.nranks):
                        if r == rank:
                            continue
                        a, b = slab_rows(me.n, r, self.nranks)
                        a -= 1 if r == 0 else 0
                        b += 1 if r == self.nranks - 1 else 0
                        mine[a:b].copy_(self.solvers[r].field_tensor(fid)[a:b])
            torch.cuda.synchronize()
            self.barrier.wait()                      # nobody overwrites rows still being read
            return None

        return cb


def run_ranks(n, nranks, halo, fields, body, jacobi=0, storage=0, params=None):
    """Run body(solver) on every fake rank; returns the gathered fields."""
    from fluidsimulationcuda_amd.slab import SlabSolver
    fab = FakeFabric(nranks)
    solvers = []
    for r in range(nranks):
        s = SlabSolver.__new__(SlabSolver)
        _init_fake(s, n, r, nranks, halo, jacobi, storage, params)
        s.set_exchange(fab.make_callback(r))
        fab.solvers[r] = s
        s.load_global(**fields)
        solvers.append(s)
    errs = []

    def work(r):
        try:
            body(solvers[r])
            solvers[r].synchronize()
        except Exception as e:      # noqa: BLE001
            errs.append((r, e))
            fab.barrier.abort()

    ts = [threading.Thread(target=work, args=(r,)) for r in range(nranks)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs, errs
    out = {}
    for name in ("u", "v", "dens", "u_prev", "v_prev", "dens_prev"):
        full = np.zeros((n + 2, n + 2), np.float32)
        for s in solvers:
            lo, hi = s.owned_rows
            lo -= 1 if s.rank == 0 else 0
            hi += 1 if s.rank == nranks - 1 else 0
            s.download_rows(name, full, lo, hi)
        out[name] = full
    for s in solvers:
        s.close()
    return out, fab


def _init_fake(s, n, rank, nranks, halo, jacobi, storage=0, params=None):
    """SlabSolver.__init__ minus the torch.distributed exchange."""
    import ctypes as C
    import torch
    from fluidsimulationcuda_amd import capi
    from fluidsimulationcuda_amd.solver import FluidSolver
    L = capi.lib()
    nbytes = L.fluid_arena_bytes_ex(n, storage)
    pitch, xoff, ff = C.c_int(), C.c_int(), C.c_size_t()
    capi.check(L.fluid_layout(n, C.byref(pitch), C.byref(xoff), C.byref(ff)))
    s.pitch, s.xoff, s.device = pitch.value, xoff.value, torch.device("cuda", 0)
    s.arena = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
    s.torch_stream = torch.cuda.Stream()
    torch.cuda.synchronize()
    FluidSolver.__init__(s, n, rank=rank, nranks=nranks, halo=halo, jacobi=jacobi,
                         stream=s.torch_stream.cuda_stream,
                         arena_ptr=s.arena.data_ptr(), arena_bytes=nbytes, storage=storage,
                         params={capi.PARAM_TB_MIN_CELLS: 0, **(params or {})})   # fuse sweeps even on small slabs
    esz, dt = (2, torch.float16) if storage else (4, torch.float32)
    s._fb = ff.value * esz
    s._views = [s.arena[k * s._fb:(k + 1) * s._fb].view(dt).view(n + 2, s.pitch) for k in range(capi.NFIELDS)]
    s.exchange = None


def single(n, fields, body, storage=0):
    import fluidsimulationcuda_amd as F
    from fluidsimulationcuda_amd import capi
    with F.FluidSolver(n, storage=storage, params={capi.PARAM_TB_MIN_CELLS: 0}) as s:
        s.upload(**fields)
        body(s)
        return {k: s.download(k) for k in ("u", "v", "dens", "u_prev", "v_prev", "dens_prev")}


def synthetic(n, seed=1):
    from fluidsimulationcuda_amd.harness import initialize_parameters
    return initialize_parameters(n, seed=seed)


@pytest.mark.parametrize("jacobi", [0, 3])
@pytest.mark.parametrize("n,nranks,halo", [(126, 2, 8), (126, 4, 3), (61, 3, 1), (257, 8, 5), (254, 2, 40),
                                           (126, 2, 7), (510, 2, 24), (1022, 2, 42), (1022, 4, 0)])
def test_steps_bit_identical_to_one_gpu(n, nranks, halo, jacobi):
    """Three full steps: slabs == single context, every field, every bit
    (includes uneven splits, halo depths that do / do not divide 40)."""
    fields = synthetic(n)

    def body(s):
        s.step(1, use_sources=True)
        s.step(2)

    want = single(n, fields, body)
    got, fab = run_ranks(n, nranks, halo, fields, body, jacobi=jacobi)
    for k in ("u", "v", "dens"):
        assert_bit_equal(got[k], want[k], "%s, %d slabs halo %d kernel %d" % (k, nranks, halo, jacobi))
    from fluidsimulationcuda_amd import capi
    for r in range(1, nranks):                            # collectives pair up only if every rank asks for the same ones
        assert fab.log[r] == fab.log[0], "rank %d issued a different exchange sequence" % r
    kinds = [e[0] for e in fab.log[0]]
    assert kinds.count(capi.XCHG_MAX_END) == 2 * 3        # two advect bounds per step
    # deep ghost zones: far fewer halo exchanges than the 200 sweeps of a step
    depth = max(1, min(halo, n // nranks - 1))
    per_solve = 1 + (40 - 1) // depth
    assert kinds.count(capi.XCHG_HALO) <= 3 * (5 * per_solve + 2 * 2 + 2 + 1)
    if depth >= 42:
        # tall slabs, deep ghost zones: diffusion x3 share one exchange, each projection
        # needs one, each advect at most one (SURVEY.md 8(e) asks for one PER SWEEP: 200)
        assert kinds.count(capi.XCHG_HALO) <= 3 * 5


@pytest.mark.parametrize("n,nranks,halo,iters", [(1022, 2, 42, 40), (1022, 4, 0, 40), (510, 3, 8, 40), (1022, 8, 20, 20)])
def test_first_launches_split_around_the_exchange(n, nranks, halo, iters):
    """FLUID_PARAM_XCHG_OVERLAP (exchange / compute overlap): the halo exchange that feeds a solve is issued without the
    compute stream waiting for it; the solve's next launch runs its interior strips -- output rows [own0 + T, own1 - T), whose
    inputs are the slab's own rows -- first, and the strips at the slab's inner edges behind the exchange's event.  Every cell
    goes through the same arithmetic whichever part it falls into: three steps must equal the single context bit for bit,
    with the split on (and actually happening: the counter) and off."""
    from fluidsimulationcuda_amd import capi
    fields = synthetic(n, seed=5)
    splits = {}

    def body(s):
        s.step(1, use_sources=True, iters=iters)
        s.step(2, iters=iters)
        splits[s.rank] = s.split_launches()

    want = single(n, fields, body)
    assert splits[0] == 0                                  # one context: nothing to split around
    for overlap in (1, 0):
        splits.clear()
        got, fab = run_ranks(n, nranks, halo, fields, body, jacobi=3, params={capi.PARAM_XCHG_OVERLAP: overlap})
        for k in ("u", "v", "dens", "u_prev", "v_prev", "dens_prev"):
            assert_bit_equal(got[k], want[k], "%s, %d slabs halo %d, overlap %d" % (k, nranks, halo, overlap))
        for r in range(1, nranks):
            assert fab.log[r] == fab.log[0], "rank %d issued a different exchange sequence" % r
        if overlap:
            # per step at least: the diffusion's first launch(es) and the first launch of each projection's solve
            assert all(v >= 3 * 3 for v in splits.values()), splits
        else:
            assert all(v == 0 for v in splits.values()), splits


@pytest.mark.parametrize("min_cells", [31 * 126 + 1, 32 * 126, 32 * 126 * 3 - 5])
def test_uneven_slabs_take_the_same_decisions(min_cells):
    """126 rows over 4 ranks = slabs of 32, 32, 31, 31 rows.  Size-dependent choices (fuse sweeps or not,
    how many per launch) with their thresholds placed between the two slab sizes: every rank must
    still issue the same exchange sequence, and the result must not change."""
    from fluidsimulationcuda_amd import capi
    n, nranks = 126, 4
    fields = synthetic(n, seed=5)

    def body(s):
        s.step(1, use_sources=True)
        s.step(1)

    want = single(n, fields, body)
    got, fab = run_ranks(n, nranks, 6, fields, body, jacobi=3, params={capi.PARAM_TB_MIN_CELLS: min_cells})
    for r in range(1, nranks):
        assert fab.log[r] == fab.log[0], "rank %d issued a different exchange sequence" % r
    for k in ("u", "v", "dens"):
        assert_bit_equal(got[k], want[k], "%s with TB_MIN_CELLS=%d" % (k, min_cells))


def test_advect_large_velocity_falls_back_to_gather():
    """Back-trace longer than a neighbour's slab: the solver must gather whole
    fields (FLUID_XCHG_GATHER) and still match one GPU bit for bit."""
    n, nranks = 126, 4
    rng = np.random.default_rng(11)
    fields = dict(u=rnd(rng, n, -30, 30), v=rnd(rng, n, -30, 30), dens_prev=rnd(rng, n),
                  dens=np.zeros((n + 2, n + 2), np.float32))

    def body(s):
        s.advect(0, "dens", "dens_prev", "u", "v", DT)

    want = single(n, fields, body)
    got, fab = run_ranks(n, nranks, 4, fields, body)
    assert_bit_equal(got["dens"], want["dens"], "advect with gather fallback")
    from fluidsimulationcuda_amd import capi
    assert any(e[0] == capi.XCHG_GATHER for e in fab.log[0])


def test_advect_small_velocity_uses_bounded_halo():
    n, nranks = 254, 4
    rng = np.random.default_rng(12)
    fields = dict(u=rnd(rng, n, -0.5, 0.5), v=rnd(rng, n, -0.5, 0.5), dens_prev=rnd(rng, n),
                  dens=np.zeros((n + 2, n + 2), np.float32))

    def body(s):
        s.advect(0, "dens", "dens_prev", "u", "v", DT)

    want = single(n, fields, body)
    got, fab = run_ranks(n, nranks, 4, fields, body)
    assert_bit_equal(got["dens"], want["dens"], "advect with bounded halo")
    from fluidsimulationcuda_amd import capi
    halos = [e for e in fab.log[1] if e[0] == capi.XCHG_HALO]
    # dt0*vmax = 0.016*254*0.5 ~ 2.03 -> ceil + 2 = 5 rows
    assert len(halos) == 1 and halos[0][2] == 5 and not any(e[0] == capi.XCHG_GATHER for e in fab.log[1])


@pytest.mark.parametrize("variant", [0, 1, 2, 3])
def test_operators_on_slabs(variant):
    n, nranks = 126, 3
    rng = np.random.default_rng(13)
    fields = dict(u=rnd(rng, n), v=rnd(rng, n), dens=rnd(rng, n), u_prev=rnd(rng, n), v_prev=rnd(rng, n),
                  dens_prev=rnd(rng, n))

    def body(s):
        s.add_source("u", "u_prev", DT)
        s.diffuse(1, "u_prev", "u", 0.3, 2.2, 6)
        s.computeDivergenceAndPressure("u", "v", "dens", "dens_prev")
        s.diffuse(0, "dens", "dens_prev", 1.0, 4.0, 10)
        s.lastProject("u", "v", "dens")

    want = single(n, fields, body)
    got, _ = run_ranks(n, nranks, 4, fields, body, jacobi=variant)
    for k in want:
        assert_bit_equal(got[k], want[k], k)


@pytest.mark.parametrize("halo", [4, 8, 30])
def test_two_term_division_on_slabs_with_tiny_values_in_the_neighbours_rows(oracle, halo):
    """Division mode 3 (FAST_DIVISION = 1) takes its two-term path only where the tiles of |x0| minima allow it, and the
    minima are taken once per solve over the rows valid at that moment.  A right-hand side that is tiny (2^-120 .. 2^-80:
    where the two-term quotient can be an ulp off) ONLY in rows just beyond a slab's edge -- rows that become valid later,
    through an exchange inside the solve -- must not be vouched for by the rows of the same tile that were looked at: a
    tile only partly inside the valid rows counts as unknown.  Slabs == one context == the oracle, bit for bit."""
    from fluidsimulationcuda_amd import capi
    import fluidsimulationcuda_amd as F
    n, nranks = 254, 2
    rng = np.random.default_rng(halo)
    alpha, beta = F.coefficients(n, DT, VISC)
    x, x0 = rnd(rng, n), rnd(rng, n, 0.5, 1.5)
    edge = n // nranks + 1                                   # first row of the second slab
    for lo, hi in ((edge + 1, edge + 12), (edge - 12, edge - 1)):
        x0[lo:hi] *= (np.float32(2.0) ** rng.integers(-120, -80, (hi - lo, n + 2))).astype(np.float32)
    fields = {k: np.zeros((n + 2, n + 2), np.float32) for k in ("u", "v", "dens", "u_prev", "v_prev", "dens_prev")}
    fields["u"], fields["v"] = x, x0

    def body(s):
        assert s.division_mode(alpha, beta) == 3
        s.diffuse(1, "u", "v", alpha, beta, 40)

    params = {capi.PARAM_TB_FAST_DIVISION: 1}
    got, _ = run_ranks(n, nranks, halo, fields, body, jacobi=3, params=params)
    want = x.copy()
    oracle.diffuse(1, want, x0, alpha, beta, 40)
    assert_bit_equal(got["u"], want, "two-term division on slabs, halo %d" % halo)


def test_halo_depth_is_checked_against_the_shortest_slab_on_every_rank():
    """N = 257 over 3 slabs: 86, 86 and 85 rows.  A halo of 86 rows fits the tall slabs and not the short one; checked
    against each rank's own height, the tall ranks would enter the exchange and wait for a peer that has already
    returned an error.  fluid_exchange_now validates against the shortest slab, which every rank knows: all three refuse,
    none calls its exchange; 85 rows pass on all three."""
    import ctypes as C
    from fluidsimulationcuda_amd import capi
    n, nranks = 257, 3
    rcs = {}

    def body(s):
        ids = (C.c_int * 1)(capi.U)
        rcs[s.rank] = (capi.lib().fluid_exchange_now(s._h, capi.XCHG_HALO, ids, 1, 86),
                       capi.lib().fluid_exchange_now(s._h, capi.XCHG_HALO, ids, 1, 85))

    _, fab = run_ranks(n, nranks, 0, synthetic(n), body, jacobi=3)
    assert all(rcs[r] == (capi.E_INVALID, capi.OK) for r in range(nranks)), rcs
    for r in range(nranks):
        assert [e for e in fab.log[r] if e[0] == capi.XCHG_HALO] == [(capi.XCHG_HALO, (capi.U,), 85)]


def test_too_many_slabs_is_rejected():
    from fluidsimulationcuda_amd import capi
    from fluidsimulationcuda_amd.solver import FluidSolver
    with pytest.raises(capi.FluidError) as e:
        FluidSolver(7, rank=0, nranks=4)
    assert e.value.code == capi.E_INVALID
    with pytest.raises(capi.FluidError):
        FluidSolver(30, rank=2, nranks=2)


def test_missing_exchange_callback_is_an_error():
    from fluidsimulationcuda_amd import capi
    from fluidsimulationcuda_amd.solver import FluidSolver
    with FluidSolver(30, rank=0, nranks=2) as s:
        with pytest.raises(capi.FluidError) as e:
            s.step(1)
        assert e.value.code == capi.E_COMM


@pytest.mark.parametrize("iters", [40, 20, 28, 18, 6])
@pytest.mark.parametrize("n,nranks,halo", [(254, 2, 0), (510, 4, 16), (257, 3, 5)])
def test_fp16_storage_on_slabs_is_bit_identical_to_one_gpu(n, nranks, halo, iters):
    """Same contract with fp16 fields: the slabs exchange half rows and fuse the
    same launches per field, so they reproduce the single-context bits.  With fp16 storage every launch
    rounds once, so the schedule is part of the result: sweep counts whose schedule is not a row of
    eights (20 = 8 + 8 + 4, 28, 18, 6) must be split for the slab path's overlap at launch boundaries."""
    fields = synthetic(n)

    def body(s):
        s.step(1, use_sources=True, iters=iters)
        s.step(1, iters=iters)

    want = single(n, fields, body, storage=1)
    got, _ = run_ranks(n, nranks, halo, fields, body, jacobi=3, storage=1)
    for k in ("u", "v", "dens"):
        assert_bit_equal(got[k], want[k], "fp16 %s, %d slabs, %d sweeps" % (k, nranks, iters))


@pytest.mark.parametrize("n,nranks,halo", [(254, 2, 0), (510, 4, 16), (1022, 2, 42)])
def test_slabs_with_16_sweep_launches(n, nranks, halo):
    """The slab path with 16-sweep launches forced on (they are the default only on slabs of 8 M cells and
    more, which test_gpu_large.py runs once at 8192^2)."""
    from fluidsimulationcuda_amd import capi
    fields = synthetic(n, seed=9)

    def body(s):
        s.step(1, use_sources=True)
        s.step(1)

    want = single(n, fields, body)
    got, fab = run_ranks(n, nranks, halo, fields, body, jacobi=3, params={capi.PARAM_TB_T16_MIN_CELLS: 0})
    for r in range(1, nranks):
        assert fab.log[r] == fab.log[0]
    for k in ("u", "v", "dens"):
        assert_bit_equal(got[k], want[k], "%s, %d slabs, T=16 forced" % (k, nranks))


@pytest.mark.parametrize("storage", [0, 1])
@pytest.mark.parametrize("n,nranks", [(126, 2), (257, 3), (1022, 4)])
def test_advect_bounds_from_the_gradient_subtraction_are_the_slabs_own_maxima(F, n, nranks, storage):
    """Inside fluid_step / fluid_vel_step the bound of each advection -- max(|u|, |v|) over the slab's own rows -- is
    reduced by the gradient subtraction that produces the velocity (k_subtract_gradient_max + k_max_partials), not by
    k_absmax2.  Every value a rank brings to a MAX exchange is compared with numpy's maximum over that rank's rows of
    the same velocity, taken from a one-context replay of the step through the operators (FluidSequential.c:189-241)."""
    from fluidsimulationcuda_amd import capi
    from fluidsimulationcuda_amd.slab import slab_rows
    fields = synthetic(n, seed=31 + n)
    want = []                                           # (u, v) as each advection sees them
    av, bv = F.coefficients(n, DT, VISC)
    ad, bd = F.coefficients(n, DT, DIFF)
    with F.FluidSolver(n, storage=storage, params={capi.PARAM_TB_MIN_CELLS: 0}) as s:
        s.upload(**fields)
        for x, src in (("u", "u_prev"), ("v", "v_prev"), ("dens", "dens_prev")):
            s.add_source(x, src)
        s.diffuse(1, "u_prev", "u", av, bv)
        s.diffuse(2, "v_prev", "v", av, bv)
        s.diffuse(0, "dens_prev", "dens", ad, bd)
        s.computeDivergenceAndPressure("u_prev", "v_prev", "u", "v")
        s.diffuse(0, "u", "v", 1.0, 4.0)
        s.lastProject("u_prev", "v_prev", "u")
        want.append((s.download("u_prev"), s.download("v_prev")))
        s.advect(1, "u", "u_prev", "u_prev", "v_prev")
        s.advect(2, "v", "v_prev", "u_prev", "v_prev")
        s.computeDivergenceAndPressure("u", "v", "u_prev", "v_prev")
        s.diffuse(0, "u_prev", "v_prev", 1.0, 4.0)
        s.lastProject("u", "v", "u_prev")
        want.append((s.download("u"), s.download("v")))
        s.advect(0, "dens", "dens_prev", "u", "v")
        one = {k: s.download(k) for k in ("u", "v", "dens")}
    # (fp16 storage: the operators above store a plain divergence and pressure; a step keeps them scaled unless told not to)
    got, fab = run_ranks(n, nranks, 0, fields, lambda s: s.step(1, use_sources=True), jacobi=3, storage=storage,
                         params={capi.PARAM_F16_PRESSURE_SCALE: 0})
    for k in ("u", "v", "dens"):
        assert_bit_equal(got[k], one[k], "%s: step on %d slabs vs the operators in one context" % (k, nranks))
    for r in range(nranks):
        lo, hi = slab_rows(n, r, nranks)
        assert len(fab.maxima[r]) == 2
        for k, (u, v) in enumerate(want):
            m = max(np.abs(u[lo:hi, 1:n + 1]).max(), np.abs(v[lo:hi, 1:n + 1]).max())
            assert np.float32(fab.maxima[r][k]).view(np.uint32) == np.float32(m).view(np.uint32), (r, k, fab.maxima[r][k], m)


@pytest.mark.parametrize("grow,expect", [(0.5, "kept"), (40.0, "repeated"), (4000.0, "gathered")])
@pytest.mark.parametrize("entry", ["step", "vel_dens"])
def test_advection_started_on_the_previous_bound(F, grow, expect, entry):
    """Row slabs start each advection of a step on the bound of the step before (+25 %) while the new bound is still
    on its way to the host (FLUID_PARAM_EARLY_ADVECT, advect_bounded in fluid_solver.hip).  Velocity sources that
    shrink keep the early advection; sources that make the velocity jump force the exchange and the advection to be
    repeated with the real bound -- or with whole gathered fields when that bound outgrows a slab.  Every case bit-identical
    to one context, with and without the early start, and the exchange sequences show which way each advection went."""
    from fluidsimulationcuda_amd import capi
    n, nranks = 254, 3
    first = synthetic(n, seed=77)
    rng = np.random.default_rng(78)
    second = {k: (first[k] * np.float32(grow) + rnd(rng, n, -0.001, 0.001)).astype(np.float32) for k in ("u_prev", "v_prev", "dens_prev")}

    def body(s):
        def go():
            if entry == "step":
                s.step(1, use_sources=True)
            else:
                s.vel_step()
                s.dens_step()
        go()
        if isinstance(s, F.FluidSolver) and s.nranks > 1:
            s.load_global(**second)
        else:
            s.upload(**second)
        go()

    want = single(n, first, body)
    logs = {}
    for early in (1, 0):
        got, fab = run_ranks(n, nranks, 0, first, body, jacobi=3, params={capi.PARAM_EARLY_ADVECT: early})
        for k in ("u", "v", "dens", "u_prev", "v_prev", "dens_prev"):
            assert_bit_equal(got[k], want[k], "%s, early advect %d, sources x %g" % (k, early, grow))
        assert fab.log[0] == fab.log[1] == fab.log[2]
        logs[early] = [e[0] for e in fab.log[0]]
    H, G = capi.XCHG_HALO, capi.XCHG_GATHER
    halos = {k: v.count(H) for k, v in logs.items()}
    gathers = {k: v.count(G) for k, v in logs.items()}
    # first step: no previous bound, both runs alike.  Second step, two advections: an early start that holds costs no
    # exchange more than waiting would; one that does not hold has sent its halo rows for nothing, once per advection
    assert halos[1] - halos[0] == {"kept": 0, "repeated": 2, "gathered": 2}[expect], (halos, gathers)
    assert gathers[1] == gathers[0] == (2 if expect == "gathered" else 0), (halos, gathers)
