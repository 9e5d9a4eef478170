#!/usr/bin/env python3
"""TEST INFRASTRUCTURE (run as a subprocess by tests/test_gpu_rccl_mock.py with FLUID_RCCL_LIB pointing at the thread-ranks
stand-in built from mock_rccl.cpp): P contexts, one thread each, all on this one GPU, exchange rows through the LIBRARY'S
OWN exchange (csrc/fluid_exchange_rccl.hip) -- attach, steps, gather -- and must reproduce a single context bit for bit.
    run_ranks.py N NRANKS HALO STORAGE ITERS [big | grow]
big: velocities whose back-traces outgrow a slab (the gather fall-back); grow: a fourth step whose sources make the
velocity jump, so that the advections started on the previous step's bound must be repeated"""
import ctypes as C
import os
import sys
import threading

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))                    # tests/
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))   # repo root

from fluidsimulationcuda_amd import capi  # noqa: E402
from test_gpu_slab import _init_fake, single, synthetic  # noqa: E402
from fluidsimulationcuda_amd.slab import SlabSolver  # noqa: E402

n, nranks, halo, storage, iters = (int(x) for x in sys.argv[1:6])
big = len(sys.argv) > 6 and sys.argv[6] == "big"
grow = len(sys.argv) > 6 and sys.argv[6] == "grow"
assert os.environ.get("FLUID_RCCL_LIB"), "meant to run against the stand-in"
L = capi.lib()
fields = synthetic(n, seed=n + nranks)
if big:                                                     # back-traces longer than a slab: the gather fall-back
    rng = np.random.default_rng(5)
    fields["u_prev"] = (fields["u_prev"] * 5000).astype(np.float32)
    fields["v_prev"] = (rng.random(fields["v_prev"].shape, dtype=np.float32) * 5000).astype(np.float32)


more = {k: (fields[k] * np.float32(60)).astype(np.float32) for k in ("u_prev", "v_prev", "dens_prev")}


def body(s):
    s.step(1, use_sources=True, iters=iters)
    s.step(2, iters=iters)
    if grow:
        if s.nranks > 1:
            s.load_global(**more)
        else:
            s.upload(**more)
        s.step(1, use_sources=True, iters=iters)


want = single(n, fields, body, storage=storage)
uid = (C.c_ubyte * capi.RCCL_ID_BYTES)()
capi.check(L.fluid_rccl_unique_id(uid, capi.RCCL_ID_BYTES))
solvers, errs, out = [], [], {}
for r in range(nranks):
    s = SlabSolver.__new__(SlabSolver)
    _init_fake(s, n, r, nranks, halo, 3, storage, None)
    s.native_exchange = True
    solvers.append(s)


def work(r):
    try:
        s = solvers[r]
        capi.check(L.fluid_exchange_rccl_attach(s._h, uid, capi.RCCL_ID_BYTES))     # blocks until every rank has joined
        assert s._native_exchange_selftest() == 1, "rows of the attach-time self-test arrived wrong"   # what SlabSolver runs
        s.load_global(**fields)
        body(s)
        calls = s.exchange_calls()                                                    # of the steps alone
        got = {k: s.gather_global(k) for k in ("u", "v", "dens")}                     # fluid_exchange_now(GATHER) on every rank
        out[r] = (got, calls)
        capi.check(L.fluid_exchange_rccl_detach(s._h))
    except Exception as e:      # noqa: BLE001
        errs.append((r, repr(e)))
        os._exit(3 if not errs[1:] else 4)                  # a rank that died would leave the others in a barrier


ts = [threading.Thread(target=work, args=(r,)) for r in range(nranks)]
[t.start() for t in ts]
[t.join() for t in ts]
assert not errs, errs
for r in range(nranks):
    got, calls = out[r]
    for k in ("u", "v", "dens"):
        same = np.array_equal(got[k].view(np.uint32), want[k].view(np.uint32))
        if not same:
            bad = int((got[k].view(np.uint32) != want[k].view(np.uint32)).sum())
            sys.exit("rank %d: %s differs from the single context in %d cells" % (r, k, bad))
    assert calls == out[0][1], "ranks counted different exchanges: %r vs %r" % (calls, out[0][1])
c = out[0][1]
print("ok: %d ranks (threads) through the library's exchange == one context; in the steps: halo=%d gather=%d max=%d"
      % (nranks, c[capi.XCHG_HALO], c[capi.XCHG_GATHER], c[capi.XCHG_MAX]))
