"""GPU: the library's own RCCL exchange (csrc/fluid_exchange_rccl.hip) as far as ONE GPU can take it: librccl is
found and bound at run time, a one-rank communicator comes up on the context's device, the attach-time all-reduce and
send/receive probe run on the context's stream, and the exchange entry points (gather = grouped broadcasts, the MAX
all-reduce on the device word) execute through it.  RCCL refuses two ranks on one device, so rows moving between
slabs are covered by the fake-fabric tests (test_gpu_slab.py) and the gloo tests (test_exchange_gloo.py) on the same
orchestration; the multi-GPU run itself is the driver's (bench.py --gpus N)."""
import ctypes as C

import numpy as np
import pytest

from conftest import assert_bit_equal, rnd

pytestmark = pytest.mark.gpu


def test_one_rank_communicator_attach_exchange_detach():
    import fluidsimulationcuda_amd as F
    from fluidsimulationcuda_amd import capi
    L = capi.lib()
    n = 126
    rng = np.random.default_rng(1)
    x = rnd(rng, n)
    with F.FluidSolver(n) as s:
        s.upload(u=x, v=x * 2)
        uid = (C.c_ubyte * capi.RCCL_ID_BYTES)()
        capi.check(L.fluid_rccl_unique_id(uid, capi.RCCL_ID_BYTES))
        assert any(uid), "ncclGetUniqueId left the id empty"
        capi.check(L.fluid_exchange_rccl_attach(s._h, uid, capi.RCCL_ID_BYTES))
        ids = (C.c_int * 2)(capi.U, capi.V)
        capi.check(L.fluid_exchange_now(s._h, capi.XCHG_GATHER, ids, 2, 0))
        capi.check(L.fluid_exchange_now(s._h, capi.XCHG_HALO, ids, 2, 3))     # one slab: nothing to send, the group is empty
        s.synchronize()
        assert_bit_equal(s.download("u"), x, "in-place broadcast of the only slab")
        assert s.absmax_velocity("u", "v") == np.abs(x[1:-1, 1:-1] * 2).max()   # FLUID_XCHG_MAX through ncclAllReduce
        h, g, m = C.c_longlong(), C.c_longlong(), C.c_longlong()
        capi.check(L.fluid_exchange_rccl_calls(s._h, C.byref(h), C.byref(g), C.byref(m)))
        assert (h.value, g.value) == (1, 1)
        s.step(1)                                    # a one-slab step never calls out
        capi.check(L.fluid_exchange_rccl_detach(s._h))
        with pytest.raises(capi.FluidError):
            capi.check(L.fluid_exchange_rccl_calls(s._h, None, None, None))


def test_attach_rejects_a_mismatched_communicator():
    import fluidsimulationcuda_amd as F
    from fluidsimulationcuda_amd import capi
    L = capi.lib()
    with pytest.raises(capi.FluidError):
        capi.check(L.fluid_rccl_unique_id(None, 0))
    with F.FluidSolver(30) as s:
        with pytest.raises(capi.FluidError):
            capi.check(L.fluid_exchange_rccl_attach(s._h, None, 0))
