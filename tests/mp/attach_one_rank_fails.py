"""Run by tests/test_gpu_multiproc.py under torch.distributed.run with two ranks (gloo) on one GPU: rank 1 cannot load
librccl ($FLUID_RCCL_LIB names a missing file there, and only there).  Before round 3 that rank returned from
fluid_exchange_rccl_attach before ncclCommInitRank while rank 0 waited inside it; now every rank reports whether it can
bind the library and all agree BEFORE anyone attaches.  exchange="auto" must fall back to torch.distributed on both
ranks and still step; exchange="rccl" must raise on both.  Prints one line per rank; any hang ends in the test's timeout."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
rank = int(os.environ["RANK"])
if rank == 1:
    os.environ["FLUID_RCCL_LIB"] = "/nonexistent/librccl.so"      # before the library is loaded

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from fluidsimulationcuda_amd import capi  # noqa: E402
from fluidsimulationcuda_amd.harness import initialize_parameters  # noqa: E402
from fluidsimulationcuda_amd.slab import SlabSolver  # noqa: E402

torch.cuda.set_device(0)
dist.init_process_group("gloo")
avail = capi.lib().fluid_rccl_available()
assert (avail == capi.OK) == (rank == 0), "rank %d: fluid_rccl_available() = %d" % (rank, avail)
n = 126
s = SlabSolver(n, exchange="auto")
assert not s.native_exchange and s.exchange is not None, "rank %d did not fall back" % rank
s.load_global(**initialize_parameters(n, seed=3))
s.step(1, use_sources=True)
s.step(1)
got = s.gather_global("u")
s.close()
assert np.isfinite(got).all()
raised = False
try:
    SlabSolver(n, exchange="rccl")
except RuntimeError as exc:
    raised = "not available on every rank" in str(exc)
assert raised, "rank %d: exchange='rccl' must raise on every rank when one of them cannot bind librccl" % rank
dist.barrier()
print("attach-failure ok rank %d" % rank, flush=True)
dist.destroy_process_group()
