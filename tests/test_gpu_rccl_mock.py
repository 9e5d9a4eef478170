"""GPU: the library's own exchange (csrc/fluid_exchange_rccl.hip) with MORE THAN ONE RANK.  RCCL itself refuses two
ranks on one device, so the ranks here are threads of one process and librccl is replaced (FLUID_RCCL_LIB) by
tests/mock_rccl/mock_rccl.cpp, which implements the NCCL calls the exchange binds -- grouped send/receive pairing,
in-place all-reduce(max) on a device word, grouped broadcasts, stream ordering, a communicator per rank -- with
hipMemcpy between the ranks' device buffers, and reports mismatched collectives instead of hanging.  What this pins is
OUR side: which rows go to which peer at which byte counts, the grouping, the order of collectives on every rank, the
attach-time probe, the gather used to collect results.  Bar: the ranks' steps reproduce a single context bit for bit."""
import os
import shutil
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.fixture(scope="module")
def mock_lib(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("needs hipcc to build the stand-in")
    so = str(tmp_path_factory.mktemp("mock") / "libmock_rccl.so")
    subprocess.check_call([HIPCC, "-O1", "-std=c++17", "-shared", "-fPIC", "-I/opt/rocm/include",
                           os.path.join(ROOT, "tests", "mock_rccl", "mock_rccl.cpp"), "-o", so], stderr=subprocess.DEVNULL)
    return so


@pytest.mark.parametrize("n,nranks,halo,storage,iters,big", [
    (254, 2, 0, 0, 40, False),        # default ghost zones
    (257, 3, 5, 0, 40, False),        # uneven slabs, shallow ghost zones: exchanges inside the solves
    (510, 4, 42, 0, 40, False),       # deep ghost zones: second-stream overlap, one exchange per solve
    (1022, 8, 0, 0, 40, False),       # eight ranks
    (254, 2, 40, 1, 20, False),       # fp16 rows (half the bytes), 8 + 8 + 4 schedule
    (126, 4, 4, 0, 8, True),          # back-traces longer than a slab: FLUID_XCHG_GATHER through grouped broadcasts
    (254, 3, 0, 0, 40, "grow"),       # a fourth step whose velocity jumps: the early advections are repeated (advect_bounded)
])
def test_thread_ranks_through_the_native_exchange(mock_lib, n, nranks, halo, storage, iters, big):
    env = dict(os.environ, FLUID_RCCL_LIB=mock_lib)
    cmd = [sys.executable, os.path.join(ROOT, "tests", "mock_rccl", "run_ranks.py"), str(n), str(nranks), str(halo), str(storage),
           str(iters)] + (["grow"] if big == "grow" else ["big"] if big else [])
    steps = 4 if big == "grow" else 3
    big = big is True
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-3000:])
    last = p.stdout.strip().splitlines()[-1]
    assert last.startswith("ok: %d ranks" % nranks), last
    counts = dict(kv.split("=") for kv in last.split("in the steps: ")[1].split())
    assert int(counts["max"]) == 2 * steps and int(counts["halo"]) >= 3                  # two advect bounds per step
    assert (int(counts["gather"]) > 0) == big, "the gather fall-back runs exactly when back-traces outgrow a slab: " + last
