"""GPU: the multi-process launch of bench.py as the driver starts it (python -m torch.distributed.run, one process per
rank), rehearsed with two ranks on the ONE GPU of this box: rows staged through the host over gloo (RCCL refuses two
ranks on one device).  Covers what the in-process fake fabric cannot: process groups, SlabSolver's construction under
torch.distributed.run, and -- with --exchange try-rccl -- the attempt to bring up the library's own RCCL exchange, its
failure here (duplicate device), and the all-reduce by which every rank agrees to fall back to the torch.distributed
exchange instead of hanging on mismatched transports.  --check compares the slabs' result with a one-context run bit
for bit."""
import json
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("exchange", ["torch", "try-rccl"])
def test_two_rank_bench_over_gloo_on_one_gpu(exchange):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo",
           "--exchange", exchange, "--grid", "512", "--steps", "2", "--warmup", "1", "--check"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    assert "check ok: 2 slabs bit-identical to one context at 512x512" in p.stderr
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "rank 0 prints exactly one JSON line"
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "strong" and d["value"] > 0
    x = d["exchanges_per_rank"]                          # warm-up + timed steps of the measured context
    # two advect bounds per step; halo exchanges: 5 per step with ghost zones of 40+ rows, more on these 255-row slabs (31)
    assert x["gather"] == 0 and x["max"] == 2 * 3 and 5 * 3 <= x["halo"] <= 12 * 3, x
    assert "gloo" in d["config"]["parallelism"]          # the native exchange cannot come up with two ranks on one device
