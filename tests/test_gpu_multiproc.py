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


def _clean_env():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return env


def test_bench_gpus_2_as_typed_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it (how a person -- or a driver that does not wrap it -- types
    it): it must start its ranks as child processes itself and relay rank 0's one JSON line, with both of north_star's
    grids in it (here the small --grid as the headline and 4096^2 beside it) and rank 0's single-GPU runs."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--exchange", "torch",
           "--grid", "512", "--steps", "2", "--warmup", "1"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=_clean_env(), cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line on stdout"
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "strong" and d["value"] > 0
    assert d["config"]["grid"] == 512 and d["native_exchange"] is False
    ov = d["exchange_overlap"]                             # measured both ways before the warm-up, the faster kept
    assert ov["chosen"] in (0, 1) and ov["overlap_ms_per_step"] > 0 and ov["in_line_ms_per_step"] > 0
    assert 0 < d["roofline"]["frac_compulsory"] < 1
    g = d["grid_4096"]
    assert g["value"] > 0 and g["ms_per_step"] > 0 and 0 < g["roofline"]["frac_compulsory"] < 1
    assert g["exchanges_per_rank"]["gather"] == 0 and 5 <= g["exchanges_per_rank"]["per_step"]["halo"] <= 6      # 2047-row slabs: deep ghost zones
    one = d["single_gpu"]
    assert one["512"]["ms_per_step"] > 0 and one["4096"]["ms_per_step"] > 0
    assert d["speedup_vs_single_gpu"] > 0 and g["speedup_vs_single_gpu"] > 0


def test_one_rank_without_librccl_falls_back_on_all_ranks_instead_of_hanging():
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "mp", "attach_one_rank_fails.py")]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=_clean_env(), cwd=ROOT)
    assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]
    assert "attach-failure ok rank 0" in p.stdout and "attach-failure ok rank 1" in p.stdout
