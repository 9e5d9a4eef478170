"""CPU, world_size 2 and 3 over gloo: the torch.distributed side of the row-slab
path (fluidsimulationcuda_amd/slab.py TorchExchange) -- halo rows land in the
right rows of the right neighbour, the gather fallback reassembles whole fields,
MAX reduces.  The same code runs over RCCL on the GPU box (backend "nccl")."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, q):
    import sys
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from fluidsimulationcuda_amd import capi
        from fluidsimulationcuda_amd.slab import TorchExchange, slab_rows
        pitch = 96
        w = n + 2
        # field value encodes (field id, owner rank, row): wrong routing is visible
        lo, hi = slab_rows(n, rank, world)

        def make(fid):
            f = torch.full((w, pitch), -1.0)
            a = lo - (1 if rank == 0 else 0)
            b = hi + (1 if rank == world - 1 else 0)
            for r in range(a, b):
                f[r] = fid * 1000 + r
            return f

        fields = [make(fid) for fid in range(capi.NFIELDS)]
        ex = TorchExchange(fields, n, rank, world)
        depth = 3
        ex(capi.XCHG_HALO, [0, 4], depth, None)
        ok = True
        for fid in (0, 4):
            f = fields[fid]
            if rank > 0:
                ok &= bool((f[lo - depth:lo, 0] == torch.tensor([fid * 1000.0 + r for r in range(lo - depth, lo)])).all())
                ok &= bool((f[lo - depth - 1, 0] == -1) or (lo - depth - 1 < 1 and rank == 0))
            if rank < world - 1:
                ok &= bool((f[hi:hi + depth, 0] == torch.tensor([fid * 1000.0 + r for r in range(hi, hi + depth)])).all())
                ok &= bool(f[hi + depth, 0] == -1)
        ok &= bool((fields[1] == make(1)).all())            # unlisted fields untouched
        got = ex(capi.XCHG_MAX, [], 0, float(rank * 2 + 1))
        ok &= got == float((world - 1) * 2 + 1)
        # split form: CPU tensors cannot be reduced "on the device", so END does the work
        ok &= ex(capi.XCHG_MAX_BEGIN, [], 0, None) is None
        ok &= ex(capi.XCHG_MAX_END, [], 0, float(10 - rank)) == 10.0
        ex(capi.XCHG_GATHER, [2], 0, None)
        ok &= bool((fields[2][:, 5] == torch.tensor([2000.0 + r for r in range(w)])).all())
        try:
            ex(capi.XCHG_HALO, [0], hi - lo + 1, None)
            ok = False
        except ValueError:
            pass
        q.put((rank, ok, dict(ex.calls)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n", [(2, 30), (3, 31)])
def test_torch_exchange_over_gloo(world, n):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, world, port, n, q)) for r in range(world)]
    [p.start() for p in ps]
    res = [q.get(timeout=120) for _ in range(world)]
    [p.join(timeout=60) for p in ps]
    assert all(p.exitcode == 0 for p in ps)
    for rank, ok, calls in res:
        assert ok, "rank %d saw wrong rows" % rank
        assert calls[0] == 2 and calls[1] == 1 and calls[2] == 2


def test_slab_rows_partition():
    from fluidsimulationcuda_amd.slab import slab_rows
    for n in (1, 7, 30, 126, 4094, 8190):
        for p in (1, 2, 3, 8):
            if n // p < 1:
                continue
            edges = [slab_rows(n, r, p) for r in range(p)]
            assert edges[0][0] == 1 and edges[-1][1] == n + 1
            assert all(edges[k][1] == edges[k + 1][0] for k in range(p - 1))
            sizes = [b - a for a, b in edges]
            assert max(sizes) - min(sizes) <= 1 and min(sizes) == n // p
