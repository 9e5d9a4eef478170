"""Row-slab multi-GPU layer: one process (and one libfluid_amd context) per GPU.

The reference is single-device (no NCCL/MPI anywhere), so this
has no reference counterpart; what it must preserve is the result: every cell
goes through the same arithmetic as on one GPU, so an N-GPU run is bit-identical
to the 1-GPU run.

Division of labour: the C++ orchestrator (csrc/fluid_solver.hip) decides *when*
rows have to move and calls back; this module only moves them, with
torch.distributed point-to-point ops (backend "nccl" = RCCL over xGMI on the
GPU box, "gloo" in the CPU tests) on torch tensors that alias the context's
device arena.  Every rank keeps full-size fields (288 GB of HBM per GPU makes
that free: 12 fields x 260 MiB at 8192^2), computes only its slab, and global row
r is local row r -- so a halo is just rows [own-depth, own) and the advect
fallback is an in-place all-gather.
"""
import numpy as np
import torch
import torch.distributed as dist

from . import capi
from .solver import FluidSolver


def slab_rows(n, rank, nranks):
    """Interior rows [lo, hi) of slab `rank` (same split as fluid_create_ex)."""
    base, rem = divmod(n, nranks)
    lo = 1 + rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class TorchExchange:
    """The fluid_exchange_fn of include/fluid_amd.h over torch.distributed.

    `fields(id)` returns the [n+2, pitch] tensor view of field `id` as it is
    placed right now (fields may trade buffers inside a solve); rows of it are
    contiguous, so a halo needs no packing."""

    def __init__(self, fields, n, rank, nranks, group=None, stream=None, stream_of=None):
        if not callable(fields):
            table = fields
            fields = table.__getitem__
        self.fields, self.n, self.rank, self.nranks, self.group = fields, n, rank, nranks, group
        self.stream = stream        # torch.cuda.Stream the solver's kernels run on (None: CPU tensors)
        self.stream_of = stream_of  # () -> hipStream_t the library wants this exchange on (its exchange stream), or None
        self._ext = {}
        self.lo, self.hi = slab_rows(n, rank, nranks)
        self.calls = {capi.XCHG_HALO: 0, capi.XCHG_GATHER: 0, capi.XCHG_MAX: 0}
        # RCCL moves device memory directly.  gloo cannot, so device rows are
        # staged through host buffers: used only to rehearse the multi-process
        # path on a box with fewer GPUs than ranks (tests, bench --backend gloo).
        self.staged = fields(0).is_cuda and dist.get_backend(group) != "nccl"
        # Where the velocity bound is reduced is fixed here, identically on every rank (it follows from the
        # backend alone): on the device word, in place, when the transport addresses device memory (set_scalar),
        # else on the host at END.  Ranks that disagreed would issue different collective sequences and hang.
        self.scalar = None          # 1-element int32 view of the solver's device reduction word

    def _peer(self, r):
        return dist.get_global_rank(self.group, r) if self.group is not None else r

    def __call__(self, kind, ids, depth, scalar):
        # collectives must order against the solver's kernels: make its stream
        # torch's current stream for the duration of the exchange
        if self.stream is not None:
            stream = self.stream
            if self.stream_of is not None:
                # the library runs exchanges on a stream of its own (FLUID_PARAM_XCHG_OVERLAP), ordered against its kernels by
                # events: enqueue there, so that the first launch of the solve can run beside the rows in flight
                ptr = self.stream_of()
                if ptr and ptr != self.stream.cuda_stream:
                    stream = self._ext.get(ptr)
                    if stream is None:
                        stream = self._ext[ptr] = torch.cuda.ExternalStream(ptr, device=self.stream.device)
            with torch.cuda.stream(stream):
                return self._dispatch(kind, ids, depth, scalar)
        return self._dispatch(kind, ids, depth, scalar)

    def _dispatch(self, kind, ids, depth, scalar):
        if kind == capi.XCHG_MAX_BEGIN:
            self.calls[capi.XCHG_MAX] += 1
            if self.scalar is not None:
                # the word holds the bit pattern of a non-negative float (the kernel's atomicMax works on it as an
                # unsigned integer): MAX over the patterns is exact, order independent and total even for NaN
                dist.all_reduce(self.scalar, op=dist.ReduceOp.MAX, group=self.group)
            return None
        if kind == capi.XCHG_MAX_END:
            return scalar if self.scalar is not None else self.maximum(scalar)
        self.calls[kind] += 1
        if kind == capi.XCHG_HALO:
            return self.halo(ids, depth)
        if kind == capi.XCHG_GATHER:
            return self.gather(ids)
        if kind == capi.XCHG_MAX:
            return self.maximum(scalar)
        raise ValueError("unknown exchange kind %r" % kind)

    def halo(self, ids, depth):
        lo, hi = self.lo, self.hi
        shortest = self.n // self.nranks                 # the same verdict on every rank (slabs differ by a row when N % P != 0)
        if depth < 1 or depth > shortest:
            raise ValueError("halo depth %d does not fit the slabs (shortest: %d rows)" % (depth, shortest))
        ops, landing = [], []

        def send(rows, peer):
            ops.append(dist.P2POp(dist.isend, rows.cpu() if self.staged else rows, peer, self.group))

        def recv(rows, peer):
            buf = torch.empty(rows.shape, dtype=rows.dtype) if self.staged else rows
            ops.append(dist.P2POp(dist.irecv, buf, peer, self.group))
            if self.staged:
                landing.append((rows, buf))

        for fid in ids:             # same order on every rank: sends and receives pair up
            f = self.fields(fid)
            if self.rank > 0:
                up = self._peer(self.rank - 1)
                send(f[lo:lo + depth], up)
                recv(f[lo - depth:lo], up)
            if self.rank < self.nranks - 1:
                dn = self._peer(self.rank + 1)
                send(f[hi - depth:hi], dn)
                recv(f[hi:hi + depth], dn)
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        for rows, buf in landing:
            rows.copy_(buf)

    def gather(self, ids):
        for fid in ids:
            f = self.fields(fid)
            for r in range(self.nranks):
                lo, hi = slab_rows(self.n, r, self.nranks)
                lo -= 1 if r == 0 else 0                    # end slabs own the wall rows
                hi += 1 if r == self.nranks - 1 else 0
                if self.staged:
                    buf = f[lo:hi].cpu()
                    dist.broadcast(buf, src=self._peer(r), group=self.group)
                    if r != self.rank:
                        f[lo:hi].copy_(buf)
                else:
                    dist.broadcast(f[lo:hi], src=self._peer(r), group=self.group)

    def set_scalar(self, word):
        """`word`: 1-element int32 tensor aliasing the solver's device reduction word.  Taken only when the
        transport addresses device memory (backend nccl) -- the same decision on every rank."""
        self.scalar = word if (word is not None and word.is_cuda and not self.staged) else None

    def maximum(self, value):
        dev = "cpu" if (self.staged or not self.fields(0).is_cuda) else self.fields(0).device
        bits = int(np.float32(value).view(np.int32))       # non-negative floats order like their bit patterns
        t = torch.tensor([bits], dtype=torch.int32, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        return float(np.int32(t.item()).view(np.float32))


class SlabSolver(FluidSolver):
    """FluidSolver for rank `rank` of `nranks`, device memory owned by torch so
    RCCL can address it; kernels run on torch's current stream so the
    collectives order against them without host synchronisation."""

    def __init__(self, n, rank=None, nranks=None, halo=0, jacobi=capi.JACOBI_TB, device=None, group=None,
                 storage=capi.STORAGE_F32, params=None, exchange="torch"):
        """exchange: "torch" -- rows move through torch.distributed (TorchExchange, a Python callback; works on
        every backend, host-staged on gloo); "rccl" -- the library's own exchange (csrc/fluid_exchange_rccl.hip:
        grouped ncclSend/ncclRecv on the solver's stream, no Python and no host wait in the path); its
        communicator's id is handed round with one torch.distributed broadcast."""
        if rank is None:
            rank = dist.get_rank(group) if dist.is_initialized() else 0
        if nranks is None:
            nranks = dist.get_world_size(group) if dist.is_initialized() else 1
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device())
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("SlabSolver computes with HIP kernels only; got device %s" % device)
        L = capi.lib()
        nbytes = L.fluid_arena_bytes_ex(n, storage)
        if nbytes == 0:
            raise ValueError("bad N or storage type")
        import ctypes as C
        pitch, xoff, ff = C.c_int(), C.c_int(), C.c_size_t()
        capi.check(L.fluid_layout(n, C.byref(pitch), C.byref(xoff), C.byref(ff)))
        self.pitch, self.xoff = pitch.value, xoff.value
        with torch.cuda.device(self.device):
            self.arena = torch.zeros(nbytes, dtype=torch.uint8, device=self.device)
            # a real (non-null) stream: handle 0 would mean "library-owned stream"
            # to fluid_create_ex, and nothing would order RCCL against the kernels
            self.torch_stream = torch.cuda.Stream(device=self.device)
            torch.cuda.synchronize(self.device)        # the zero-fill ran on torch's default stream
            super().__init__(n, rank=rank, nranks=nranks, halo=halo, jacobi=jacobi,
                             stream=self.torch_stream.cuda_stream,
                             arena_ptr=self.arena.data_ptr(), arena_bytes=nbytes, storage=storage, params=params)
        esz, dt = (2, torch.float16) if storage == capi.STORAGE_F16 else (4, torch.float32)
        self._fb = ff.value * esz           # bytes per field
        self._views = [self.arena[k * self._fb:(k + 1) * self._fb].view(dt).view(n + 2, self.pitch)
                       for k in range(capi.NFIELDS)]
        self.exchange = None
        self.native_exchange = False
        if nranks > 1 and exchange in ("rccl", "auto"):
            self._bring_up_native_exchange(L, group, exchange)
        if nranks > 1 and not self.native_exchange:
            if dist.get_backend(group) == "nccl":
                # batched send/recv must not be the first operation on a NCCL group: start with an all-reduce
                hello = torch.ones(1, device=self.device)
                dist.all_reduce(hello, group=group)
                torch.cuda.synchronize(self.device)
            self.exchange = TorchExchange(self.field_tensor, n, rank, nranks, group, stream=self.torch_stream,
                                          stream_of=self.exchange_stream)
            off = self.scalar_ptr() - self.arena.data_ptr()
            self.exchange.set_scalar(self.arena[off:off + 4].view(torch.int32))
            self.set_exchange(self.exchange)

    def _bring_up_native_exchange(self, L, group, exchange):
        """The library's own RCCL exchange, or -- under "auto" -- the agreement of ALL ranks to do without it.

        Every step is followed by an agreement (a MIN all-reduce over torch.distributed) before the next collective
        step is entered, and nothing raises before the ranks have agreed: a rank that failed alone and left (or raised)
        would leave its peers waiting inside a collective -- ncclCommInitRank, the id broadcast, the all-reduce itself --
        until the launcher kills the job.  With "rccl" a failure raises on every rank; with "auto" every rank detaches
        and takes the torch.distributed exchange."""
        import ctypes as C
        on_dev = dist.get_backend(group) == "nccl"
        where = self.device if on_dev else "cpu"

        def all_ok(ok):
            t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=where)
            dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
            return int(t.item()) == 1

        def give_up(what):
            capi.check(L.fluid_exchange_rccl_detach(self._h))
            if exchange == "rccl":
                raise RuntimeError("the library's RCCL exchange is not available on every rank: " + what)

        # 1. can every rank load and bind librccl?  (no communication inside; a rank that cannot would return from the
        #    attach below before ncclCommInitRank, which the others are already blocked in)
        mine = L.fluid_rccl_available() == capi.OK
        why = "" if mine else L.fluid_last_error().decode(errors="replace")
        if not all_ok(mine):
            return give_up("librccl could not be bound" + (" here: " + why if why else " on another rank"))
        # 2. the communicator's id from rank 0; a failure travels as an all-zero id, which every rank sees
        uid = torch.zeros(capi.RCCL_ID_BYTES, dtype=torch.uint8, device=where)
        if self.rank == 0:
            buf = (C.c_ubyte * capi.RCCL_ID_BYTES)()
            if L.fluid_rccl_unique_id(buf, capi.RCCL_ID_BYTES) == capi.OK:
                uid.copy_(torch.frombuffer(bytearray(buf), dtype=torch.uint8))
        src = dist.get_global_rank(group, 0) if group is not None else 0
        dist.broadcast(uid, src=src, group=group)
        raw = bytes(uid.cpu().numpy().tobytes())
        if not any(raw):
            return give_up("rank 0 could not create a communicator id")
        # 3. the collective attach: every rank enters it (step 1 made sure of that); its result is agreed on afterwards
        with torch.cuda.device(self.device):
            rc = L.fluid_exchange_rccl_attach(self._h, raw, len(raw))
        why = "" if rc == capi.OK else L.fluid_last_error().decode(errors="replace")
        if not all_ok(rc == capi.OK):
            return give_up("attach failed" + (" here: " + why if why else " on another rank"))
        # 4. one real exchange before anything depends on it: every rank marks a scratch field with its number, two halo
        #    rows travel each way, and each rank checks whose rows arrived
        if not all_ok(self._native_exchange_selftest() == 1):
            return give_up("its self-test delivered the wrong rows")
        self.native_exchange = True

    def _native_exchange_selftest(self):
        import ctypes as C
        lo, hi = self.owned_rows
        depth = min(2, hi - lo)
        self.fill("tmp2", float(self.rank + 1))
        ids = (C.c_int * 1)(capi.TMP2)
        if capi.lib().fluid_exchange_now(self._h, capi.XCHG_HALO, ids, 1, depth) != capi.OK:
            return 0
        self.synchronize()
        rows = self.field_tensor(capi.TMP2)
        ok = True
        if self.rank > 0:
            ok &= bool((rows[lo - depth:lo, self.xoff:self.xoff + self.n + 2] == float(self.rank)).all())
        if self.rank < self.nranks - 1:
            ok &= bool((rows[hi:hi + depth, self.xoff:self.xoff + self.n + 2] == float(self.rank + 2)).all())
        ok &= bool((rows[lo:hi, self.xoff:self.xoff + self.n + 2] == float(self.rank + 1)).all())
        self.fill("tmp2", 0.0)
        return 1 if ok else 0

    def field_tensor(self, fid):
        """[n+2, pitch] view of the buffer field `fid` occupies right now."""
        slot, rem = divmod(self.field_ptr(fid) - self.arena.data_ptr(), self._fb)
        assert rem == 0 and 0 <= slot < capi.NFIELDS
        return self._views[slot]

    def interior(self, field):
        """[n+2, n+2] strided view of a field (ghost ring included)."""
        fid = capi.FIELD_NAMES.index(field) if isinstance(field, str) else int(field)
        return self.field_tensor(fid)[:, self.xoff:self.xoff + self.n + 2]

    def load_global(self, **host_fields):
        """Every rank uploads the rows it computes on (its slab + wall rows);
        the rest of its copy stays zero until an exchange fills it."""
        lo, hi = self.owned_rows
        lo -= 1 if self.rank == 0 else 0
        hi += 1 if self.rank == self.nranks - 1 else 0
        for name, arr in host_fields.items():
            self.upload_rows(name, arr, lo, hi)

    def gather_global(self, field):
        """Full field on every rank's host (tests / result collection):
        an in-place all-gather of the slabs, then one download (bit preserving)."""
        fid = capi.FIELD_NAMES.index(field) if isinstance(field, str) else int(field)
        if self.nranks > 1:
            import ctypes as C
            ids = (C.c_int * 1)(fid)
            capi.check(capi.lib().fluid_exchange_now(self._h, capi.XCHG_GATHER, ids, 1, 0))
            torch.cuda.synchronize(self.device)
        return self.download(fid)

    def exchange_calls(self):
        """{halo, gather, max} exchanges this rank has issued."""
        if self.native_exchange:
            import ctypes as C
            h, g, m = C.c_longlong(), C.c_longlong(), C.c_longlong()
            capi.check(capi.lib().fluid_exchange_rccl_calls(self._h, C.byref(h), C.byref(g), C.byref(m)))
            return {capi.XCHG_HALO: h.value, capi.XCHG_GATHER: g.value, capi.XCHG_MAX: m.value}
        return dict(self.exchange.calls) if self.exchange else None
