// mock_rccl.cpp -- TEST INFRASTRUCTURE: a stand-in for librccl whose "ranks" are threads of one process on one GPU.
//
// RCCL refuses two ranks on one device, and this pool hands out one GPU per call, so the library's own exchange
// (fluidsimulationcuda_amd/csrc/fluid_exchange_rccl.hip) could otherwise never run with more than one rank.  This file
// implements exactly the slice of the NCCL API that exchange binds (ncclGetUniqueId, ncclCommInitRank, ncclCommCount,
// ncclCommUserRank, ncclGroupStart/End, ncclSend, ncclRecv, ncclAllReduce(max, uint32), ncclBroadcast,
// ncclCommDestroy, ncclGetErrorString) with the semantics RCCL documents for them, moving the bytes with hipMemcpy
// between the ranks' device buffers:  sends and receives pair up per (peer, order of issue) inside a group; a group's
// operations take effect after everything earlier on the given stream; collectives must be called by every rank in
// the same order.  Mismatches (a receive without its send, different byte counts, a rank that leaves a collective out)
// are reported as errors instead of hanging where that can be told.  Loaded through FLUID_RCCL_LIB by
// tests/test_gpu_rccl_mock.py; it validates OUR use of the API (peers, row addresses, byte counts, grouping, ordering),
// not RCCL.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <condition_variable>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

namespace {

struct Op {
    enum Kind { SEND, RECV, ALLREDUCE, BCAST } kind;
    const void* src;
    void* dst;
    size_t bytes;
    int peer;          // SEND / RECV: the other rank; BCAST: root
    ncclRedOp_t red;
    ncclDataType_t type;
    hipStream_t stream;
};

struct World {
    int nranks = 0, joined = 0;
    std::mutex mu;
    std::condition_variable cv;
    int arrived = 0;
    unsigned long long generation = 0;
    std::vector<std::vector<Op>> ops;       // per rank: the group being executed
    std::vector<unsigned> scratch;          // all-reduce staging
    bool failed = false;
    std::string why;
    void barrier()
    {
        std::unique_lock<std::mutex> lk(mu);
        const unsigned long long gen = generation;
        if (++arrived == nranks) {
            arrived = 0;
            ++generation;
            cv.notify_all();
        } else {
            cv.wait(lk, [&] { return generation != gen; });
        }
    }
    void fail(const std::string& msg)
    {
        std::lock_guard<std::mutex> lk(mu);
        if (!failed) why = msg;
        failed = true;
    }
};

struct Comm {
    World* world;
    int rank;
};

std::mutex g_mu;
std::map<std::string, World*> g_worlds;
unsigned long long g_next_id = 1;
thread_local int t_depth = 0;
thread_local std::vector<std::pair<Comm*, Op>> t_pending;

size_t type_bytes(ncclDataType_t t)
{
    switch (t) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclFloat16: return 2;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
    default: return 0;
    }
}

ncclResult_t run_group()
{
    if (t_pending.empty()) return ncclSuccess;
    Comm* c = t_pending.front().first;
    World* w = c->world;
    for (auto& p : t_pending)
        if (p.first != c) return ncclInvalidUsage;
    // everything earlier on the operations' stream(s) has happened before the bytes move
    for (auto& p : t_pending)
        if (hipStreamSynchronize(p.second.stream) != hipSuccess) return ncclUnhandledCudaError;
    {
        std::lock_guard<std::mutex> lk(w->mu);
        w->ops[c->rank].clear();
        for (auto& p : t_pending) w->ops[c->rank].push_back(p.second);
    }
    t_pending.clear();
    w->barrier();                                           // every rank has posted its group
    const std::vector<Op>& mine = w->ops[c->rank];
    // --- all-reduces (uint32 max, in place allowed): read everybody's word, then write
    std::vector<size_t> reduces;
    for (size_t k = 0; k < mine.size(); ++k)
        if (mine[k].kind == Op::ALLREDUCE) reduces.push_back(k);
    std::vector<unsigned> result(reduces.size(), 0u);
    for (size_t q = 0; q < reduces.size(); ++q) {
        const Op& me = mine[reduces[q]];
        if (me.type != ncclUint32 || me.red != ncclMax || me.bytes != 4) { w->fail("mock: only ncclAllReduce(1 x uint32, max) is implemented"); break; }
        for (int r = 0; r < w->nranks; ++r) {
            size_t seen = 0;
            const Op* theirs = nullptr;
            for (const Op& o : w->ops[r])
                if (o.kind == Op::ALLREDUCE && seen++ == q) theirs = &o;
            if (!theirs) { w->fail("rank " + std::to_string(r) + " left an all-reduce out of a group rank " + std::to_string(c->rank) + " put one in"); break; }
            unsigned v = 0;
            if (hipMemcpy(&v, theirs->src, 4, hipMemcpyDeviceToHost) != hipSuccess) w->fail("hipMemcpy (all-reduce read)");
            result[q] = std::max(result[q], v);
        }
    }
    w->barrier();                                           // all reads of the (in-place) words are done
    for (size_t q = 0; q < reduces.size(); ++q)
        if (hipMemcpy(mine[reduces[q]].dst, &result[q], 4, hipMemcpyHostToDevice) != hipSuccess) w->fail("hipMemcpy (all-reduce write)");
    // --- receives: the k-th receive from peer p takes the k-th send of p to me
    std::map<int, size_t> nth;
    for (const Op& me : mine) {
        if (me.kind != Op::RECV) continue;
        const size_t k = nth[me.peer]++;
        if (me.peer < 0 || me.peer >= w->nranks) { w->fail("receive from rank " + std::to_string(me.peer)); continue; }
        size_t seen = 0;
        const Op* theirs = nullptr;
        for (const Op& o : w->ops[me.peer])
            if (o.kind == Op::SEND && o.peer == c->rank && seen++ == k) theirs = &o;
        if (!theirs) { w->fail("rank " + std::to_string(c->rank) + " receives from " + std::to_string(me.peer) + " what was never sent"); continue; }
        if (theirs->bytes != me.bytes) { w->fail("send / receive sizes differ: " + std::to_string(theirs->bytes) + " vs " + std::to_string(me.bytes)); continue; }
        if (hipMemcpy(me.dst, theirs->src, me.bytes, hipMemcpyDeviceToDevice) != hipSuccess) w->fail("hipMemcpy (receive)");
    }
    // every send must have been received
    std::map<int, size_t> sent;
    for (const Op& me : mine)
        if (me.kind == Op::SEND) {
            const size_t k = sent[me.peer]++;
            size_t recvs = 0;
            if (me.peer >= 0 && me.peer < w->nranks)
                for (const Op& o : w->ops[me.peer]) recvs += (o.kind == Op::RECV && o.peer == c->rank);
            if (recvs <= k) w->fail("rank " + std::to_string(c->rank) + " sends to " + std::to_string(me.peer) + " what is never received");
        }
    // --- broadcasts: everybody but the root copies the root's buffer of the same position in the group
    size_t bpos = 0;
    for (const Op& me : mine) {
        if (me.kind != Op::BCAST) continue;
        const size_t k = bpos++;
        if (me.peer == c->rank) continue;
        size_t seen = 0;
        const Op* theirs = nullptr;
        for (const Op& o : w->ops[me.peer])
            if (o.kind == Op::BCAST && seen++ == k) theirs = &o;
        if (!theirs || theirs->peer != me.peer || theirs->bytes != me.bytes) { w->fail("broadcasts do not line up across ranks"); continue; }
        if (hipMemcpy(me.dst, theirs->src, me.bytes, hipMemcpyDeviceToDevice) != hipSuccess) w->fail("hipMemcpy (broadcast)");
    }
    if (hipDeviceSynchronize() != hipSuccess) w->fail("hipDeviceSynchronize");
    w->barrier();                                           // nobody overwrites a buffer another rank is still reading
    return w->failed ? ncclInternalError : ncclSuccess;
}

ncclResult_t post(ncclComm_t comm, const Op& op)
{
    if (!comm) return ncclInvalidArgument;
    t_pending.emplace_back(reinterpret_cast<Comm*>(comm), op);
    return t_depth > 0 ? ncclSuccess : run_group();
}

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id)
{
    std::lock_guard<std::mutex> lk(g_mu);
    std::memset(id, 0, sizeof *id);
    std::snprintf(id->internal, sizeof id->internal, "mock-rccl-%llu", g_next_id++);
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank)
{
    if (!comm || nranks < 1 || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    World* w;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        World*& slot = g_worlds[std::string(id.internal, sizeof id.internal)];
        if (!slot) {
            slot = new World;
            slot->nranks = nranks;
            slot->ops.resize(nranks);
        }
        w = slot;
        if (w->nranks != nranks) return ncclInvalidArgument;
    }
    Comm* c = new Comm{w, rank};
    *comm = reinterpret_cast<ncclComm_t>(c);
    w->barrier();                                           // like the real thing: returns once every rank has joined
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm)
{
    delete reinterpret_cast<Comm*>(comm);
    return ncclSuccess;
}

ncclResult_t ncclCommCount(const ncclComm_t comm, int* count)
{
    *count = reinterpret_cast<Comm*>(comm)->world->nranks;
    return ncclSuccess;
}

ncclResult_t ncclCommUserRank(const ncclComm_t comm, int* rank)
{
    *rank = reinterpret_cast<Comm*>(comm)->rank;
    return ncclSuccess;
}

ncclResult_t ncclGroupStart()
{
    ++t_depth;
    return ncclSuccess;
}

ncclResult_t ncclGroupEnd()
{
    if (t_depth <= 0) return ncclInvalidUsage;
    if (--t_depth > 0) return ncclSuccess;
    return run_group();
}

ncclResult_t ncclSend(const void* buf, size_t count, ncclDataType_t type, int peer, ncclComm_t comm, hipStream_t stream)
{
    return post(comm, Op{Op::SEND, buf, nullptr, count * type_bytes(type), peer, ncclSum, type, stream});
}

ncclResult_t ncclRecv(void* buf, size_t count, ncclDataType_t type, int peer, ncclComm_t comm, hipStream_t stream)
{
    return post(comm, Op{Op::RECV, nullptr, buf, count * type_bytes(type), peer, ncclSum, type, stream});
}

ncclResult_t ncclAllReduce(const void* send, void* recv, size_t count, ncclDataType_t type, ncclRedOp_t op, ncclComm_t comm, hipStream_t stream)
{
    return post(comm, Op{Op::ALLREDUCE, send, recv, count * type_bytes(type), -1, op, type, stream});
}

ncclResult_t ncclBroadcast(const void* send, void* recv, size_t count, ncclDataType_t type, int root, ncclComm_t comm, hipStream_t stream)
{
    return post(comm, Op{Op::BCAST, send, recv, count * type_bytes(type), root, ncclSum, type, stream});
}

const char* ncclGetErrorString(ncclResult_t r)
{
    if (r == ncclSuccess) return "no error";
    std::lock_guard<std::mutex> lk(g_mu);
    static thread_local std::string msg;
    msg = "mock rccl error " + std::to_string((int)r);
    for (auto& kv : g_worlds)
        if (kv.second->failed) msg += ": " + kv.second->why;
    return msg.c_str();
}

}  // extern "C"
