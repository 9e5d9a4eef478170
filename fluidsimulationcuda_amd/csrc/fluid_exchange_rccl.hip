// fluid_exchange_rccl.hip -- the row-slab exchange of include/fluid_amd.h (fluid_exchange_fn) implemented inside
// the library on RCCL: halo rows as grouped ncclSend / ncclRecv between neighbouring slabs, the advect fall-back as
// grouped broadcasts, the velocity bound as an in-place ncclAllReduce(max) on the device scalar -- all enqueued on
// the context's stream with no host wait, so a C or C++ caller of libfluid_amd.so gets multi-GPU steps without any
// Python in the path.  The reference is single-device; BASELINE.json's north_star asks for
// "one-row ghost cells exchanged via RCCL Sendrecv over xGMI"; fluid_solver.hip decides when rows move and how deep.
//
// librccl is bound at run time (dlopen), not at link time: a process that already holds an RCCL (PyTorch ships its
// own copy next to its HIP runtime) must keep using that one, and a single-GPU user needs none.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstring>
#include <mutex>

#include "fluid_ctx.h"

namespace fluid_detail {

struct RcclApi {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string error;
};

// the process's RCCL
static RcclApi* rccl_api()
{
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        // $FLUID_RCCL_LIB wins (a site's own build; the tests' thread-ranks stand-in); else a copy the process already holds;
        // else the system's
        if (const char* env = std::getenv("FLUID_RCCL_LIB")) {
            api.handle = dlopen(env, RTLD_NOW | RTLD_GLOBAL);
            if (!api.handle) {
                const char* why = dlerror();             // (a second call returns NULL: the message is handed out once)
                api.error = std::string("cannot load $FLUID_RCCL_LIB: ") + (why ? why : env);
                return;
            }
        }
        const char* names[] = {"librccl.so", "librccl.so.1"};
        for (const char* nm : names)
            if (!api.handle) api.handle = dlopen(nm, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
        const char* fallbacks[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* nm : fallbacks)
            if (!api.handle) api.handle = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
        if (!api.handle) {
            const char* why = dlerror();
            api.error = std::string("cannot load librccl: ") + (why ? why : "not found");
            return;
        }
        auto bind = [&](auto& fn, const char* sym) {
            fn = reinterpret_cast<std::remove_reference_t<decltype(fn)>>(dlsym(api.handle, sym));
            if (!fn && api.error.empty()) api.error = std::string("librccl lacks ") + sym;
        };
        bind(api.GetUniqueId, "ncclGetUniqueId");
        bind(api.CommInitRank, "ncclCommInitRank");
        bind(api.CommDestroy, "ncclCommDestroy");
        bind(api.CommCount, "ncclCommCount");
        bind(api.CommUserRank, "ncclCommUserRank");
        bind(api.GroupStart, "ncclGroupStart");
        bind(api.GroupEnd, "ncclGroupEnd");
        bind(api.Send, "ncclSend");
        bind(api.Recv, "ncclRecv");
        bind(api.AllReduce, "ncclAllReduce");
        bind(api.Broadcast, "ncclBroadcast");
        bind(api.GetErrorString, "ncclGetErrorString");
    });
    return &api;
}

struct RcclExchange {
    fluid_ctx* ctx = nullptr;
    ncclComm_t comm = nullptr;
    bool own_comm = false;
    long long calls[3] = {0, 0, 0};     // halo, gather, max
};

#define NCCL_TRY(api, expr)                                                                                   \
    do {                                                                                                      \
        ncclResult_t r_ = (expr);                                                                             \
        if (r_ != ncclSuccess)                                                                                \
            return fail(FLUID_E_COMM, "%s: %s", #expr, (api)->GetErrorString ? (api)->GetErrorString(r_) : "RCCL error"); \
    } while (0)

// interior rows [lo, hi) of slab r (the split of fluid_create_ex)
static void slab_rows(int n, int r, int P, int* lo, int* hi)
{
    const int base = n / P, rem = n % P;
    *lo = 1 + r * base + (r < rem ? r : rem);
    *hi = *lo + base + (r < rem ? 1 : 0);
}

static int zero_if_marked(fluid_ctx* c, int f)
{
    // rows travel as they are in memory: a field that is zero only by definition gets its zeros now; a pending
    // add_source increment stays pending on every rank alike (fluid_solver.hip: need_list)
    if (!c->zero[f]) return FLUID_OK;
    HIP_TRY(hipMemsetAsync(c->f[f], 0, c->field_bytes, c->stream));
    c->zero[f] = false;
    return FLUID_OK;
}

static int rccl_exchange(void* user, int kind, const int* fields, int nfields, int depth, float* scalar)
{
    RcclExchange* x = static_cast<RcclExchange*>(user);
    fluid_ctx* c = x->ctx;
    RcclApi* api = rccl_api();
    const size_t row_bytes = (size_t)c->pitch * c->esz;
    switch (kind) {
    case FLUID_XCHG_HALO: {
        // against the SHORTEST slab (ranks differ by a row when N does not divide evenly): the verdict must be the same on
        // every rank, or the tall ranks enter the group while the short ones return
        if (depth < 1 || depth > c->min_slab) return fail(FLUID_E_COMM, "halo depth %d does not fit the slabs (shortest: %d rows)", depth, c->min_slab);
        x->calls[0] += 1;
        for (int k = 0; k < nfields; ++k) TRY(zero_if_marked(c, fields[k]));
        NCCL_TRY(api, api->GroupStart());
        for (int k = 0; k < nfields; ++k) {              // same order on every rank: sends and receives pair up
            const int f = fields[k];
            const size_t bytes = (size_t)depth * row_bytes;
            if (c->rank > 0) {
                NCCL_TRY(api, api->Send(c->row(f, c->own0), bytes, ncclUint8, c->rank - 1, x->comm, c->stream));
                NCCL_TRY(api, api->Recv(c->row(f, c->own0 - depth), bytes, ncclUint8, c->rank - 1, x->comm, c->stream));
            }
            if (c->rank < c->nranks - 1) {
                NCCL_TRY(api, api->Send(c->row(f, c->own1 - depth), bytes, ncclUint8, c->rank + 1, x->comm, c->stream));
                NCCL_TRY(api, api->Recv(c->row(f, c->own1), bytes, ncclUint8, c->rank + 1, x->comm, c->stream));
            }
        }
        NCCL_TRY(api, api->GroupEnd());
        return FLUID_OK;
    }
    case FLUID_XCHG_GATHER: {
        x->calls[1] += 1;
        for (int k = 0; k < nfields; ++k) TRY(zero_if_marked(c, fields[k]));
        NCCL_TRY(api, api->GroupStart());
        for (int k = 0; k < nfields; ++k)
            for (int r = 0; r < c->nranks; ++r) {
                int lo, hi;
                slab_rows(c->n, r, c->nranks, &lo, &hi);
                lo -= r == 0 ? 1 : 0;                    // the end slabs own the wall rows
                hi += r == c->nranks - 1 ? 1 : 0;
                void* rows = c->row(fields[k], lo);
                NCCL_TRY(api, api->Broadcast(rows, rows, (size_t)(hi - lo) * row_bytes, ncclUint8, r, x->comm, c->stream));
            }
        NCCL_TRY(api, api->GroupEnd());
        return FLUID_OK;
    }
    case FLUID_XCHG_MAX_BEGIN:
        // non-negative floats order like their bit patterns (and the kernel's atomicMax already works on those):
        // reducing the words as unsigned integers is exact, order independent and total even for NaN
        x->calls[2] += 1;
        NCCL_TRY(api, api->AllReduce(c->d_scalar, c->d_scalar, 1, ncclUint32, ncclMax, x->comm, c->stream));
        return FLUID_OK;
    case FLUID_XCHG_MAX_END:
        return FLUID_OK;                                 // *scalar already holds the reduced value (copied behind BEGIN)
    case FLUID_XCHG_MAX: {
        if (!scalar) return fail(FLUID_E_COMM, "FLUID_XCHG_MAX without a value");
        x->calls[2] += 1;
        unsigned bits;
        std::memcpy(&bits, scalar, sizeof bits);
        *c->h_scalar = bits;
        HIP_TRY(hipMemcpyAsync(c->d_scalar, c->h_scalar, sizeof bits, hipMemcpyHostToDevice, c->stream));
        NCCL_TRY(api, api->AllReduce(c->d_scalar, c->d_scalar, 1, ncclUint32, ncclMax, x->comm, c->stream));
        HIP_TRY(hipMemcpyAsync(c->h_scalar, c->d_scalar, sizeof bits, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        std::memcpy(scalar, c->h_scalar, sizeof bits);
        return FLUID_OK;
    }
    default:
        return fail(FLUID_E_COMM, "unknown exchange kind %d", kind);
    }
}

void rccl_release(RcclExchange* x)
{
    if (!x) return;
    if (x->own_comm && x->comm) (void)rccl_api()->CommDestroy(x->comm);
    delete x;
}

static int attach(fluid_ctx* c, ncclComm_t comm, bool own)
{
    RcclApi* api = rccl_api();
    int count = 0, rank = -1;
    NCCL_TRY(api, api->CommCount(comm, &count));
    NCCL_TRY(api, api->CommUserRank(comm, &rank));
    if (count != c->nranks || rank != c->rank)
        return fail(FLUID_E_INVALID, "communicator is rank %d of %d, the context is slab %d of %d", rank, count, c->rank, c->nranks);
    // a collective every rank takes part in before the first grouped send / receive (and RCCL's lazy set-up with it)
    HIP_TRY(hipMemsetAsync(c->d_scalar, 0, sizeof(unsigned), c->stream));
    NCCL_TRY(api, api->AllReduce(c->d_scalar, c->d_scalar, 1, ncclUint32, ncclMax, comm, c->stream));
    // ... and one point-to-point round trip (to itself: legal inside a group), so that a library whose send / receive
    // path does not work on this system fails here and not in the middle of a step
    unsigned* probe = c->d_scalar + 16;                  // spare words of the arena's 256-byte control block
    HIP_TRY(hipMemsetAsync(probe, 0, 2 * sizeof(unsigned), c->stream));
    HIP_TRY(hipMemsetAsync(probe, 0x5A, sizeof(unsigned), c->stream));
    NCCL_TRY(api, api->GroupStart());
    NCCL_TRY(api, api->Send(probe, sizeof(unsigned), ncclUint8, rank, comm, c->stream));
    NCCL_TRY(api, api->Recv(probe + 1, sizeof(unsigned), ncclUint8, rank, comm, c->stream));
    NCCL_TRY(api, api->GroupEnd());
    HIP_TRY(hipMemcpyAsync(c->h_scalar + 16, probe + 1, sizeof(unsigned), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (c->h_scalar[16] != 0x5A5A5A5Au) return fail(FLUID_E_COMM, "RCCL send/receive probe returned %08x", c->h_scalar[16]);
    rccl_release(c->rccl);
    c->rccl = new RcclExchange;
    c->rccl->ctx = c;
    c->rccl->comm = comm;
    c->rccl->own_comm = own;
    c->xchg = rccl_exchange;
    c->xchg_user = c->rccl;
    return FLUID_OK;
}

}  // namespace fluid_detail

using namespace fluid_detail;

extern "C" {

int fluid_rccl_available(void)
{
    RcclApi* api = rccl_api();
    if (!api->error.empty()) return fail(FLUID_E_COMM, "%s", api->error.c_str());
    return FLUID_OK;
}

int fluid_rccl_unique_id(void* id, size_t bytes)
{
    if (!id || bytes < FLUID_RCCL_ID_BYTES) return fail(FLUID_E_INVALID, "fluid_rccl_unique_id: need a buffer of %d bytes", FLUID_RCCL_ID_BYTES);
    static_assert(sizeof(ncclUniqueId) == FLUID_RCCL_ID_BYTES, "FLUID_RCCL_ID_BYTES must match ncclUniqueId");
    RcclApi* api = rccl_api();
    if (!api->error.empty()) return fail(FLUID_E_COMM, "%s", api->error.c_str());
    ncclUniqueId uid;
    NCCL_TRY(api, api->GetUniqueId(&uid));
    std::memcpy(id, &uid, sizeof uid);
    return FLUID_OK;
}

int fluid_exchange_rccl_attach(fluid_ctx* c, const void* id, size_t bytes)
{
    if (!c) return fail(FLUID_E_INVALID, "null context");
    if (!id || bytes < FLUID_RCCL_ID_BYTES) return fail(FLUID_E_INVALID, "fluid_exchange_rccl_attach: need the %d-byte id", FLUID_RCCL_ID_BYTES);
    RcclApi* api = rccl_api();
    if (!api->error.empty()) return fail(FLUID_E_COMM, "%s", api->error.c_str());
    ncclUniqueId uid;
    std::memcpy(&uid, id, sizeof uid);
    ncclComm_t comm = nullptr;
    NCCL_TRY(api, api->CommInitRank(&comm, c->nranks, uid, c->rank));     // on the calling thread's current HIP device
    const int rc = attach(c, comm, true);
    if (rc != FLUID_OK) (void)api->CommDestroy(comm);
    return rc;
}

int fluid_exchange_rccl_attach_comm(fluid_ctx* c, void* nccl_comm)
{
    if (!c || !nccl_comm) return fail(FLUID_E_INVALID, "null argument");
    RcclApi* api = rccl_api();
    if (!api->error.empty()) return fail(FLUID_E_COMM, "%s", api->error.c_str());
    return attach(c, static_cast<ncclComm_t>(nccl_comm), false);
}

int fluid_exchange_rccl_detach(fluid_ctx* c)
{
    if (!c) return fail(FLUID_E_INVALID, "null context");
    if (c->rccl) {
        if (c->stream) (void)hipStreamSynchronize(c->stream);
        if (c->xchg_user == c->rccl) {
            c->xchg = nullptr;
            c->xchg_user = nullptr;
        }
        rccl_release(c->rccl);
        c->rccl = nullptr;
    }
    return FLUID_OK;
}

int fluid_exchange_rccl_calls(fluid_ctx* c, long long* halo, long long* gather, long long* max)
{
    if (!c) return fail(FLUID_E_INVALID, "null context");
    if (!c->rccl) return fail(FLUID_E_COMM, "no RCCL exchange attached");
    if (halo) *halo = c->rccl->calls[0];
    if (gather) *gather = c->rccl->calls[1];
    if (max) *max = c->rccl->calls[2];
    return FLUID_OK;
}

}  // extern "C"
