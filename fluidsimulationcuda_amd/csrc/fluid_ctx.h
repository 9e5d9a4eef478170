// fluid_ctx.h -- the context behind include/fluid_amd.h's opaque fluid_ctx, shared by the orchestrator
// (fluid_solver.hip) and the native RCCL exchange (fluid_exchange_rccl.hip).  Private to csrc/.
#pragma once
#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "../../include/fluid_amd.h"
#include "fluid_kernels.h"

namespace fluid_detail {
int fail(int code, const char* fmt, ...);      // records the calling thread's error string, returns `code`
struct RcclExchange;                            // fluid_exchange_rccl.hip
void rccl_release(RcclExchange* x);
}  // namespace fluid_detail

#define HIP_TRY(expr)                                                                                      \
    do {                                                                                                   \
        hipError_t e_ = (expr);                                                                            \
        if (e_ != hipSuccess)                                                                              \
            return fluid_detail::fail(e_ == hipErrorOutOfMemory ? FLUID_E_NOMEM : FLUID_E_HIP, "%s: %s",   \
                                      #expr, hipGetErrorString(e_));                                       \
    } while (0)

#define TRY(expr)                        \
    do {                                 \
        int rc_ = (expr);                \
        if (rc_ != FLUID_OK) return rc_; \
    } while (0)

struct fluid_ctx {
    int n = 0, w = 0, pitch = 0;
    size_t field_floats = 0;
    char* arena = nullptr;
    bool own_arena = false;
    int st = fluid::STORAGE_F32;          // field storage type
    size_t esz = 4;                       // bytes per stored element
    hipStream_t stream = nullptr;
    bool own_stream = false;
    hipStream_t stream2 = nullptr;        // slabs: the density diffusion runs beside the velocity path (full_step)
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    bool slab_overlap = true;
    // slabs: every exchange is enqueued on a stream of its own (one stream per communicator: the ranks issue their
    // collectives in one order), ordered against the compute stream(s) by events.  An exchange issued `async` is not waited
    // for at once: the first launch of the solve it feeds runs its interior strips -- rows that depend on this slab's own rows
    // only -- while the halo rows travel, and its edge strips behind the exchange's event (fluid_solver.hip: call_exchange,
    // op_diffuse_batch).  FLUID_PARAM_XCHG_OVERLAP.
    hipStream_t xstream = nullptr;
    hipEvent_t ev_xbegin = nullptr, ev_xdone = nullptr;
    bool xchg_overlap = true;
    bool xpend = false;                    // an async exchange is in flight: whoever consumes its rows waits on ev_xdone first
    long long split_launches = 0;          // first launches that ran as interior + edge strips around an exchange
    bool early_advect = true;                      // FLUID_PARAM_EARLY_ADVECT
    float vmax_prev[2] = {-1.0f, -1.0f};           // the last global bounds of the velocity / density advection (-1: none yet)
    void* f[FLUID_NFIELDS] = {};
    size_t field_bytes = 0;
    unsigned int* d_scalar = nullptr;     // device word for the reductions
    float* d_partials = nullptr;          // slabs: per-block maxima of the gradient subtraction (launch_subtract_gradient)
    unsigned int* tiles = nullptr;        // 3 x tile_rows x tile_pitch words: |x0| minima per tile for division mode 3
    unsigned int* h_scalar = nullptr;     // pinned host mirror
    hipEvent_t scalar_ready = nullptr;    // recorded behind the scalar's device-to-host copy
    int variant = fluid::JACOBI_TB;
    int tb_max_t = 16, tb_rows = 0, num_cu = 256;   // temporal blocking: sweeps/launch cap, rows/strip (0 = auto)
    long long tb_min_cells = 0;                    // smaller slabs use single-sweep launches (never faster since the 2-column lanes)
    long long tb_t16_min_cells = -1;               // >= 0: 16-sweep launches on every slab of at least this many cells (tests, tuning);
                                                   // -1: the measured rule of pick_sweeps()
    bool fuse_divergence = true;                   // a projection's divergence is computed inside its solve's first launch
    bool autotune = true;                          // strip heights of the fused kernel measured at run time (fluid_solver.hip: RbTuner)
    struct Trial { unsigned long long key; int cand; hipEvent_t a, b; };
    std::vector<Trial> trials;                     // launches being timed for the tuner
    std::vector<hipEvent_t> free_events;
    bool defer_zero_source = true;                 // see settle()
    bool in_halo_exchange = false;
    int tb_nv = 2;                                 // columns per lane of the fused kernel (2: 4 waves/SIMD; 4: 2 waves/SIMD)
    int tb_edge_pct = 40;                          // strip height of the two edge windows, % of the others'
    int fast_div = 2;                              // FLUID_PARAM_TB_FAST_DIVISION: 0 always divide; 2 (default) division modes 5 / 4;
                                                   // 3 modes 2 / 4 (round 2's default); 1 also the two-term mode 3 where |x0|
                                                   // allows it.  Every (mode, beta) is proven on the device first
    // slab decomposition
    int rank = 0, nranks = 1, own0 = 1, own1 = 1, min_slab = 0, halo = 1;
    int reach[FLUID_NFIELDS] = {};            // see "row-slab bookkeeping" below
    bool zero[FLUID_NFIELDS] = {};            // field is all +0 by definition; its memory is NOT (yet) zeroed
    bool pend[FLUID_NFIELDS] = {};            // field owes itself `+ pend_inc[f]` in every cell (deferred add_source of a zero source)
    float pend_inc[FLUID_NFIELDS] = {};
    // add_source of a real source field, deferred into the first launch of the diffusion that consumes the sum
    // (fluid_solver.hip: op_add_source / op_diffuse_batch): field f owes itself + src_dt[f] * (field src_of[f] - 1)
    int src_of[FLUID_NFIELDS] = {};           // 0: nothing owed; else 1 + the id of the source field
    float src_dt[FLUID_NFIELDS] = {};
    bool fuse_add_source = true;              // FLUID_PARAM_FUSE_ADD_SOURCE
    // fp16 storage: the pressure of a projection is of the order h * |velocity| -- 1e-5 at 16384^2, inside fp16's subnormal
    // range -- so inside a step the divergence and the pressure are kept multiplied by a power of two (fluid_solver.hip:
    // project); fscale[f] is that factor for field f (1: plain values), undone exactly when the field is downloaded and by a
    // pass over it for any reader that does not know
    float pscale = 1.0f;
    float fscale[FLUID_NFIELDS];
    fluid_exchange_fn xchg = nullptr;
    void* xchg_user = nullptr;
    fluid_detail::RcclExchange* rccl = nullptr;   // the library's own exchange, when attached (fluid_exchange_rccl_attach)
    // timing
    bool timing = false;
    struct Ev { hipEvent_t a, b; int cat; bool pressure; };
    std::vector<Ev> ev_pool;
    size_t ev_used = 0;
    double cat_ms[FLUID_TIMING_CATEGORIES] = {};
    long long cat_calls[FLUID_TIMING_CATEGORIES] = {};
    long long sweeps = 0, pending_sweeps = 0, launches = 0, field_launches = 0;
    double pressure_ms = 0.0;                      // the part of cat_ms[DIFFUSION] spent in pressure solves (project())
    long long pressure_sweeps = 0, pending_pressure_sweeps = 0;
    bool in_pressure_solve = false;

    bool valid_field(int id) const { return id >= 0 && id < FLUID_NFIELDS; }
    void* row(int id, int r) const { return static_cast<char*>(f[id]) + (size_t)r * pitch * esz; }
    int lo_all() const { return own0 - (rank == 0 ? 1 : 0); }          // owned rows incl. ghost row
    int hi_all() const { return own1 + (rank == nranks - 1 ? 1 : 0); }
};

