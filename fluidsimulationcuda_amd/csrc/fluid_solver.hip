// fluid_solver.hip -- C++ host orchestrator + the extern "C" shim of
// include/fluid_amd.h.  Owns device memory, sequences the kernels of
// fluid_kernels.hip exactly as the reference's vel_step / dens_step do
// (project/sequential/FluidSequential.c:176-241), and drives the row-slab halo
// exchange through a callback when the grid is split over several GPUs.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <functional>
#include <mutex>
#include <new>
#include <string>
#include <unordered_map>
#include <vector>

#include "fluid_ctx.h"

namespace fluid_detail {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

}  // namespace fluid_detail

namespace {

using fluid_detail::fail;
using fluid_detail::g_err;

constexpr size_t kControlBytes = 256;   // tail of the arena: reduction scalar (+0), division-proof counter (+8)
constexpr int kMaxN = 65533;     // one grid row per blockIdx.y in the pointwise kernels (HIP: gridDim.y <= 65535 = N + 2);
                                 // 65535^2 x 9 fields is 155 GB of the 288 GB, so nothing practical is cut off

}  // namespace

namespace {

using fluid::XOFF;

int check_ctx(const fluid_ctx* c)
{
    if (!c) return fail(FLUID_E_INVALID, "null context");
    return FLUID_OK;
}

int check_fields(const fluid_ctx* c, std::initializer_list<int> ids)
{
    for (int id : ids)
        if (!c->valid_field(id)) return fail(FLUID_E_INVALID, "bad field id %d", id);
    return FLUID_OK;
}

// Every exchange goes through here.  One issued `async` (slabs, FLUID_PARAM_XCHG_OVERLAP) runs with the exchange stream as
// "the context's stream": behind everything enqueued on the compute stream so far (event), and the compute stream behind it
// again only when xchg_join() is called by whoever consumes the rows.  A callback that enqueues elsewhere or waits on the
// host (the tests' in-process fabric) is merely not overlapped.  The ranks issue their collectives in one order whichever
// stream each goes to (RCCL serialises a communicator's operations in issue order across streams).
int call_exchange(fluid_ctx* c, int kind, const int* ids, int count, int depth, float* scalar, bool async = false)
{
    if (c->xpend) {                                      // one exchange in flight at a time: the earlier one is joined first
        HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_xdone, 0));
        c->xpend = false;
    }
    // An exchange the caller waits for at once stays in line on the compute stream: a hop to another stream and back
    // costs ~10 us each way on this platform (measured: a no-op exchange routed through the second stream leaves the GPU
    // idle for 20 us), which only an exchange that runs beside a launch can pay for.
    if (!async || !c->xstream || !c->xchg_overlap) return c->xchg(c->xchg_user, kind, ids, count, depth, scalar);
    HIP_TRY(hipEventRecord(c->ev_xbegin, c->stream));
    HIP_TRY(hipStreamWaitEvent(c->xstream, c->ev_xbegin, 0));
    hipStream_t compute = c->stream;
    c->stream = c->xstream;
    const int rc = c->xchg(c->xchg_user, kind, ids, count, depth, scalar);
    c->stream = compute;
    if (rc != 0) return rc;
    HIP_TRY(hipEventRecord(c->ev_xdone, c->xstream));
    c->xpend = true;
    return 0;
}

// the current compute stream waits for the exchange in flight (if any); `keep`: another stream still has to join it too
int xchg_join(fluid_ctx* c, bool keep = false)
{
    if (!c->xpend) return FLUID_OK;
    HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_xdone, 0));
    if (!keep) c->xpend = false;
    return FLUID_OK;
}

int exchange(fluid_ctx* c, int kind, std::initializer_list<int> fields, int depth, float* scalar = nullptr)
{
    if (c->nranks == 1 && !c->rccl) return FLUID_OK;      // (a one-rank communicator may be attached: it is then exercised)
    if (!c->xchg) return fail(FLUID_E_COMM, "multi-GPU context without an exchange callback");
    std::vector<int> ids(fields);
    // the same field listed twice (self-advection) is exchanged once
    std::sort(ids.begin(), ids.end());
    ids.erase(std::unique(ids.begin(), ids.end()), ids.end());
    const int rc = call_exchange(c, kind, ids.data(), (int)ids.size(), depth, scalar);
    if (rc != 0) return fail(FLUID_E_COMM, "exchange callback failed (kind %d, rc %d)", kind, rc);
    return FLUID_OK;
}

int reduce_to_host(fluid_ctx* c, float* out)
{
    HIP_TRY(hipMemcpyAsync(c->h_scalar, c->d_scalar, sizeof(unsigned), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    float v;
    std::memcpy(&v, c->h_scalar, sizeof v);
    *out = v;
    return FLUID_OK;
}

// ---- timing ---------------------------------------------------------------
int timing_begin(fluid_ctx* c, int cat, hipEvent_t* stop_out)
{
    *stop_out = nullptr;
    if (!c->timing) return FLUID_OK;
    if (c->ev_used == c->ev_pool.size()) {
        fluid_ctx::Ev e{};
        HIP_TRY(hipEventCreate(&e.a));
        HIP_TRY(hipEventCreate(&e.b));
        c->ev_pool.push_back(e);
    }
    auto& p = c->ev_pool[c->ev_used++];
    p.cat = cat;
    p.pressure = cat == FLUID_TIME_DIFFUSION && c->in_pressure_solve;
    HIP_TRY(hipEventRecord(p.a, c->stream));
    *stop_out = p.b;
    return FLUID_OK;
}

int timing_end(fluid_ctx* c, hipEvent_t stop, int sweeps)
{
    if (!stop) return FLUID_OK;
    HIP_TRY(hipEventRecord(stop, c->stream));
    c->pending_sweeps += sweeps;
    if (c->in_pressure_solve) c->pending_pressure_sweeps += sweeps;
    return FLUID_OK;
}

int timing_collect(fluid_ctx* c)
{
    if (c->ev_used == 0) return FLUID_OK;
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (c->stream2) HIP_TRY(hipStreamSynchronize(c->stream2));
    for (size_t k = 0; k < c->ev_used; ++k) {
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, c->ev_pool[k].a, c->ev_pool[k].b));
        c->cat_ms[c->ev_pool[k].cat] += ms;
        c->cat_calls[c->ev_pool[k].cat] += 1;
        if (c->ev_pool[k].pressure) c->pressure_ms += ms;
    }
    c->sweeps += c->pending_sweeps;
    c->pending_sweeps = 0;
    c->pressure_sweeps += c->pending_pressure_sweeps;
    c->pending_pressure_sweeps = 0;
    c->ev_used = 0;
    return FLUID_OK;
}

// RAII-free helper: time one operator launch under category `cat`
#define TIMED(c, cat, stmt)                      \
    do {                                         \
        hipEvent_t stop_;                        \
        TRY(timing_begin((c), (cat), &stop_));   \
        stmt;                                    \
        TRY(timing_end((c), stop_, 0));          \
    } while (0)

// Division mode for `beta` in the temporally blocked kernel (fluid_kernels.hip, DIVMODE): 0 = true division,
// 4 = multiply by the exact reciprocal (beta a power of two, alpha 1), 5 = scaled residual correction (four packed
// instructions per pair of cells, any data), 3 = two-term reciprocal where |x0| allows it (hi, lo; two packed
// instructions), 2 = f64 reciprocal multiply (six scalar ones).  Modes 2, 3, 4 and 5 are only
// used after k_validate_div has proven them against a/beta for every one of the 2^32 float inputs on this
// device -- a few ms, once per (mode, beta) and PROCESS: the proof is about the arithmetic of the device
// type, so contexts share it (a fresh context used to spend 3 x 2.5 ms re-proving the step's three betas).
struct DivProofs {
    std::mutex mu;
    std::unordered_map<unsigned long long, int> proven;     // (device, wanted mode, beta bits) -> mode to use
};
DivProofs& div_proofs()
{
    static DivProofs p;
    return p;
}

struct DivPlan {
    int mode = 0;
    float arg = 0.f;      // what the kernel receives as `beta`: beta (0, 2, 3), 1/beta (4)
    float hi = 0.f, lo = 0.f;   // mode 3: the two-term reciprocal; mode 5: beta * 2^24 and -(RN32(1/beta) * 2^-24)
    unsigned tile_thr = 0;      // mode 3: bits of beta * 2^-72, what |x0| must reach on a tile (fluid_kernels.hip, DIVMODE 3)
    double yd = 0.0;      // modes 2, 3
};

float round_down_to_float(double v)
{
    float f = (float)v;
    if ((double)f > v) f = std::nextafterf(f, -INFINITY);
    return f;
}

DivPlan division_mode(fluid_ctx* c, float beta, float alpha)
{
    DivPlan plan;
    plan.arg = beta;
    plan.yd = 1.0 / (double)beta;
    if (c->fast_div == 0 || !(beta > 0.f) || !std::isfinite(beta)) return plan;
    int e2 = 0;
    const float rbeta = 1.0f / beta;
    const bool pow2 = std::frexp(beta, &e2) == 0.5f && std::isnormal(rbeta) && rbeta * beta == 1.0f;
    const float hi = round_down_to_float(plan.yd), lo = (float)(plan.yd - (double)hi);
    // mode 4 (pressure solve: alpha 1, beta 4): multiply by the exact reciprocal, and x * 1.0f is x so
    // alpha is not applied at all; else mode 3 when 1/beta splits into two normal floats with lo > 0
    // (not for powers of two: lo = 0 turns inf * lo into NaN); else mode 2, the double-precision reciprocal.
    // A mode that fails its proof hands over to the next: 3 -> 2 -> 0, 4 -> 2 -> 0, 5 -> 2 -> 0.
    const bool two_term = c->fast_div == 1 && c->tiles && beta >= 1.0f && beta <= 0x1p24f && std::isnormal(hi) && std::isnormal(lo) && lo > 0.f;
    // mode 5 (the default for every other beta): r = RN32(1/beta), beta * 2^24 and r * 2^-24 must be ordinary numbers
    const bool residual = c->fast_div == 2 && beta >= 0x1p-60f && beta <= 0x1p60f;
    const float r5_hi = beta * 0x1p24f, r5_lo = -(rbeta * 0x1p-24f);
    int want = (pow2 && alpha == 1.0f) ? 4 : two_term ? 3 : residual ? 5 : 2;
    unsigned bits;
    std::memcpy(&bits, &beta, sizeof bits);
    int dev = 0;
    (void)hipGetDevice(&dev);
    DivProofs& proofs = div_proofs();
    std::lock_guard<std::mutex> lock(proofs.mu);
    int mode = 0;
    while (want != 0) {
        const unsigned long long key = ((unsigned long long)(dev & 0xFF) << 40) | ((unsigned long long)want << 32) | bits;
        auto it = proofs.proven.find(key);
        if (it != proofs.proven.end()) {
            mode = it->second;
        } else {
            unsigned long long* bad = reinterpret_cast<unsigned long long*>(c->d_scalar) + 1;   // 8-byte slot of the 256-B block
            unsigned long long* hbad = reinterpret_cast<unsigned long long*>(c->h_scalar) + 1;
            if (hipMemsetAsync(bad, 0, sizeof *bad, c->stream) != hipSuccess) return plan;
            fluid::launch_validate_div(c->stream, want, beta, (want == 4 || want == 5) ? rbeta : beta, plan.yd, want == 5 ? r5_hi : hi,
                                       want == 5 ? r5_lo : lo, bad);
            if (hipMemcpyAsync(hbad, bad, sizeof *bad, hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
                hipStreamSynchronize(c->stream) != hipSuccess)
                return plan;
            mode = *hbad == 0 ? want : 0;
            proofs.proven.emplace(key, mode);
        }
        if (mode != 0) break;
        want = want == 2 ? 0 : 2;
    }
    plan.mode = mode;
    if (mode == 4 || mode == 5) plan.arg = rbeta;
    if (mode == 5) {
        plan.hi = r5_hi;
        plan.lo = r5_lo;
    }
    if (mode == 3) {
        plan.hi = hi;
        plan.lo = lo;
        const float thr = beta * 0x1p-72f;
        std::memcpy(&plan.tile_thr, &thr, sizeof thr);
    }
    return plan;
}

// ---- row-slab bookkeeping ------------------------------------------------------
// reach[f] = how many rows beyond each INNER edge of this slab currently hold the
// same values as their owner's copy (kEverywhere: the field is known identical
// on all ranks, e.g. freshly zeroed).  Writers set it (an operator that computes
// `r` rows past the slab leaves reach r), need() raises it with ONE exchange for
// all the fields that fall short.  With one slab everything is a no-op.
constexpr int kEverywhere = 1 << 28;

int exchange_cap(const fluid_ctx* c) { return c->min_slab - 1; }     // rows a neighbour can always supply

void wrote(fluid_ctx* c, int f, int reach)
{
    c->reach[f] = c->nranks > 1 ? reach : kEverywhere;
    c->pend[f] = false;                        // overwritten: whatever the old contents still owed is moot
    c->src_of[f] = 0;
    c->fscale[f] = 1.0f;                       // (a writer that keeps a scale sets it again afterwards)
}

// `async`: the compute stream does not wait for the rows (call_exchange); the caller's next solve joins them (xchg_join)
int need_list(fluid_ctx* c, const std::vector<int>& fields, int reach, bool async = false)
{
    if (c->nranks == 1 || reach <= 0) return FLUID_OK;
    if (reach > exchange_cap(c)) return fail(FLUID_E_COMM, "halo of %d rows exceeds the slab height", reach);
    std::vector<int> ids;
    for (int f : fields)
        if (c->reach[f] < reach) ids.push_back(f);
    if (ids.empty()) return FLUID_OK;
    if (!c->xchg) return fail(FLUID_E_COMM, "multi-GPU context without an exchange callback");
    std::sort(ids.begin(), ids.end());
    ids.erase(std::unique(ids.begin(), ids.end()), ids.end());
    c->in_halo_exchange = true;            // rows travel as they are in memory; a pending increment stays pending on every rank alike
    const int rc = call_exchange(c, FLUID_XCHG_HALO, ids.data(), (int)ids.size(), reach, nullptr, async);
    c->in_halo_exchange = false;
    if (rc != 0) return fail(FLUID_E_COMM, "halo exchange failed (rc %d)", rc);
    for (int f : ids) c->reach[f] = reach;
    return FLUID_OK;
}

int need(fluid_ctx* c, std::initializer_list<int> fields, int reach, bool async = false)
{
    return need_list(c, std::vector<int>(fields), reach, async);
}

// interior rows [lo,hi) this slab computes when it works `reach` rows past its inner edges
void rows(const fluid_ctx* c, int reach, int* lo, int* hi)
{
    *lo = std::max(1, c->own0 - reach);
    *hi = std::min(c->n + 1, c->own1 + reach);
}

// ---- fields that are zero by definition ---------------------------------------------
// The sources of every step after the first and the pressure's first guess are
// all +0.  Writing those zeros and reading them back is pure traffic, so such a
// field is only MARKED zero; the three consumers that matter take the mark
// (add_source adds the constant dt*0, the fused Jacobi kernel reads nothing, the
// divergence kernel skips its p stores) and everything else materialises the
// zeros first.
// add_source with a source that is zero by definition adds the constant dt*0 to every cell: it
// changes nothing but the sign of -0 (and NaN/inf rules for a non-finite dt), yet costs a read and a
// write of the whole field.  With the fused Jacobi kernel the increment stays PENDING instead: the
// solve that consumes the field as its right-hand side adds it to each row as it loads it, any other
// reader settles it with the real kernel first (here), and a writer that replaces the field drops it.
int settle_source(fluid_ctx* c, int f);

// a field kept scaled (fscale) goes back to plain values: every row of it, one multiplication by a power of two
int unscale(fluid_ctx* c, int f)
{
    if (c->fscale[f] == 1.0f) return FLUID_OK;
    const float inv = 1.0f / c->fscale[f];
    c->fscale[f] = 1.0f;
    if (c->zero[f]) return FLUID_OK;
    fluid::launch_scale(c->stream, c->st, c->f[f], c->pitch, 0, c->n + 2, inv);
    return FLUID_OK;
}

int settle(fluid_ctx* c, int f, bool keep_scale = false)
{
    if (!keep_scale) TRY(unscale(c, f));
    if (c->src_of[f]) return settle_source(c, f);
    if (!c->pend[f]) return FLUID_OK;
    const int reach = c->nranks > 1 ? std::min(c->reach[f], exchange_cap(c)) : 0;
    int lo, hi;
    rows(c, reach, &lo, &hi);
    if (lo == 1) lo = 0;
    if (hi == c->n + 1) hi = c->n + 2;
    const float inc = c->pend_inc[f];
    c->pend[f] = false;
    TIMED(c, FLUID_TIME_SOURCE, fluid::launch_add_source(c->stream, c->st, c->f[f], nullptr, c->pitch, lo, hi, inc));
    return FLUID_OK;
}

// the deferred add_source of a real source field only (a pending constant increment stays pending: the fused kernel
// applies it on load)
int settle_source(fluid_ctx* c, int f)
{
    if (c->src_of[f]) {
        // an add_source of a real source field that no diffusion launch took over (see op_add_source): the kernel of its own
        const int s = c->src_of[f] - 1;
        const int reach = c->nranks > 1 ? std::min({c->reach[f], c->reach[s], exchange_cap(c)}) : 0;
        int lo, hi;
        rows(c, reach, &lo, &hi);
        if (lo == 1) lo = 0;
        if (hi == c->n + 1) hi = c->n + 2;
        c->src_of[f] = 0;
        TIMED(c, FLUID_TIME_SOURCE, fluid::launch_add_source(c->stream, c->st, c->f[f], c->f[s], c->pitch, lo, hi, c->src_dt[f]));
        wrote(c, f, reach);
    }
    return FLUID_OK;
}

int materialize_zero(fluid_ctx* c, int f)
{
    if (!c->zero[f]) return FLUID_OK;
    HIP_TRY(hipMemsetAsync(c->f[f], 0, c->field_bytes, c->stream));
    c->zero[f] = false;
    return FLUID_OK;
}

int materialize(fluid_ctx* c, int f, bool keep_scale = false)
{
    TRY(materialize_zero(c, f));
    return settle(c, f, keep_scale);
}

int materialize(fluid_ctx* c, std::initializer_list<int> fs)
{
    for (int f : fs) TRY(materialize(c, f));
    return FLUID_OK;
}

void mark_zero(fluid_ctx* c, int f)
{
    c->zero[f] = true;
    c->pend[f] = false;
    c->src_of[f] = 0;
    c->fscale[f] = 1.0f;
    c->reach[f] = kEverywhere;
}

// ---- operators -------------------------------------------------------------------
// `defer` (only the step functions pass it: nothing can touch x or s between this call and the diffusion that follows it
// there): with the fused Jacobi kernel a real source is not added now -- the first launch of the solve whose right-hand
// side x is and whose first guess s is (FluidSequential.c:181, :201, :209: SWAP, then diffuse) reads both fields anyway,
// forms x + dt*s as it loads them and stores the sum out of place (op_diffuse_batch).  Any other reader settles it first.
int op_add_source(fluid_ctx* c, int x, int s, float dt, bool defer = false)
{
    // pointwise: valid as far out as both operands are
    const int reach = c->nranks > 1 ? std::min({c->reach[x], c->reach[s], exchange_cap(c)}) : 0;
    int lo, hi;
    rows(c, reach, &lo, &hi);
    if (lo == 1) lo = 0;                     // wall rows are cells like any other here (FluidSequential.c:78-82)
    if (hi == c->n + 1) hi = c->n + 2;
    TRY(materialize(c, x));
    if (c->zero[s]) {
        volatile float z = 0.0f;
        const float inc = dt * z;          // the reference's dt * s[i] with s[i] = +0 (sign and NaN rules included)
        if (c->variant == fluid::JACOBI_TB && c->defer_zero_source) {
            c->pend[x] = true;             // (x was settled just above: one pending increment at a time)
            c->pend_inc[x] = inc;
            return FLUID_OK;               // reach[x] unchanged: nothing was written
        }
        TIMED(c, FLUID_TIME_SOURCE, fluid::launch_add_source(c->stream, c->st, c->f[x], nullptr, c->pitch, lo, hi, inc));
    } else {
        TRY(settle(c, s));                 // (a source that is itself owed something: never inside a step)
        if (defer && c->fuse_add_source && c->variant == fluid::JACOBI_TB && c->tb_nv == 2) {
            c->src_of[x] = 1 + s;
            c->src_dt[x] = dt;
            return FLUID_OK;               // reach[x] unchanged: nothing was written
        }
        TIMED(c, FLUID_TIME_SOURCE, fluid::launch_add_source(c->stream, c->st, c->f[x], c->f[s], c->pitch, lo, hi, dt));
    }
    wrote(c, x, reach);
    return FLUID_OK;
}

// ---- strip-height tuner -----------------------------------------------------------------------------------
// The fused Jacobi kernel's speed depends on the strip height `rb` in ways no closed form captured (how the
// (window, strip) blocks fall into rounds on the 256 CUs and 8 XCDs, how much pipeline fill they repeat): over a sweep
// of heights the closed-form pick was 10-23 % off the best on most shapes (tools/slab_rb_sweep.py).  Results do not
// depend on rb, so the library measures instead: for each launch shape -- sweeps per launch, fields per launch, form,
// rows, N, lane width, storage -- the first launches cycle through a handful of heights with an event pair around
// each, and once every candidate has two samples the fastest is kept.  The table is shared by all contexts of the
// process (a benchmark can tune in a throw-away context); harvesting is by hipEventQuery, never a wait.
struct RbTuner {
    struct Entry {
        std::vector<int> cand;
        std::vector<float> best_ms;
        std::vector<int> samples, issued;
        int fixed = 0;
    };
    std::mutex mu;
    std::unordered_map<unsigned long long, Entry> table;
};
RbTuner& rb_tuner()
{
    static RbTuner t;
    return t;
}
constexpr int kTuneSamples = 2;

unsigned long long tune_key(const fluid_ctx* c, int T, int m, int divmode, long long rows_n)
{
    int dev = 0;
    (void)hipGetDevice(&dev);
    unsigned long long h = 1469598103934665603ull;
    for (unsigned long long v : {(unsigned long long)dev, (unsigned long long)c->n, (unsigned long long)rows_n, (unsigned long long)T,
                                 (unsigned long long)m, (unsigned long long)divmode, (unsigned long long)c->tb_nv, (unsigned long long)c->st,
                                 (unsigned long long)c->tb_edge_pct}) {
        h ^= v;
        h *= 1099511628211ull;
    }
    return h;
}

// finished trials -> the table (never waits)
void tune_harvest(fluid_ctx* c)
{
    if (c->trials.empty()) return;
    RbTuner& t = rb_tuner();
    std::lock_guard<std::mutex> lock(t.mu);
    size_t keep = 0;
    for (size_t k = 0; k < c->trials.size(); ++k) {
        fluid_ctx::Trial& tr = c->trials[k];
        if (hipEventQuery(tr.b) != hipSuccess) {
            c->trials[keep++] = tr;
            continue;
        }
        float ms = 0.f;
        auto it = t.table.find(tr.key);
        if (hipEventElapsedTime(&ms, tr.a, tr.b) == hipSuccess && it != t.table.end() && !it->second.fixed) {
            RbTuner::Entry& e = it->second;
            e.best_ms[tr.cand] = e.samples[tr.cand] ? std::min(e.best_ms[tr.cand], ms) : ms;
            e.samples[tr.cand] += 1;
            bool done = true;
            for (int sdone : e.samples) done = done && sdone >= kTuneSamples;
            if (done) {
                e.fixed = e.cand[std::min_element(e.best_ms.begin(), e.best_ms.end()) - e.best_ms.begin()];
                if (std::getenv("FLUID_TUNE_LOG")) {
                    std::string msg;
                    for (size_t q = 0; q < e.cand.size(); ++q) msg += " " + std::to_string(e.cand[q]) + ":" + std::to_string((int)(e.best_ms[q] * 1e3f));
                    fprintf(stderr, "[fluid tune] key %016llx -> %d rows  (height:us%s)\n", tr.key, e.fixed, msg.c_str());
                }
            }
        }
        c->free_events.push_back(tr.a);
        c->free_events.push_back(tr.b);
    }
    c->trials.resize(keep);
}

// the height to use for this launch; *trial >= 0: it is a measurement of candidate *trial (tune_begin / tune_end bracket the launch)
int tune_pick(fluid_ctx* c, unsigned long long key, int heuristic, int T, long long rows_n, int* trial)
{
    *trial = -1;
    tune_harvest(c);
    RbTuner& t = rb_tuner();
    std::lock_guard<std::mutex> lock(t.mu);
    RbTuner::Entry& e = t.table[key];
    if (e.fixed) return e.fixed;
    if (e.cand.empty()) {
        e.cand.push_back(heuristic);
        for (int r : {48, 56, 64, 80, 96, 112, 128, 160, 192}) {
            if (r < 2 * T || r >= rows_n + 2 * T) continue;     // (a height past the rows is one strip: the tallest candidate covers it)
            if (std::find(e.cand.begin(), e.cand.end(), r) == e.cand.end()) e.cand.push_back(r);
        }
        // small grids are bound by the latency of one wave's march, rb + 2T steps: strips shorter than the pipeline is
        // deep pay there (256^2, 8 sweeps per launch: 0.21 ms per step at 4 rows against 0.26 at 16)
        if (rows_n <= 2200)
            for (int r : {2, 4, 8, 12, 24, 32, 40})
                if (r < rows_n && std::find(e.cand.begin(), e.cand.end(), r) == e.cand.end()) e.cand.push_back(r);
        if (e.cand.size() == 1) {                                // nothing to choose from (tiny grids)
            e.fixed = heuristic;
            return heuristic;
        }
        e.best_ms.assign(e.cand.size(), 0.f);
        e.samples.assign(e.cand.size(), 0);
        e.issued.assign(e.cand.size(), 0);
    }
    // the candidate with the fewest trials issued so far (finished or still in flight)
    const int pick = (int)(std::min_element(e.issued.begin(), e.issued.end()) - e.issued.begin());
    if (e.issued[pick] >= kTuneSamples + 2) return e.cand[0];   // all issued, results still in flight: the closed-form pick meanwhile
    e.issued[pick] += 1;
    *trial = pick;
    return e.cand[pick];
}

int tune_begin(fluid_ctx* c, unsigned long long key, int trial)
{
    fluid_ctx::Trial tr{};
    tr.key = key;
    tr.cand = trial;
    for (hipEvent_t* ev : {&tr.a, &tr.b}) {
        if (!c->free_events.empty()) {
            *ev = c->free_events.back();
            c->free_events.pop_back();
        } else {
            HIP_TRY(hipEventCreate(ev));
        }
    }
    HIP_TRY(hipEventRecord(tr.a, c->stream));
    c->trials.push_back(tr);
    return FLUID_OK;
}

int tune_end(fluid_ctx* c)
{
    HIP_TRY(hipEventRecord(c->trials.back().b, c->stream));
    return FLUID_OK;
}

// Sweeps fused into the next launch of a solve with `remaining` sweeps to go, of which `room` can run before rows must be
// exchanged (slabs; = remaining on one GPU).  Depths 16 / 12 / 8 / 4 / 2 exist (2-column lanes, fp32 storage for 16 and
// 12).  Per sweep the deep launches are the cheap ones where they pay at all, and a shallow remainder is dear (measured,
// us per sweep of a pressure solve at 4096^2: 4.26 at 16, 4.05 at 12, 5.02 at 8; at 8192^2 13.6 / 16.3 / 23.5), so the
// depths of a solve are PLANNED: the multiset of allowed depths that adds up to `remaining` at the least estimated cost,
// deepest first -- 40 sweeps run as 16 + 12 + 12 rather than 16 + 16 + 8.
//   - 12 pays on grids (or slabs) of 8 M cells and more.  16 too, but for the general form (a double-precision multiply
//     per cell: bound by arithmetic, which deeper blocking only adds to) only once a field outgrows the Infinity Cache
//     (96 MiB rule), and it wants rows: on a 1024-row slab 12 beats it (3.3 against 3.8 us per sweep).
//   - FLUID_PARAM_TB_T16_MIN_CELLS replaces the size rules by one floor (0: always), so that tests can run the deep
//     kernels of either form on grids the oracle finishes in milliseconds.
//   - fp16 storage rounds once per launch, so its schedule is part of the result and stays the greedy 8 / 4 / 2 one.
int pick_sweeps(const fluid_ctx* c, int remaining, int room, bool canonical, bool small, long long slab_cells, bool all_mode4)
{
    // the fused kernel addresses a field through 32-bit buffer offsets: fields of 2 GiB and more
    // (beyond ~23000^2 in fp32) take single-sweep launches
    if (c->variant != fluid::JACOBI_TB || small || c->field_bytes >= 0x7F000000ull) return 1;
    room = std::min(room, remaining);
    const int greedy = (room >= 8 && c->tb_max_t >= 8) ? 8 : (room >= 4 && c->tb_max_t >= 4) ? 4 : room >= 2 ? 2 : 1;
    if (canonical || c->tb_nv != 2 || c->tb_max_t < 12 || room < 12 || (remaining & 1)) return greedy;
    const bool forced = c->tb_t16_min_cells >= 0;
    const bool big = forced ? slab_cells >= c->tb_t16_min_cells : slab_cells >= (8ll << 20);
    if (!big) return greedy;
    const long long slab_rows = slab_cells / std::max(c->n, 1);
    // 16: the pressure form always; the general form once a field outgrows the Infinity Cache; never on short slabs
    const bool pays16 = forced || ((all_mode4 || (unsigned long long)slab_cells * c->esz > (96ull << 20)) && slab_rows >= 3000);
    static const int depth[5] = {16, 12, 8, 4, 2};
    static const double per_sweep[5] = {1.00, 1.04, 1.30, 2.6, 5.0};      // relative cost of one sweep at that depth
    const double per_launch = 0.5;
    bool allowed[5] = {c->tb_max_t >= 16 && pays16, true, c->tb_max_t >= 8, c->tb_max_t >= 4, true};
    // least cost to run exactly r sweeps (r even)
    std::vector<double> cost(remaining + 1, 1e300);
    cost[0] = 0.0;
    for (int r = 2; r <= remaining; r += 2)
        for (int k = 0; k < 5; ++k) {
            if (!allowed[k] || depth[k] > r) continue;
            const double v = cost[r - depth[k]] + depth[k] * per_sweep[k] + per_launch;
            if (v < cost[r] - 1e-9) cost[r] = v;
        }
    // the plan's launches, deepest first; take the deepest one that fits the room
    int best = 0;
    for (int r = remaining; r > 0;) {
        int pickd = 0;
        for (int k = 0; k < 5 && !pickd; ++k)
            if (allowed[k] && depth[k] <= r && std::fabs(cost[r - depth[k]] + depth[k] * per_sweep[k] + per_launch - cost[r]) < 1e-9)
                pickd = depth[k];
        if (!pickd) break;
        if (pickd <= room) best = std::max(best, pickd);
        r -= pickd;
    }
    return best ? best : greedy;
}

// One Jacobi solve of the step: field x (first guess in, result out), right-hand
// side x0, wall rule b.  Up to three such solves of the same length run as one
// batch (u, v and density diffusion are independent of one another).
struct Solve {
    int b, x, x0;
    float alpha, beta;
};

// FluidSequential.c:85-104.  Results land in the fields `x`.  A sweep that writes
// `r` rows past the slab needs x valid r+1 rows out and x0 r rows out, so a solve
// that starts with reach R runs R sweeps before it must exchange again -- on
// ranges that shrink one row per sweep per inner edge, the same arithmetic per
// cell as the 1-GPU run (bit-identical).  `final_reach`: rows past the slab the
// caller would like valid afterwards (the gradient wants 1).  The temporally
// blocked kernel runs T of the sweeps per launch, all solves of the batch in the
// same launch.  Sweeps ping-pong between x's buffer and a scratch field's; if a
// result ends in the scratch buffer the two fields trade buffers (pointer swap,
// no copy) -- field ids, not addresses, are stable.
// `ds` (one GPU, a single pressure solve): the right-hand side sv[0].x0 is the divergence of (ds->u, ds->v), not yet
// computed -- the first launch computes it row by row as it goes and stores it (fluid_kernels.hip, DIVSRC).
struct DivSource {
    int u, v;
    float scale;          // -0.5f * h
};

// `keep_pending`: an exchange the caller issued async feeds another batch on another stream too -- this batch joins it on
// its own stream but leaves it marked as in flight (full_step: the density diffusion beside the velocity path)
int op_diffuse_batch(fluid_ctx* c, const Solve* sv, int count, int iters, int final_reach = 0, const DivSource* ds = nullptr,
                     int scratch_base = 0, bool keep_pending = false)
{
    static const int kScratchAll[3] = {FLUID_TMP0, FLUID_TMP1, FLUID_TMP2};
    static const int kSumAll[3] = {FLUID_TMP3, FLUID_TMP4, FLUID_TMP5};      // x0 + dt*s of a deferred add_source lands here
    if (scratch_base < 0 || scratch_base + count > 3) return fail(FLUID_E_INVALID, "a batch holds 1 to 3 solves");
    const int* kScratch = kScratchAll + scratch_base;      // (a solve that runs beside another batch on a second stream takes the last one)
    const int* kSum = kSumAll + scratch_base;
    if (iters < 0 || (iters & 1)) return fail(FLUID_E_INVALID, "sweep count must be even and >= 0 (got %d)", iters);
    if (count < 1 || count > 3) return fail(FLUID_E_INVALID, "a batch holds 1 to 3 solves");
    for (int k = 0; k < count; ++k) {
        if (sv[k].x == sv[k].x0 || sv[k].x >= FLUID_TMP0 || sv[k].x0 >= FLUID_TMP0)
            return fail(FLUID_E_INVALID, "diffuse: x and x0 must be distinct non-scratch fields");
        for (int j = 0; j < k; ++j)
            if (sv[j].x == sv[k].x || sv[j].x == sv[k].x0 || sv[j].x0 == sv[k].x)
                return fail(FLUID_E_INVALID, "diffuse: the solves of a batch must not share fields");
    }
    // The solve is linear in (x, x0): a right-hand side kept scaled (fp16 storage: the divergence, project()) gives a
    // solution with the same factor, provided the first guess carries it too (a guess that is zero by definition does)
    float out_scale[3] = {1.0f, 1.0f, 1.0f};
    for (int k = 0; k < count; ++k) {
        const float sx = c->zero[sv[k].x] ? c->fscale[sv[k].x0] : c->fscale[sv[k].x];
        if (sx != c->fscale[sv[k].x0] || c->src_of[sv[k].x0]) {
            if (c->fscale[sv[k].x] != 1.0f || c->fscale[sv[k].x0] != 1.0f) TRY(xchg_join(c, keep_pending));
            TRY(unscale(c, sv[k].x));
            TRY(unscale(c, sv[k].x0));
        }
        out_scale[k] = c->fscale[sv[k].x0];
    }
    if (iters == 0) {
        TRY(xchg_join(c, keep_pending));
        for (int k = 0; k < count; ++k) TRY(settle_source(c, sv[k].x0));    // (no launch to take a deferred source over)
        return FLUID_OK;
    }
    for (int k = 0; k < count; ++k) TRY(materialize_zero(c, sv[k].x0));     // a pending increment rides along (TbBatch::x0_inc)
    // a deferred add_source (op_add_source) rides in the first launch if that is a fused one of a shape that exists with the
    // second store, the source is this solve's first guess and all solves of the launch agree; else it is settled now
    bool add_src = c->src_of[sv[0].x0] != 0;
    for (int k = 0; k < count; ++k)
        add_src = add_src && c->src_of[sv[k].x0] == 1 + sv[k].x && c->src_dt[sv[k].x0] == c->src_dt[sv[0].x0] && !c->zero[sv[k].x] &&
                  !c->pend[sv[k].x] && !c->src_of[sv[k].x] && ds == nullptr;
    hipEvent_t stop;
    TRY(timing_begin(c, FLUID_TIME_DIFFUSION, &stop));
    const bool multi = c->nranks > 1;
    int cur[3], nxt[3], divmode[3];
    DivPlan plan[3];
    bool same_mode = true, all_mode4 = true;
    for (int k = 0; k < count; ++k) {
        cur[k] = sv[k].x;
        nxt[k] = kScratch[k];
        plan[k].arg = sv[k].beta;
        if (c->variant == fluid::JACOBI_TB) plan[k] = division_mode(c, sv[k].beta, sv[k].alpha);
        divmode[k] = plan[k].mode;
        same_mode = same_mode && divmode[k] == divmode[0];
        all_mode4 = all_mode4 && divmode[k] == 4;
    }
    {
        const bool canonical0 = c->st == fluid::STORAGE_F16;
        const long long cells0 = (long long)(c->nranks > 1 ? c->min_slab : c->n) * c->n;
        const bool small0 = (canonical0 ? (long long)c->n * c->n : cells0 * count) < c->tb_min_cells;
        // the depth of the first launch: what the loop below will pick when nothing is short (a short reach there exchanges
        // first, or shortens the launch -- in which case the source is settled there, before that launch)
        const int T0 = pick_sweeps(c, iters, iters, canonical0, small0, cells0, all_mode4);
        add_src = add_src && same_mode && fluid::jacobi_tb_addsrc_exists(T0, divmode[0], c->tb_nv);
        if (!add_src)
            for (int k = 0; k < count; ++k)
                if (c->src_of[sv[k].x0]) {
                    TRY(xchg_join(c, keep_pending));                        // (its kernel touches rows that may be on their way)
                    TRY(settle_source(c, sv[k].x0));
                }
    }
    // division mode 3 needs |x0| >= beta * 2^-72 wherever it is used: minima of |x0| per tile, once per solve (x0 does not
    // change during it), over the rows of x0 that are valid here; tiles beyond them read 0 = "divide the long way"
    const size_t tile_words = (size_t)fluid::tile_rows(c->n) * fluid::tile_pitch(c->n);
    {
        fluid::TileBatch tb{};
        int m = 0, valid = kEverywhere;
        for (int k = 0; k < count; ++k)
            if (divmode[k] == 3) {
                tb.field[m] = c->f[sv[k].x0];
                tb.tiles[m] = c->tiles + (size_t)(scratch_base + k) * tile_words;      // a solve's tiles go with its scratch field: a
                                                                                        // batch on the second stream has slots of its own
                valid = std::min(valid, c->nranks > 1 ? c->reach[sv[k].x0] : kEverywhere);
                ++m;
            }
        if (m > 0) {
            TRY(xchg_join(c, keep_pending));
            if (c->nranks > 1)
                HIP_TRY(hipMemsetAsync(c->tiles + (size_t)scratch_base * tile_words, 0, (size_t)count * tile_words * sizeof(unsigned), c->stream));
            int lo, hi;
            rows(c, std::min(valid, c->n), &lo, &hi);
            fluid::launch_tile_min_abs(c->stream, c->st, tb, m, c->pitch, c->n, lo, hi, fluid::tile_pitch(c->n));
        }
    }
    auto reach_now = [&]() {
        int r = kEverywhere;
        for (int k = 0; k < count; ++k) r = std::min(r, std::min(c->reach[cur[k]], c->reach[sv[k].x0] + 1));
        return r;
    };
    int r = multi ? reach_now() : kEverywhere;            // sweeps possible right now
    // FLUID_PARAM_TB_MIN_CELLS: slabs below it run one launch per sweep.  Default 0: with 2-column
    // lanes the fused kernel wins at every size measured (32^2 .. 16384^2) -- tiny grids are bound by
    // launch latency and it needs 5 launches per solve instead of 40.
    // fp16 storage rounds once per launch, so there the launch schedule is part of the result: it
    // must not depend on how the grid is split, how deep the ghost zones are or how solves are
    // batched.  It is derived from the global problem alone and a short reach triggers an early
    // exchange instead of a shorter launch.  (fp32 results do not depend on the schedule.)
    const bool canonical = c->st == fluid::STORAGE_F16;
    // what a rank sweeps (the whole grid on one GPU).  From the base slab height, not this rank's own (ranks differ by a
    // row when N does not divide evenly), so that every rank takes the same size-dependent decisions.  (Exchanges are
    // driven by `reach`, which every sweep consumes alike whatever the launch depth, so they would pair up regardless.)
    const long long slab_cells = (long long)(c->nranks > 1 ? c->min_slab : c->n) * c->n;
    const bool small = (canonical ? (long long)c->n * c->n : slab_cells * count) < c->tb_min_cells;
    bool joined = false;                  // this batch's stream has waited for the exchange in flight (if any)
    for (int k = 0; k < iters;) {
        const int remaining = iters - k;
        auto pick = [&](int room) { return pick_sweeps(c, remaining, room, canonical, small, slab_cells, all_mode4); };
        const int wantT = canonical ? pick(remaining) : 1;     // slabs with fp16 storage keep halo >= 8 (fluid_create_ex)
        if (r < wantT) {
            const int depth = std::max(wantT, std::min(c->halo, remaining + final_reach));
            std::vector<int> ids;
            for (int j = 0; j < count; ++j) {
                ids.push_back(cur[j]);
                if (c->reach[sv[j].x0] < depth - 1) ids.push_back(sv[j].x0);
            }
            TRY(xchg_join(c, keep_pending));                                  // (the caller's exchange first, if it is still out)
            TRY(need_list(c, ids, depth, /*async=*/true));                    // the launch below is split around it
            keep_pending = false;                                             // (this one is this batch's own)
            joined = false;
            r = reach_now();
        }
        const int T = canonical ? wantT : pick(std::min(r, remaining));
        if (add_src && k == 0 && !fluid::jacobi_tb_addsrc_exists(T, divmode[0], c->tb_nv)) {
            add_src = false;                   // a shallower first launch than planned (short reach): the kernel of its own after all
            if (!joined) TRY(xchg_join(c, keep_pending));
            joined = true;
            for (int j = 0; j < count; ++j) TRY(settle_source(c, sv[j].x0));
        }
        int lo, hi;
        rows(c, multi ? std::min(r - T, exchange_cap(c)) : 0, &lo, &hi);
        if (T == 1) {
            const int v = c->variant == fluid::JACOBI_TB ? (small ? fluid::JACOBI_NAIVE : fluid::JACOBI_STREAM) : c->variant;
            if (!joined) TRY(xchg_join(c, keep_pending));
            joined = true;
            for (int j = 0; j < count; ++j) TRY(materialize(c, cur[j], /*keep_scale=*/true));      // single-sweep kernels read x
            for (int j = 0; j < count; ++j) TRY(settle(c, sv[j].x0, /*keep_scale=*/true));         // ... and x0 as it is in memory
            for (int j = 0; j < count; ++j)
                fluid::launch_jacobi(c->stream, c->st, v, c->f[cur[j]], c->f[sv[j].x0], c->f[nxt[j]], c->pitch, c->n, lo, hi,
                                     sv[j].alpha, sv[j].beta, sv[j].b);
            if (c->timing) {
                c->launches += count;
                c->field_launches += count;
            }
        } else {
            // one launch per group of solves that share a division mode (normally: all of them)
            for (int first = 0; first < count;) {
                fluid::TbBatch bt{};
                int m = 0;
                int last = first;
                for (int j = first; j < count && (same_mode || j == first); ++j, ++last) {
                    bt.x[m] = c->f[cur[j]];
                    bt.x0[m] = c->f[sv[j].x0];
                    bt.out[m] = c->f[nxt[j]];
                    bt.alpha[m] = sv[j].alpha;
                    bt.beta[m] = plan[j].arg;
                    bt.yd[m] = plan[j].yd;
                    bt.hi[m] = plan[j].hi;
                    bt.lo[m] = plan[j].lo;
                    bt.tiles[m] = divmode[j] == 3 ? c->tiles + (size_t)(scratch_base + j) * tile_words : nullptr;
                    bt.tile_thr[m] = plan[j].tile_thr;
                    bt.b[m] = sv[j].b;
                    bt.x_zero[m] = c->zero[cur[j]] ? 1 : 0;
                    bt.x0_inc[m] = c->pend[sv[j].x0] ? c->pend_inc[sv[j].x0] : -0.0f;     // x + (-0) is x for every x
                    ++m;
                }
                bt.count = m;
                bt.tile_pitch = fluid::tile_pitch(c->n);
                const bool divsrc = ds != nullptr && k == 0;
                const bool addsrc = add_src && k == 0;
                if (addsrc) {
                    for (int j = first, q = 0; j < last; ++j, ++q) {
                        bt.div[q] = c->f[kSum[j]];
                        bt.x0_inc[q] = -0.0f;
                    }
                    bt.div_scale = c->src_dt[sv[first].x0];
                }
                if (divsrc) {
                    bt.x[0] = c->f[ds->u];
                    bt.x0[0] = c->f[ds->v];
                    bt.div[0] = c->f[sv[0].x0];
                    bt.div_scale = ds->scale;
                    bt.x0_inc[0] = -0.0f;
                }
                const int edge_pct = c->tb_edge_pct > 0 ? c->tb_edge_pct : 100;
                auto edge_rows = [&](int r) { return std::max(2 * T, r * edge_pct / 100); };
                // this launch on output rows [lo_, hi_) (the whole launch, or one part of a launch split around an exchange)
                auto launch_rows = [&](int lo_, int hi_, int hole_lo = 0, int hole_hi = 0) -> int {
                    int rb = c->tb_rows;
                    if (rb <= 0) {
                        // auto (tools/tb_sweep.py on MI355X): the kernel hides its latencies only behind other
                        // waves, so it wants every block resident at once -- a 256-thread block is one wave per
                        // SIMD, tb_waves_per_simd blocks fit a CU -- and then the tallest strips that still
                        // allow (each strip pays 2T rows of pipeline fill).  Smallest strip height whose
                        // non-empty blocks fit 92 % of one round; grids too large for one round stop at 80
                        // rows (160 for a batch; 192 at T = 16), past which more strips win again.
                        const int nv = c->tb_nv;
                        const int HL = (T + nv - 1) / nv, VS = 64 - 2 * HL;
                        const long long windows = ((c->n + nv - 1) / nv + VS - 1) / VS;
                        const int resident = nv == 2 ? (T >= 16 ? 2 : 4) : (T >= 8 ? 2 : 3);
                        const long long room = (long long)c->num_cu * resident * 92 / 100;
                        const long long rows_n = (hi_ - lo_) - std::max(0, hole_hi - hole_lo);
                        auto blocks = [&](int r) {
                            const long long si = (rows_n + r - 1) / r, se = (rows_n + edge_rows(r) - 1) / edge_rows(r);
                            const long long inner = windows > 2 ? windows - 2 : 0, outer = windows - inner;
                            return (inner * ((si + 3) / 4) + outer * ((se + 3) / 4)) * m;
                        };
                        const int cap = T >= 16 ? 192 : (T >= 8 ? 80 : 96) * (m > 1 ? 2 : 1);
                        rb = 2 * T;
                        while (rb < cap && blocks(rb) > room) rb += 2;
                        // small grids: a launch lasts as long as one wave's march of rb + 2T rows, and the best height
                        // measured is about rows / 64 (2 at 128^2, 4 at 256^2, 8 at 512^2, 16 and more from 1024^2)
                        if (rows_n <= 1100) rb = std::max(2, std::min(rb, (int)(rows_n / 64) & ~1));
                    }
                    // ... and that closed form is only the first candidate of the run-time tuner (see RbTuner)
                    int trial = -1;
                    unsigned long long key = 0;
                    if (c->tb_rows <= 0 && c->autotune) {
                        // (the two edge parts of a split launch: keyed by their rows, and as a shape of their own)
                        const long long rows_k = (hi_ - lo_) - std::max(0, hole_hi - hole_lo);
                        key = tune_key(c, T, m + (divsrc ? 8 : 0) + (addsrc ? 16 : 0) + (hole_hi > hole_lo ? 32 : 0), divmode[first], rows_k);
                        rb = tune_pick(c, key, rb, T, rows_k, &trial);
                    }
                    // edge windows (ghost columns) cost ~1.6x per row: shorter strips there keep the launch balanced
                    const int rb_edge = std::min(rb, edge_rows(rb));
                    if (trial >= 0) TRY(tune_begin(c, key, trial));
                    fluid::launch_jacobi_tb(c->stream, c->st, T, divmode[first], c->tb_nv, bt, c->pitch, c->n, lo_, hi_, rb, rb_edge, divsrc, addsrc,
                                            hole_lo, hole_hi);
                    if (trial >= 0) TRY(tune_end(c));
                    return FLUID_OK;
                };
                // An exchange is still in flight (issued async just above, or by the caller): the strips whose inputs are this
                // slab's own rows -- output rows [own0 + T, own1 - T): T sweeps reach T rows -- go first and run while the halo
                // rows travel; the strips next to the slab's inner edges wait for the exchange's event.  Same arithmetic per
                // cell whichever launch it falls into.
                int in_lo = lo, in_hi = hi;
                if (c->xpend && !joined && multi) {
                    if (c->rank > 0) in_lo = std::max(lo, c->own0 + T);
                    if (c->rank < c->nranks - 1) in_hi = std::min(hi, c->own1 - T);
                }
                if (c->xpend && !joined && in_hi - in_lo >= 2 * T && (in_lo > lo || in_hi < hi)) {
                    TRY(launch_rows(in_lo, in_hi));
                    TRY(xchg_join(c, keep_pending));
                    TRY(launch_rows(lo, hi, in_lo, in_hi));              // both edge parts in one launch
                    c->split_launches += 1;
                } else {
                    if (!joined) TRY(xchg_join(c, keep_pending));
                    TRY(launch_rows(lo, hi));
                }
                joined = true;
                if (c->timing) {
                    c->launches += 1;
                    c->field_launches += m;
                }
                first = last;
            }
        }
        r = multi ? std::min(r - T, exchange_cap(c)) : kEverywhere;
        if (ds && k == 0 && multi) c->reach[sv[0].x0] = std::max(0, std::min(c->reach[sv[0].x0], r));   // the divergence exists where this launch stored it
        if (add_src && k == 0) {
            // the sums were stored out of place, on the rows this launch computed (+ the wall rows next to them): the
            // right-hand side takes that buffer, the scratch field the old one
            for (int j = 0; j < count; ++j) {
                const int x0 = sv[j].x0;
                std::swap(c->f[x0], c->f[kSum[j]]);
                c->zero[kSum[j]] = false;
                wrote(c, kSum[j], 0);
                wrote(c, x0, std::max(0, r));
            }
            add_src = false;
        }
        for (int j = 0; j < count; ++j) {
            c->zero[cur[j]] = false;       // from now on this buffer is just the other half of the ping-pong
            c->zero[nxt[j]] = false;
            wrote(c, nxt[j], r);
            std::swap(cur[j], nxt[j]);
        }
        k += T;
    }
    HIP_TRY(hipGetLastError());
    for (int j = 0; j < count; ++j) {
        if (cur[j] != sv[j].x) {
            std::swap(c->f[sv[j].x], c->f[kScratch[j]]);
            std::swap(c->reach[sv[j].x], c->reach[kScratch[j]]);
            std::swap(c->pend[sv[j].x], c->pend[kScratch[j]]);
            std::swap(c->pend_inc[sv[j].x], c->pend_inc[kScratch[j]]);
        }
        wrote(c, kScratch[j], 0);
        c->fscale[sv[j].x] = out_scale[j];
    }
    return timing_end(c, stop, iters * count);
}

int op_diffuse(fluid_ctx* c, int b, int x, int x0, float alpha, float beta, int iters, int final_reach = 0, const DivSource* ds = nullptr)
{
    const Solve one{b, x, x0, alpha, beta};
    return op_diffuse_batch(c, &one, 1, iters, final_reach, ds);
}

// FluidSequential.c:107-141.  The back-trace reaches dt0*max|vel| cells, so a
// slab first learns the global bound (wavefront reduction + MAX exchange) and
// makes sure that many rows of the advected field(s) are valid past its edges;
// when the reach exceeds what a neighbour can supply it gathers whole fields.
// The bound is a host decision, i.e. a pipeline drain: vmax_begin() only
// enqueues (reduction kernel, device-side all-reduce, copy to pinned memory),
// so the caller can put independent work behind it before advect_halo() waits.
// `have_max`: the gradient subtraction that has just produced (u, v) left their maximum in the device word already
// (op_subtract_gradient with_max)
int vmax_begin(fluid_ctx* c, int u, int v, bool have_max = false)
{
    if (c->nranks == 1) return FLUID_OK;
    if (!have_max) {
        TRY(materialize(c, {u, v}));
        HIP_TRY(hipMemsetAsync(c->d_scalar, 0, sizeof(unsigned), c->stream));
        fluid::launch_absmax2(c->stream, c->st, c->f[u], c->f[v], c->pitch, c->n, c->own0, c->own1, c->d_scalar);
    }
    TRY(exchange(c, FLUID_XCHG_MAX_BEGIN, {}, 0));       // in-place MAX over ranks on the device scalar
    HIP_TRY(hipMemcpyAsync(c->h_scalar, c->d_scalar, sizeof(unsigned), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipEventRecord(c->scalar_ready, c->stream));
    return FLUID_OK;
}

int advect_halo(fluid_ctx* c, std::initializer_list<int> sources, float dt0)
{
    if (c->nranks == 1) return FLUID_OK;
    HIP_TRY(hipEventSynchronize(c->scalar_ready));
    float vmax = 0.f;
    std::memcpy(&vmax, c->h_scalar, sizeof vmax);
    TRY(exchange(c, FLUID_XCHG_MAX_END, {}, 0, &vmax));  // transports that reduce on the host finish here
    const double reach = std::ceil((double)std::fabs(dt0) * (double)vmax) + 2.0;
    if (!(reach <= (double)exchange_cap(c))) {     // also catches NaN/inf
        TRY(exchange(c, FLUID_XCHG_GATHER, sources, 0));
        for (int f : sources) c->reach[f] = kEverywhere;
        return FLUID_OK;
    }
    return need(c, sources, (int)reach);
}

// vmax_begin() has been enqueued; `run` enqueues the advection of `sources` (FluidSequential.c:107-141) along that velocity.
// The classic order -- wait for the bound, bring in that many rows, advect -- leaves the GPU idle while the host reads the
// bound and enqueues the rest.  Velocities change little from step to step, so with a bound from the previous step at
// hand (slot 0: the velocity's own advection, 1: the density's) the exchange and the advection are enqueued FIRST, on
// that bound plus a quarter, and the host then checks the new bound against it: the same on every rank (both are
// all-reduced values), so all ranks agree on whether to do it over.  Doing it over is safe: an advection writes fields
// it does not read, and nothing that overwrites its inputs is enqueued before this function returns.
int advect_bounded(fluid_ctx* c, int slot, std::initializer_list<int> sources, float dt0, const std::function<int()>& run)
{
    if (c->nranks == 1) return run();
    int guess = 0;
    const float prev = c->vmax_prev[slot];
    if (c->early_advect && prev >= 0.0f) {
        const double g = std::ceil((double)std::fabs(dt0) * (double)prev * 1.25) + 2.0;
        if (g <= (double)exchange_cap(c)) {
            guess = (int)g;
            TRY(need(c, sources, guess));
            TRY(run());
        }
    }
    HIP_TRY(hipEventSynchronize(c->scalar_ready));
    float vmax = 0.f;
    std::memcpy(&vmax, c->h_scalar, sizeof vmax);
    TRY(exchange(c, FLUID_XCHG_MAX_END, {}, 0, &vmax));  // transports that reduce on the host finish here
    const double reach = std::ceil((double)std::fabs(dt0) * (double)vmax) + 2.0;
    c->vmax_prev[slot] = std::isfinite(vmax) ? vmax : -1.0f;
    if (guess > 0 && reach <= (double)guess) return FLUID_OK;
    if (!(reach <= (double)exchange_cap(c))) {     // also catches NaN/inf
        TRY(exchange(c, FLUID_XCHG_GATHER, sources, 0));
        for (int f : sources) c->reach[f] = kEverywhere;
    } else {
        TRY(need(c, sources, (int)reach));
    }
    return run();
}

int advect_prepare(fluid_ctx* c, std::initializer_list<int> sources, int u, int v, float dt0)
{
    TRY(vmax_begin(c, u, v));
    return advect_halo(c, sources, dt0);
}

int op_advect(fluid_ctx* c, int b, int d, int d0, int u, int v, float dt)
{
    if (d == d0 || d == u || d == v) return fail(FLUID_E_INVALID, "advect: output must not alias an input");
    const float dt0 = dt * (float)c->n;
    TRY(materialize(c, {d0, u, v}));
    c->zero[d] = false;
    TIMED(c, FLUID_TIME_ADVECTION,
          fluid::launch_advect(c->stream, c->st, c->f[d], c->f[d0], c->f[u], c->f[v], c->pitch, c->n, c->own0, c->own1, dt0, b));
    wrote(c, d, 0);
    return FLUID_OK;
}

// two advections along the same velocity (u, v) in one launch; results identical to two op_advect calls
int op_advect2(fluid_ctx* c, int ba, int da, int d0a, int bb, int db, int d0b, int u, int v, float dt)
{
    for (int d : {da, db})
        if (d == d0a || d == d0b || d == u || d == v) return fail(FLUID_E_INVALID, "advect: output must not alias an input");
    if (da == db) return fail(FLUID_E_INVALID, "advect: outputs must be distinct");
    const float dt0 = dt * (float)c->n;
    TRY(materialize(c, {d0a, d0b, u, v}));
    c->zero[da] = false;
    c->zero[db] = false;
    TIMED(c, FLUID_TIME_ADVECTION,
          fluid::launch_advect2(c->stream, c->st, c->f[da], c->f[d0a], ba, c->f[db], c->f[d0b], bb, c->f[u], c->f[v], c->pitch,
                                c->n, c->own0, c->own1, dt0));
    wrote(c, da, 0);
    wrote(c, db, 0);
    return FLUID_OK;
}

// FluidSequential.c:143-158.  `want`: rows past the slab on which the caller
// would like the divergence (so that the pressure solve that follows needs no
// exchange of its own); u and v are brought in one row further than that.
// `pscale`: the divergence is stored multiplied by this power of two (project() with fp16 storage; 1 from the operator API)
int op_divergence(fluid_ctx* c, int u, int v, int p, int div, int want = 0, float pscale = 1.0f)
{
    if (p == u || p == v || div == u || div == v || p == div)
        return fail(FLUID_E_INVALID, "divergence: outputs must not alias inputs");
    const float h = 1.0f / (float)c->n;
    int reach = 0;
    TRY(materialize(c, {u, v}));
    if (c->nranks > 1) {
        reach = std::max(0, std::min(want, exchange_cap(c) - 1));
        TRY(need(c, {u, v}, reach + 1));
    }
    int lo, hi;
    rows(c, reach, &lo, &hi);
    // p = 0 everywhere (FluidSequential.c:153 + set_bnd(0,p)): marked, not written
    TIMED(c, FLUID_TIME_DIVERGENCE,
          fluid::launch_divergence(c->stream, c->st, c->f[u], c->f[v], c->f[p], c->f[div], c->pitch, c->n, lo, hi, h,
                                   /*write_p=*/0, pscale));
    c->zero[div] = false;
    wrote(c, div, reach);
    c->fscale[div] = pscale;
    mark_zero(c, p);
    return FLUID_OK;
}

// `with_max` (slabs): max(|u|, |v|) of the result, over the slab's own rows, is left in the device word for the
// vmax_begin(.., have_max) that follows -- the bound of the advection along (u, v), without a pass of its own
int op_subtract_gradient(fluid_ctx* c, int u, int v, int p, bool with_max = false)
{
    if (p == u || p == v || u == v) return fail(FLUID_E_INVALID, "subtract_gradient: fields must be distinct");
    const float h = 1.0f / (float)c->n;
    TRY(materialize(c, {u, v}));
    TRY(materialize(c, p, /*keep_scale=*/true));           // a scaled pressure is divided back inside the kernel, exactly
    TRY(need(c, {p}, 1));
    with_max = with_max && c->nranks > 1 && c->d_partials;
    TIMED(c, FLUID_TIME_PROJECTION,
          fluid::launch_subtract_gradient(c->stream, c->st, c->f[u], c->f[v], c->f[p], c->pitch, c->n, c->own0, c->own1, h,
                                          c->d_partials, with_max ? c->d_scalar : nullptr, 1.0f / c->fscale[p]));
    wrote(c, u, 0);
    wrote(c, v, 0);
    return FLUID_OK;
}

// the gradient subtraction of a projection and the advection of `d` (from d0, wall rule b) along the projected velocity,
// in one launch: results identical to op_subtract_gradient followed by op_advect.  One GPU only (on slabs the advection
// waits for a reduction over the velocity it follows).
int op_gradient_advect(fluid_ctx* c, int u, int v, int p, int b, int d, int d0, float dt)
{
    if (p == u || p == v || u == v) return fail(FLUID_E_INVALID, "subtract_gradient: fields must be distinct");
    if (d == d0 || d == u || d == v || d == p || d0 == u || d0 == v)
        return fail(FLUID_E_INVALID, "advect: output must not alias an input");
    if (c->nranks != 1) return fail(FLUID_E_INVALID, "op_gradient_advect is a one-GPU operator");
    const float h = 1.0f / (float)c->n;
    TRY(materialize(c, {u, v, d0}));
    TRY(materialize(c, p, /*keep_scale=*/true));
    c->zero[d] = false;
    TIMED(c, FLUID_TIME_PROJECTION,
          fluid::launch_gradient_advect(c->stream, c->st, c->f[u], c->f[v], c->f[p], c->f[d], c->f[d0], c->pitch, c->n, c->own0,
                                        c->own1, h, dt * (float)c->n, b, 1.0f / c->fscale[p]));
    wrote(c, u, 0);
    wrote(c, v, 0);
    wrote(c, d, 0);
    return FLUID_OK;
}

void coefficients(int n, float dt, float coef, float* alpha, float* beta)
{
    // ((dt*coef)*n)*n in float, then 1 + 4*alpha (FluidSequential.c:179-180,199-200)
    volatile float a = dt * coef;
    a = a * (float)n;
    a = a * (float)n;
    volatile float four_a = 4.0f * a;
    *alpha = a;
    *beta = 1.0f + four_a;
}

// divergence -> pressure solve -> gradient subtraction (FluidSequential.c:213-223
// and :238-240).  On slabs: ONE exchange (u, v, iters+2 rows) covers the divergence,
// every sweep of the solve and the gradient's one-row halo of p.
// `then_advect` (one GPU): d, d0, b, dt of an advection along (u, v) to run in the same launch as the gradient subtraction
struct AdvectAfter {
    int b, d, d0;
    float dt;
};

// can the divergence be computed inside the first launch of the pressure solve that follows it?  One GPU, the fused
// kernel with 2-column lanes, a first launch of at least 8 sweeps, the exact-reciprocal division of (alpha 1, beta 4)
bool divergence_fuses(fluid_ctx* c, int iters, int reach)
{
    if (c->variant != fluid::JACOBI_TB || c->tb_nv != 2 || !c->fuse_divergence || iters < 8) return false;
    const bool canonical = c->st == fluid::STORAGE_F16;
    const long long slab_cells = (long long)(c->nranks > 1 ? c->min_slab : c->n) * c->n;
    if ((canonical ? (long long)c->n * c->n : slab_cells) < c->tb_min_cells) return false;
    const int room = c->nranks > 1 ? std::min(reach + 1, iters) : iters;        // sweeps the first launch may fuse (op_diffuse_batch)
    if (room < 8 || pick_sweeps(c, iters, room, canonical, false, slab_cells, true) < 8) return false;
    return division_mode(c, 4.0f, 1.0f).mode == 4;
}

// `with_max` (slabs): see op_subtract_gradient
int project(fluid_ctx* c, int u, int v, int p, int div, int iters, const AdvectAfter* then_advect = nullptr, bool with_max = false)
{
    // rows past the slab on which the divergence is wanted (so that the solve needs no exchange of its own), as op_divergence
    const int reach = c->nranks > 1 ? std::max(0, std::min(std::min(iters, c->halo - 1), exchange_cap(c) - 1)) : 0;
    if (divergence_fuses(c, iters, reach)) {
        // computeDivergenceAndPressure (FluidSequential.c:143-158) inside the solve's first launch: p = 0 is a mark, the
        // divergence is produced row by row as that launch's right-hand side and stored, ghost cells included
        if (p == u || p == v || div == u || div == v || p == div)
            return fail(FLUID_E_INVALID, "divergence: outputs must not alias inputs");
        TRY(materialize(c, {u, v}));
        if (c->nranks > 1) TRY(need(c, {u, v}, reach + 1, /*async=*/true));      // joined by the solve's first launch
        mark_zero(c, p);
        c->zero[div] = false;               // about to be overwritten entirely
        c->pend[div] = false;
        c->reach[div] = c->nranks > 1 ? reach : kEverywhere;   // what the first launch can form from (u, v); it records what it stored
        const DivSource ds{u, v, (-0.5f * (1.0f / (float)c->n)) * c->pscale};
        c->fscale[div] = c->pscale;         // (what the first launch stores; the solve is linear: p comes out with the same factor)
        c->in_pressure_solve = true;
        const int rc = op_diffuse(c, 0, p, div, 1.0f, 4.0f, iters, /*final_reach=*/1, &ds);
        c->in_pressure_solve = false;
        TRY(rc);
        TRY(xchg_join(c));
        wrote(c, div, 0);
        c->fscale[div] = c->pscale;
        if (then_advect) return op_gradient_advect(c, u, v, p, then_advect->b, then_advect->d, then_advect->d0, then_advect->dt);
        return op_subtract_gradient(c, u, v, p, with_max);
    }
    TRY(op_divergence(c, u, v, p, div, std::min(iters, c->halo - 1), c->pscale));
    c->in_pressure_solve = true;            // timing only: reported separately (fluid_timing::pressure_ms)
    const int rc_solve = op_diffuse(c, 0, p, div, 1.0f, 4.0f, iters, /*final_reach=*/1);
    c->in_pressure_solve = false;
    TRY(rc_solve);
    if (then_advect) return op_gradient_advect(c, u, v, p, then_advect->b, then_advect->d, then_advect->d0, then_advect->dt);
    return op_subtract_gradient(c, u, v, p, with_max);
}

// FluidSequential.c:189-241 with the SWAPs resolved into field roles:
// after :201/:209 the diffused velocity lives in the *_prev buffers.
int vel_step(fluid_ctx* c, float dt, float visc, int iters)
{
    const int U = FLUID_U, V = FLUID_V, U0 = FLUID_U_PREV, V0 = FLUID_V_PREV;
    float alpha, beta;
    TRY(op_add_source(c, U, U0, dt, /*defer=*/true));
    TRY(op_add_source(c, V, V0, dt, /*defer=*/true));
    coefficients(c->n, dt, visc, &alpha, &beta);
    // one exchange feeds both solves: right-hand sides iters-1 rows out, first guesses iters rows
    const int h = std::min(iters, c->halo);
    TRY(need(c, {U, V, U0, V0}, h, /*async=*/true));
    const Solve uv[2] = {{1, U0, U, alpha, beta}, {2, V0, V, alpha, beta}};
    TRY(op_diffuse_batch(c, uv, 2, iters));
    TRY(xchg_join(c));
    TRY(project(c, U0, V0, /*p=*/U, /*div=*/V, iters, nullptr, /*with_max=*/true));
    const float dt0 = dt * (float)c->n;
    TRY(vmax_begin(c, U0, V0, /*have_max=*/true));
    TRY(advect_bounded(c, 0, {U0, V0}, dt0, [&] { return op_advect2(c, 1, U, U0, 2, V, V0, U0, V0, dt); }));
    return project(c, U, V, /*p=*/U0, /*div=*/V0, iters);
}

// FluidSequential.c:176-186
int dens_step(fluid_ctx* c, float dt, float diff, int iters)
{
    const int X = FLUID_DENS, X0 = FLUID_DENS_PREV;
    float alpha, beta;
    TRY(op_add_source(c, X, X0, dt, /*defer=*/true));
    coefficients(c->n, dt, diff, &alpha, &beta);
    TRY(op_diffuse(c, 0, X0, X, alpha, beta, iters));
    TRY(vmax_begin(c, FLUID_U, FLUID_V));
    return advect_bounded(c, 1, {X0}, dt * (float)c->n, [&] { return op_advect(c, 0, X, X0, FLUID_U, FLUID_V, dt); });
}

// One loop body of the reference's main (FluidSequential.c:305-306).  The density's
// source term and diffusion are brought forward next to the velocity's: the three
// diffusions are independent of one another and of everything in between, so they
// run as ONE batch (one exchange on slabs, three times the waves per launch); the
// arithmetic per cell and the final contents of all six fields are unchanged.
int full_step(fluid_ctx* c, float dt, float diff, float visc, int iters)
{
    const int U = FLUID_U, V = FLUID_V, D = FLUID_DENS, U0 = FLUID_U_PREV, V0 = FLUID_V_PREV, D0 = FLUID_DENS_PREV;
    TRY(op_add_source(c, U, U0, dt, /*defer=*/true));
    TRY(op_add_source(c, V, V0, dt, /*defer=*/true));
    TRY(op_add_source(c, D, D0, dt, /*defer=*/true));
    // right-hand sides and first guesses together (zeroed sources are valid everywhere and skipped); async: the first launch
    // of the diffusion runs its interior strips while the rows travel (op_diffuse_batch)
    TRY(need(c, {U, V, D, U0, V0, D0}, std::min(iters, c->halo), /*async=*/true));
    float av, bv, ad, bd;
    coefficients(c->n, dt, visc, &av, &bv);
    coefficients(c->n, dt, diff, &ad, &bd);
    const float dt0 = dt * (float)c->n;
    const Solve all[3] = {{1, U0, U, av, bv}, {2, V0, V, av, bv}, {0, D0, D, ad, bd}};
    if (c->nranks == 1) {
        TRY(op_diffuse_batch(c, all, 3, iters));
        TRY(project(c, U0, V0, /*p=*/U, /*div=*/V, iters));
        TRY(op_advect2(c, 1, U, U0, 2, V, V0, U0, V0, dt));
        const AdvectAfter dens{0, D, D0, dt};
        return project(c, U, V, /*p=*/U0, /*div=*/V0, iters, &dens);
    }
    // Slabs: each advect needs the global max |velocity| on the host -- a pipeline drain.  The
    // density diffusion depends on nothing in between, so most of it rides in the same launches as
    // u and v (the fused kernel is latency-bound: a third field per launch is nearly free) and its
    // last sweeps are held back, one launch behind each reduction, so the GPU stays busy while the
    // host waits and talks to its peers.
    // The split falls on launch boundaries of the solve's own schedule (the last two launches are the ones
    // held back): with fp16 storage every launch rounds once, so cutting a launch in two would change the
    // result (iters = 20 runs as 8 + 8 + 4 on one GPU and must do so here).
    int fill1 = 0, fill2 = 0;
    {
        const bool canonical = c->st == fluid::STORAGE_F16;
        const long long slab_cells = (long long)c->min_slab * c->n;
        const bool small = (canonical ? (long long)c->n * c->n : slab_cells) < c->tb_min_cells;
        std::vector<int> launches;
        for (int left = iters; left > 0;) {
            launches.push_back(pick_sweeps(c, left, left, canonical, small, slab_cells, /*all_mode4=*/false));
            left -= launches.back();
        }
        // about eight sweeps behind each reduction (whole launches; single-sweep kernels: eight launches)
        for (int* fill : {&fill2, &fill1})
            while (*fill < 8 && !launches.empty()) {
                *fill += launches.back();
                launches.pop_back();
            }
        if (fill1 == 0) std::swap(fill1, fill2);
    }
    if (c->stream2 && c->slab_overlap && std::min(iters, c->halo) >= iters) {
        // A slab is a small problem: its launches leave most of the chip idle (one rank's share of 8192^2 / 8 keeps a
        // quarter of the wave slots busy), and the density diffusion depends on nothing in the velocity path.  So it
        // runs beside that path on a second stream -- from the moment its rows have arrived until the density advect at
        // the end -- and also fills the two stretches in which the host waits for the advect bounds.  Only when the ghost
        // zones cover the whole solve: an exchange inside it would put collectives on two streams in an order the ranks
        // could not agree on.  It ping-pongs with TMP2, the velocity path with TMP0 / TMP1.
        HIP_TRY(hipEventRecord(c->ev_fork, c->stream));
        HIP_TRY(hipStreamWaitEvent(c->stream2, c->ev_fork, 0));
        hipStream_t main_stream = c->stream;
        c->stream = c->stream2;
        const int rc_dens = op_diffuse_batch(c, all + 2, 1, iters, 0, nullptr, /*scratch_base=*/2, /*keep_pending=*/true);
        hipError_t e_join = rc_dens == FLUID_OK ? hipEventRecord(c->ev_join, c->stream2) : hipSuccess;
        c->stream = main_stream;
        TRY(rc_dens);
        HIP_TRY(e_join);
        TRY(op_diffuse_batch(c, all, 2, iters));
        TRY(project(c, U0, V0, /*p=*/U, /*div=*/V, iters, nullptr, /*with_max=*/true));
        TRY(vmax_begin(c, U0, V0, /*have_max=*/true));
        TRY(advect_bounded(c, 0, {U0, V0}, dt0, [&] { return op_advect2(c, 1, U, U0, 2, V, V0, U0, V0, dt); }));
        TRY(project(c, U, V, /*p=*/U0, /*div=*/V0, iters, nullptr, /*with_max=*/true));
        TRY(vmax_begin(c, U, V, /*have_max=*/true));
        HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_join, 0));
        return advect_bounded(c, 1, {D0}, dt0, [&] { return op_advect(c, 0, D, D0, U, V, dt); });
    }
    const int rest = fill1 + fill2, head = iters - rest;
    if (head > 0) TRY(op_diffuse_batch(c, all, 3, head));
    if (rest > 0) TRY(op_diffuse_batch(c, all, 2, rest));
    TRY(xchg_join(c));
    TRY(project(c, U0, V0, /*p=*/U, /*div=*/V, iters, nullptr, /*with_max=*/true));
    TRY(vmax_begin(c, U0, V0, /*have_max=*/true));
    if (fill1 > 0) TRY(op_diffuse_batch(c, all + 2, 1, fill1));
    TRY(advect_bounded(c, 0, {U0, V0}, dt0, [&] { return op_advect2(c, 1, U, U0, 2, V, V0, U0, V0, dt); }));
    TRY(project(c, U, V, /*p=*/U0, /*div=*/V0, iters, nullptr, /*with_max=*/true));
    TRY(vmax_begin(c, U, V, /*have_max=*/true));
    if (fill2 > 0) TRY(op_diffuse_batch(c, all + 2, 1, fill2));
    return advect_bounded(c, 1, {D0}, dt0, [&] { return op_advect(c, 0, D, D0, U, V, dt); });
}

int zero_sources(fluid_ctx* c)
{
    // the reference's zeroing loop (FluidSequential.c:298-302): marked, see "fields that are zero by definition"
    for (int id : {FLUID_U_PREV, FLUID_V_PREV, FLUID_DENS_PREV}) mark_zero(c, id);
    return FLUID_OK;
}

// `wait` = false only enqueues (fp32 storage): step() / step_src() queue all their fields and wait once
int copy_rows(fluid_ctx* c, int field, float* host, const float* chost, int row_lo, int row_hi, bool to_device, bool wait = true)
{
    if (row_lo < 0 || row_hi > c->w || row_lo > row_hi) return fail(FLUID_E_INVALID, "bad row range");
    if (row_lo == row_hi) return FLUID_OK;
    // a download of a field kept scaled (fp16 storage: pressure, divergence) divides on the host, in float: exact, where a
    // pass over the fp16 field would round the plain values into fp16's subnormals again
    const float host_scale = (!to_device && c->st != fluid::STORAGE_F32) ? 1.0f / c->fscale[field] : 1.0f;
    TRY(materialize(c, field, /*keep_scale=*/host_scale != 1.0f));
    char* dev = static_cast<char*>(c->row(field, row_lo)) + (size_t)XOFF * c->esz;
    const size_t rows = (size_t)(row_hi - row_lo), w = (size_t)c->w;
    const size_t dp = (size_t)c->pitch * c->esz;
    if (to_device) c->reach[field] = 0;     // the caller vouches only for its own rows
    if (c->st == fluid::STORAGE_F32) {
        const size_t hp = w * sizeof(float);
        if (to_device)
            HIP_TRY(hipMemcpy2DAsync(dev, dp, chost + (size_t)row_lo * w, hp, hp, rows, hipMemcpyHostToDevice, c->stream));
        else
            HIP_TRY(hipMemcpy2DAsync(host + (size_t)row_lo * w, hp, dev, dp, hp, rows, hipMemcpyDeviceToHost, c->stream));
        if (wait) HIP_TRY(hipStreamSynchronize(c->stream));
        return FLUID_OK;
    }
    // fp16 storage: the ABI's host arrays stay float; convert through a host staging buffer
    // (round to nearest even on the way in, exact on the way out)
    std::vector<fluid::half_t> stage(rows * w);
    const size_t hp = w * sizeof(fluid::half_t);
    if (to_device) {
        const float* src = chost + (size_t)row_lo * w;
        for (size_t k = 0; k < rows * w; ++k) stage[k] = (fluid::half_t)src[k];
        HIP_TRY(hipMemcpy2DAsync(dev, dp, stage.data(), hp, hp, rows, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    } else {
        HIP_TRY(hipMemcpy2DAsync(stage.data(), hp, dev, dp, hp, rows, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        float* dst = host + (size_t)row_lo * w;
        for (size_t k = 0; k < rows * w; ++k) dst[k] = (float)stage[k] * host_scale;
    }
    return FLUID_OK;
}

// the context step() / step_src() keep between calls, one per calling thread; destroyed when the thread ends
// (for the main thread that is before the HIP runtime's own static destructors run)
struct CachedCtx {
    fluid_ctx* c = nullptr;
    ~CachedCtx();
};
thread_local CachedCtx g_cached_holder;
#define g_cached (g_cached_holder.c)

}  // namespace

// ===========================================================================
// extern "C" shim
// ===========================================================================
extern "C" {

const char* fluid_last_error(void) { return g_err.c_str(); }

int fluid_coefficients(int N, float dt, float coef, float* alpha, float* beta)
{
    if (N < 1 || !alpha || !beta) return fail(FLUID_E_INVALID, "fluid_coefficients: bad argument");
    coefficients(N, dt, coef, alpha, beta);
    return FLUID_OK;
}

int fluid_layout(int N, int* pitch, int* xoff, size_t* field_floats)
{
    if (N < 1 || N > kMaxN) return fail(FLUID_E_INVALID, "N must be in [1, %d] (got %d)", kMaxN, N);
    const int p = fluid::pitch_for(N);
    if (pitch) *pitch = p;
    if (xoff) *xoff = XOFF;
    if (field_floats) *field_floats = (size_t)(N + 2) * p;
    return FLUID_OK;
}

size_t fluid_arena_bytes_ex(int N, int storage)
{
    size_t ff = 0;
    if (fluid_layout(N, nullptr, nullptr, &ff) != FLUID_OK) return 0;
    if (storage != FLUID_STORAGE_F32 && storage != FLUID_STORAGE_F16) return 0;
    return ff * FLUID_NFIELDS * fluid::storage_bytes(storage) + kControlBytes;
}

size_t fluid_arena_bytes(int N) { return fluid_arena_bytes_ex(N, FLUID_STORAGE_F32); }

int fluid_create_ex(const fluid_config* cfg, fluid_ctx** out)
{
    if (!cfg || !out) return fail(FLUID_E_INVALID, "fluid_create_ex: null argument");
    *out = nullptr;
    const int n = cfg->n;
    if (n < 1 || n > kMaxN) return fail(FLUID_E_INVALID, "N must be in [1, %d] (got %d)", kMaxN, n);
    const int P = cfg->nranks < 1 ? 1 : cfg->nranks;
    if (cfg->rank < 0 || cfg->rank >= P) return fail(FLUID_E_INVALID, "rank %d outside [0,%d)", cfg->rank, P);
    if (cfg->jacobi_variant < 0 || cfg->jacobi_variant >= fluid::JACOBI_VARIANTS)
        return fail(FLUID_E_INVALID, "unknown Jacobi variant %d", cfg->jacobi_variant);
    if (cfg->storage != FLUID_STORAGE_F32 && cfg->storage != FLUID_STORAGE_F16)
        return fail(FLUID_E_INVALID, "unknown storage type %d", cfg->storage);
    if (P > 1 && n / P < 2) return fail(FLUID_E_INVALID, "N=%d is too small for %d row slabs (need >= 2 rows each)", n, P);
    fluid_ctx* c = new (std::nothrow) fluid_ctx;
    if (!c) return fail(FLUID_E_NOMEM, "out of host memory");
    c->n = n;
    c->w = n + 2;
    c->pitch = fluid::pitch_for(n);
    c->field_floats = (size_t)c->w * c->pitch;
    c->st = cfg->storage;
    c->esz = fluid::storage_bytes(c->st);
    c->field_bytes = c->field_floats * c->esz;
    c->variant = cfg->jacobi_variant;
    for (float& f : c->fscale) f = 1.0f;
    if (c->st == fluid::STORAGE_F16 && n >= 16) {
        // 2^(floor(log2 N) - 2): h * pscale lies in (1/8, 1/4], so a scaled divergence is at most a quarter of the velocity
        // differences it is formed from -- no fp16 overflow that the plain field would not have had 2^12 earlier -- and the
        // pressure of ordinary velocities sits in fp16's normal range instead of its subnormals
        int e = 0;
        (void)std::frexp((float)n, &e);                 // n = m * 2^e, m in [0.5, 1): floor(log2 n) = e - 1
        c->pscale = std::ldexp(1.0f, e - 3);
    }
    c->rank = cfg->rank;
    c->nranks = P;
    const int base = n / P, rem = n % P;
    c->own0 = 1 + cfg->rank * base + std::min(cfg->rank, rem);
    c->own1 = c->own0 + base + (cfg->rank < rem ? 1 : 0);
    c->min_slab = base;
    // ghost-zone depth: never reaches a neighbour's wall rows (depth <= slab-1)
    // default: deep enough that a 40-sweep solve (+ the gradient's row) needs one exchange -- halo
    // rows are latency-bound on xGMI (32 KiB per row at 8192^2) -- at ~4 % redundant rows on a
    // 1024-row slab; shallower on short slabs
    const int want = cfg->halo > 0 ? cfg->halo : std::max(4, std::min(42, base / 8));
    c->halo = P > 1 ? std::max(1, std::min(want, base - 1)) : 1;
    if (P > 1 && cfg->storage == FLUID_STORAGE_F16) {
        // fp16 results depend on the launch schedule (one rounding per launch); keeping it identical to the
        // one-GPU schedule needs ghost zones at least as deep as the longest fused launch
        if (base - 1 < 8) {
            delete c;
            return fail(FLUID_E_INVALID, "fp16 storage needs row slabs of at least 9 rows (N=%d over %d slabs gives %d)", n, P, base);
        }
        c->halo = std::max(c->halo, 8);
    }
    const size_t bytes = c->field_bytes * FLUID_NFIELDS + kControlBytes;
    int rc = FLUID_OK;
    auto bail = [&](int code) { fluid_destroy(c); return code; };
    if (cfg->arena) {
        if (cfg->arena_bytes < bytes) return bail(fail(FLUID_E_INVALID, "arena too small: %zu < %zu", cfg->arena_bytes, bytes));
        if (((uintptr_t)cfg->arena & 255u) != 0) return bail(fail(FLUID_E_INVALID, "arena must be 256-byte aligned"));
        c->arena = (char*)cfg->arena;
    } else {
        hipError_t e = hipMalloc((void**)&c->arena, bytes);
        if (e != hipSuccess) return bail(fail(e == hipErrorOutOfMemory ? FLUID_E_NOMEM : FLUID_E_HIP, "hipMalloc(%zu): %s", bytes, hipGetErrorString(e)));
        c->own_arena = true;
    }
    if (cfg->stream) {
        c->stream = (hipStream_t)cfg->stream;
    } else {
        hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
        if (e != hipSuccess) return bail(fail(FLUID_E_HIP, "hipStreamCreate: %s", hipGetErrorString(e)));
        c->own_stream = true;
    }
    for (int k = 0; k < FLUID_NFIELDS; ++k) c->f[k] = c->arena + (size_t)k * c->field_bytes;
    auto hip_ok = [&](hipError_t e, const char* what) {
        if (e == hipSuccess) return true;
        rc = fail(e == hipErrorOutOfMemory ? FLUID_E_NOMEM : FLUID_E_HIP, "%s: %s", what, hipGetErrorString(e));
        return false;
    };
    {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) == hipSuccess &&
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0)
            c->num_cu = cus;
    }
    if (P > 1) {
        if (!hip_ok(hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking), "hipStreamCreate(second stream)")) return bail(rc);
        if (!hip_ok(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming), "hipEventCreate")) return bail(rc);
        if (!hip_ok(hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming), "hipEventCreate")) return bail(rc);
        if (!hip_ok(hipStreamCreateWithFlags(&c->xstream, hipStreamNonBlocking), "hipStreamCreate(exchange stream)")) return bail(rc);
        if (!hip_ok(hipEventCreateWithFlags(&c->ev_xbegin, hipEventDisableTiming), "hipEventCreate")) return bail(rc);
        if (!hip_ok(hipEventCreateWithFlags(&c->ev_xdone, hipEventDisableTiming), "hipEventCreate")) return bail(rc);
    }
    if (!hip_ok(hipMemsetAsync(c->arena, 0, bytes, c->stream), "hipMemsetAsync(arena)")) return bail(rc);
    c->d_scalar = reinterpret_cast<unsigned int*>(c->arena + c->field_bytes * FLUID_NFIELDS);   // RCCL-addressable
    {
        const size_t words = 3 * (size_t)fluid::tile_rows(n) * fluid::tile_pitch(n);
        if (!hip_ok(hipMalloc((void**)&c->tiles, words * sizeof(unsigned)), "hipMalloc(tiles)")) return bail(rc);
        if (!hip_ok(hipMemsetAsync(c->tiles, 0, words * sizeof(unsigned), c->stream), "hipMemsetAsync(tiles)")) return bail(rc);
    }
    if (P > 1 && !hip_ok(hipMalloc((void**)&c->d_partials, fluid::kMaxPartials * sizeof(float)), "hipMalloc(partials)")) return bail(rc);
    if (!hip_ok(hipHostMalloc((void**)&c->h_scalar, 256, hipHostMallocDefault), "hipHostMalloc")) return bail(rc);
    if (!hip_ok(hipEventCreateWithFlags(&c->scalar_ready, hipEventDisableTiming), "hipEventCreate")) return bail(rc);
    if (!hip_ok(hipStreamSynchronize(c->stream), "hipStreamSynchronize")) return bail(rc);
    *out = c;
    return FLUID_OK;
}

int fluid_create(int N, fluid_ctx** out)
{
    fluid_config cfg;
    std::memset(&cfg, 0, sizeof cfg);
    cfg.n = N;
    cfg.nranks = 1;
    cfg.jacobi_variant = FLUID_JACOBI_TB;
    return fluid_create_ex(&cfg, out);
}

int fluid_destroy(fluid_ctx* c)
{
    if (!c) return FLUID_OK;
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->stream2) (void)hipStreamSynchronize(c->stream2);
    if (c->xstream) (void)hipStreamSynchronize(c->xstream);
    for (auto& p : c->ev_pool) {
        (void)hipEventDestroy(p.a);
        (void)hipEventDestroy(p.b);
    }
    if (!c->trials.empty()) {                    // measurements that die with the context are handed back to the tuner
        RbTuner& t = rb_tuner();
        std::lock_guard<std::mutex> lock(t.mu);
        for (auto& tr : c->trials) {
            auto it = t.table.find(tr.key);
            if (it != t.table.end() && !it->second.fixed && it->second.issued[tr.cand] > 0) it->second.issued[tr.cand] -= 1;
            (void)hipEventDestroy(tr.a);
            (void)hipEventDestroy(tr.b);
        }
    }
    for (hipEvent_t ev : c->free_events) (void)hipEventDestroy(ev);
    fluid_detail::rccl_release(c->rccl);
    c->rccl = nullptr;
    if (c->h_scalar) (void)hipHostFree(c->h_scalar);
    if (c->tiles) (void)hipFree(c->tiles);
    if (c->d_partials) (void)hipFree(c->d_partials);
    if (c->scalar_ready) (void)hipEventDestroy(c->scalar_ready);
    if (c->own_arena && c->arena) (void)hipFree(c->arena);
    if (c->stream2) (void)hipStreamDestroy(c->stream2);
    if (c->xstream) (void)hipStreamDestroy(c->xstream);
    if (c->ev_xbegin) (void)hipEventDestroy(c->ev_xbegin);
    if (c->ev_xdone) (void)hipEventDestroy(c->ev_xdone);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    if (g_cached == c) g_cached = nullptr;
    delete c;
    return FLUID_OK;
}

int fluid_synchronize(fluid_ctx* c)
{
    TRY(check_ctx(c));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return FLUID_OK;
}

int fluid_exchange_stream(fluid_ctx* c, void** stream)
{
    TRY(check_ctx(c));
    if (!stream) return fail(FLUID_E_INVALID, "null pointer");
    *stream = c->stream;                    // inside an exchange callback: the exchange stream (call_exchange)
    return FLUID_OK;
}

int fluid_split_launches(fluid_ctx* c, long long* count)
{
    TRY(check_ctx(c));
    if (!count) return fail(FLUID_E_INVALID, "null pointer");
    *count = c->split_launches;
    return FLUID_OK;
}

int fluid_owned_rows(fluid_ctx* c, int* row_lo, int* row_hi)
{
    TRY(check_ctx(c));
    if (row_lo) *row_lo = c->own0;
    if (row_hi) *row_hi = c->own1;
    return FLUID_OK;
}

int fluid_field_ptr(fluid_ctx* c, int field, void** dev_ptr)
{
    TRY(check_ctx(c));
    TRY(check_fields(c, {field}));
    if (!dev_ptr) return fail(FLUID_E_INVALID, "null pointer");
    if (c->in_halo_exchange) TRY(materialize_zero(c, field));
    else TRY(materialize(c, field));       // whoever asks for the address may read the memory
    *dev_ptr = c->f[field];
    return FLUID_OK;
}

int fluid_scalar_ptr(fluid_ctx* c, void** dev_ptr)
{
    TRY(check_ctx(c));
    if (!dev_ptr) return fail(FLUID_E_INVALID, "null pointer");
    *dev_ptr = c->d_scalar;
    return FLUID_OK;
}

int fluid_upload_rows(fluid_ctx* c, int field, const float* host, int row_lo, int row_hi)
{
    TRY(check_ctx(c));
    TRY(check_fields(c, {field}));
    if (!host) return fail(FLUID_E_INVALID, "null host pointer");
    return copy_rows(c, field, nullptr, host, row_lo, row_hi, true);
}

int fluid_download_rows(fluid_ctx* c, int field, float* host, int row_lo, int row_hi)
{
    TRY(check_ctx(c));
    TRY(check_fields(c, {field}));
    if (!host) return fail(FLUID_E_INVALID, "null host pointer");
    return copy_rows(c, field, host, nullptr, row_lo, row_hi, false);
}

int fluid_upload(fluid_ctx* c, int field, const float* host)
{
    TRY(check_ctx(c));
    return fluid_upload_rows(c, field, host, 0, c->w);
}

int fluid_download(fluid_ctx* c, int field, float* host)
{
    TRY(check_ctx(c));
    return fluid_download_rows(c, field, host, 0, c->w);
}

int fluid_fill(fluid_ctx* c, int field, float value)
{
    TRY(check_ctx(c));
    TRY(check_fields(c, {field}));
    if (value == 0.0f && !std::signbit(value)) {
        HIP_TRY(hipMemsetAsync(c->f[field], 0, c->field_bytes, c->stream));
        c->zero[field] = false;
        wrote(c, field, kEverywhere);
        return FLUID_OK;
    }
    std::vector<float> row((size_t)c->w * c->w, value);
    TRY(fluid_upload(c, field, row.data()));
    wrote(c, field, kEverywhere);
    return FLUID_OK;
}

int fluid_set_jacobi_variant(fluid_ctx* c, int variant)
{
    TRY(check_ctx(c));
    if (variant < 0 || variant >= fluid::JACOBI_VARIANTS) return fail(FLUID_E_INVALID, "unknown Jacobi variant %d", variant);
    c->variant = variant;
    return FLUID_OK;
}

int fluid_set_param(fluid_ctx* c, int key, int value)
{
    TRY(check_ctx(c));
    switch (key) {
    case FLUID_PARAM_TB_MAX_SWEEPS:
        if (value != 16 && value != 12 && value != 8 && value != 4 && value != 2) return fail(FLUID_E_INVALID, "TB_MAX_SWEEPS must be 16, 12, 8, 4 or 2");
        c->tb_max_t = value;
        return FLUID_OK;
    case FLUID_PARAM_TB_ROWS:
        if (value < 0) return fail(FLUID_E_INVALID, "TB_ROWS must be >= 0");
        c->tb_rows = value;
        return FLUID_OK;
    case FLUID_PARAM_TB_EDGE_ROWS_PCT:
        if (value < 0 || value > 100) return fail(FLUID_E_INVALID, "TB_EDGE_ROWS_PCT must be in [0,100]");
        c->tb_edge_pct = value;
        return FLUID_OK;
    case FLUID_PARAM_TB_MIN_CELLS:
        if (value < 0) return fail(FLUID_E_INVALID, "TB_MIN_CELLS must be >= 0");
        c->tb_min_cells = value;
        return FLUID_OK;
    case FLUID_PARAM_TB_FAST_DIVISION:
        if (value < 0 || value > 3) return fail(FLUID_E_INVALID, "TB_FAST_DIVISION must be 0, 1, 2 or 3");
        c->fast_div = value;
        return FLUID_OK;
    case FLUID_PARAM_TB_AUTOTUNE:
        c->autotune = value != 0;
        return FLUID_OK;
    case FLUID_PARAM_SLAB_OVERLAP:
        c->slab_overlap = value != 0;
        return FLUID_OK;
    case FLUID_PARAM_FUSE_DIVERGENCE:
        c->fuse_divergence = value != 0;
        return FLUID_OK;
    case FLUID_PARAM_EARLY_ADVECT:
        c->early_advect = value != 0;
        return FLUID_OK;
    case FLUID_PARAM_FUSE_ADD_SOURCE:
        c->fuse_add_source = value != 0;
        return FLUID_OK;
    case FLUID_PARAM_F16_PRESSURE_SCALE: {
        for (int f = 0; f < FLUID_NFIELDS; ++f) TRY(unscale(c, f));
        int e = 0;
        (void)std::frexp((float)c->n, &e);
        c->pscale = (value != 0 && c->st == fluid::STORAGE_F16 && c->n >= 16) ? std::ldexp(1.0f, e - 3) : 1.0f;
        return FLUID_OK;
    }
    case FLUID_PARAM_XCHG_OVERLAP:
        TRY(xchg_join(c));
        c->xchg_overlap = value != 0;
        return FLUID_OK;
    case FLUID_PARAM_TB_T16_MIN_CELLS:
        if (value < -1) return fail(FLUID_E_INVALID, "TB_T16_MIN_CELLS must be >= 0, or -1 for the default rule");
        c->tb_t16_min_cells = value;
        return FLUID_OK;
    case FLUID_PARAM_TB_LANE_COLUMNS:
        if (value != 2 && value != 4) return fail(FLUID_E_INVALID, "TB_LANE_COLUMNS must be 2 or 4");
        c->tb_nv = value;
        return FLUID_OK;
    case FLUID_PARAM_HALO:
        if (value < 1) return fail(FLUID_E_INVALID, "HALO must be >= 1");
        c->halo = c->nranks > 1 ? std::max(c->st == fluid::STORAGE_F16 ? 8 : 1, std::min(value, c->min_slab - 1)) : 1;
        return FLUID_OK;
    default:
        return fail(FLUID_E_INVALID, "unknown parameter %d", key);
    }
}

int fluid_plan_sweeps(int N, int rows, int storage, int pressure_form, int iters, int max_sweeps, int t16_min_cells, int* depths,
                      int capacity, int* count)
{
    if (N < 1 || N > kMaxN || rows < 1 || rows > N || iters < 0 || (iters & 1) || !depths || !count || capacity < 0)
        return fail(FLUID_E_INVALID, "fluid_plan_sweeps: bad argument");
    if (storage != FLUID_STORAGE_F32 && storage != FLUID_STORAGE_F16) return fail(FLUID_E_INVALID, "unknown storage type %d", storage);
    if (max_sweeps != 16 && max_sweeps != 12 && max_sweeps != 8 && max_sweeps != 4 && max_sweeps != 2)
        return fail(FLUID_E_INVALID, "max_sweeps must be 16, 12, 8, 4 or 2");
    fluid_ctx c;                            // host-side description only: no device, no stream
    c.n = N;
    c.st = storage;
    c.esz = fluid::storage_bytes(storage);
    c.field_bytes = (size_t)(N + 2) * fluid::pitch_for(N) * c.esz;
    c.tb_max_t = max_sweeps;
    c.tb_t16_min_cells = t16_min_cells;
    int k = 0;
    for (int left = iters; left > 0;) {
        const int t = pick_sweeps(&c, left, left, storage == FLUID_STORAGE_F16, false, (long long)rows * N, pressure_form != 0);
        if (k < capacity) depths[k] = t;
        ++k;
        left -= t;
    }
    *count = k;
    return FLUID_OK;
}

int fluid_autotune_pending(fluid_ctx* c, int* shapes_open)
{
    TRY(check_ctx(c));
    if (!shapes_open) return fail(FLUID_E_INVALID, "null pointer");
    tune_harvest(c);
    RbTuner& t = rb_tuner();
    std::lock_guard<std::mutex> lock(t.mu);
    int open = 0;
    for (auto& kv : t.table) open += kv.second.fixed ? 0 : 1;
    *shapes_open = open;
    return FLUID_OK;
}

int fluid_division_mode(fluid_ctx* c, float alpha, float beta, int* mode)
{
    TRY(check_ctx(c));
    if (!mode) return fail(FLUID_E_INVALID, "null pointer");
    *mode = c->variant == fluid::JACOBI_TB ? division_mode(c, beta, alpha).mode : 0;
    return FLUID_OK;
}

int fluid_set_exchange(fluid_ctx* c, fluid_exchange_fn fn, void* user)
{
    TRY(check_ctx(c));
    c->xchg = fn;
    c->xchg_user = user;
    return FLUID_OK;
}

int fluid_exchange_now(fluid_ctx* c, int kind, const int* fields, int nfields, int depth)
{
    TRY(check_ctx(c));
    if (!c->xchg) return fail(FLUID_E_COMM, "no exchange installed");
    if (kind != FLUID_XCHG_HALO && kind != FLUID_XCHG_GATHER) return fail(FLUID_E_INVALID, "fluid_exchange_now moves rows: HALO or GATHER");
    if (nfields < 0 || (nfields > 0 && !fields)) return fail(FLUID_E_INVALID, "bad field list");
    // the depth is checked against the SHORTEST slab, which every rank knows: a check against this rank's own height
    // would let the tall ranks of an uneven split into the collective while the short ones return
    if (kind == FLUID_XCHG_HALO && c->nranks > 1 && (depth < 1 || depth > c->min_slab))
        return fail(FLUID_E_INVALID, "halo depth %d outside [1, %d] (the shortest slab)", depth, c->min_slab);
    for (int k = 0; k < nfields; ++k) {
        TRY(check_fields(c, {fields[k]}));
        TRY(settle(c, fields[k], /*keep_scale=*/true));   // the caller is about to look at the rows: no increment may stay pending
                                                          // (a scale stays: it is the same on every rank, and a download undoes it)
    }
    c->in_halo_exchange = true;
    const int rc = call_exchange(c, kind, fields, nfields, depth, nullptr);
    c->in_halo_exchange = false;
    if (rc != 0) return fail(FLUID_E_COMM, "exchange failed (kind %d, rc %d)", kind, rc);
    for (int k = 0; k < nfields; ++k)
        c->reach[fields[k]] = kind == FLUID_XCHG_GATHER ? kEverywhere : std::max(c->reach[fields[k]], depth);
    return FLUID_OK;
}

int fluid_vel_step(fluid_ctx* c, float dt, float visc, int iters)
{
    TRY(check_ctx(c));
    if (iters < 0 || (iters & 1)) return fail(FLUID_E_INVALID, "sweep count must be even and >= 0 (got %d)", iters);
    TRY(vel_step(c, dt, visc, iters));
    HIP_TRY(hipGetLastError());
    return FLUID_OK;
}

int fluid_dens_step(fluid_ctx* c, float dt, float diff, int iters)
{
    TRY(check_ctx(c));
    if (iters < 0 || (iters & 1)) return fail(FLUID_E_INVALID, "sweep count must be even and >= 0 (got %d)", iters);
    TRY(dens_step(c, dt, diff, iters));
    HIP_TRY(hipGetLastError());
    return FLUID_OK;
}

int fluid_step(fluid_ctx* c, float dt, float diff, float visc, int iters, int nsteps, int use_sources)
{
    TRY(check_ctx(c));
    if (iters < 0 || (iters & 1)) return fail(FLUID_E_INVALID, "sweep count must be even and >= 0 (got %d)", iters);
    if (nsteps < 0) return fail(FLUID_E_INVALID, "nsteps < 0");
    for (int z = 0; z < nsteps; ++z) {
        if (!(use_sources && z == 0)) TRY(zero_sources(c));
        TRY(full_step(c, dt, diff, visc, iters));
    }
    HIP_TRY(hipGetLastError());
    return FLUID_OK;
}

// ---- operators ---------------------------------------------------------------
int fluid_op_set_bnd(fluid_ctx* c, int b, int x)
{
    TRY(check_ctx(c));
    TRY(check_fields(c, {x}));
    if (b < 0 || b > 2) return fail(FLUID_E_INVALID, "b must be 0, 1 or 2");
    if (c->nranks != 1) return fail(FLUID_E_INVALID, "fluid_op_set_bnd is a whole-grid operator (1 GPU)");
    TRY(materialize(c, x));
    fluid::launch_set_bnd(c->stream, c->st, c->f[x], c->pitch, c->n, b);
    HIP_TRY(hipGetLastError());
    return FLUID_OK;
}

int fluid_op_add_source(fluid_ctx* c, int x, int s, float dt)
{
    TRY(check_ctx(c));
    TRY(check_fields(c, {x, s}));
    if (x == s) return fail(FLUID_E_INVALID, "add_source: x and s must differ");
    TRY(op_add_source(c, x, s, dt));
    HIP_TRY(hipGetLastError());
    return FLUID_OK;
}

int fluid_op_jacobi_sweep(fluid_ctx* c, int b, int x, int x0, int out, float alpha, float beta)
{
    TRY(check_ctx(c));
    TRY(check_fields(c, {x, x0, out}));
    if (b < 0 || b > 2) return fail(FLUID_E_INVALID, "b must be 0, 1 or 2");
    if (out == x || out == x0) return fail(FLUID_E_INVALID, "jacobi_sweep: out must not alias an input");
    TRY(materialize(c, {x, x0}));
    c->zero[out] = false;
    TRY(need(c, {x}, 1));
    const int v1 = c->variant == fluid::JACOBI_TB ? fluid::JACOBI_STREAM : c->variant;   // one sweep: nothing to block
    fluid::launch_jacobi(c->stream, c->st, v1, c->f[x], c->f[x0], c->f[out], c->pitch, c->n, c->own0, c->own1, alpha, beta, b);
    wrote(c, out, 0);
    HIP_TRY(hipGetLastError());
    return FLUID_OK;
}

int fluid_op_diffuse(fluid_ctx* c, int b, int x, int x0, float alpha, float beta, int iters)
{
    TRY(check_ctx(c));
    TRY(check_fields(c, {x, x0}));
    if (b < 0 || b > 2) return fail(FLUID_E_INVALID, "b must be 0, 1 or 2");
    return op_diffuse(c, b, x, x0, alpha, beta, iters);
}

int fluid_op_advect(fluid_ctx* c, int b, int d, int d0, int u, int v, float dt)
{
    TRY(check_ctx(c));
    TRY(check_fields(c, {d, d0, u, v}));
    if (b < 0 || b > 2) return fail(FLUID_E_INVALID, "b must be 0, 1 or 2");
    if (d == d0 || d == u || d == v) return fail(FLUID_E_INVALID, "advect: output must not alias an input");
    TRY(advect_prepare(c, {d0}, u, v, dt * (float)c->n));
    TRY(op_advect(c, b, d, d0, u, v, dt));
    HIP_TRY(hipGetLastError());
    return FLUID_OK;
}

int fluid_op_divergence(fluid_ctx* c, int u, int v, int p, int div)
{
    TRY(check_ctx(c));
    TRY(check_fields(c, {u, v, p, div}));
    TRY(op_divergence(c, u, v, p, div));
    HIP_TRY(hipGetLastError());
    return FLUID_OK;
}

int fluid_op_subtract_gradient(fluid_ctx* c, int u, int v, int p)
{
    TRY(check_ctx(c));
    TRY(check_fields(c, {u, v, p}));
    TRY(op_subtract_gradient(c, u, v, p));
    HIP_TRY(hipGetLastError());
    return FLUID_OK;
}

// ---- diagnostics --------------------------------------------------------------
int fluid_residual(fluid_ctx* c, int x, int x0, float alpha, float beta, float* out)
{
    TRY(check_ctx(c));
    TRY(check_fields(c, {x, x0}));
    if (!out) return fail(FLUID_E_INVALID, "null pointer");
    TRY(materialize(c, {x, x0}));
    TRY(need(c, {x}, 1));
    HIP_TRY(hipMemsetAsync(c->d_scalar, 0, sizeof(unsigned), c->stream));
    fluid::launch_residual(c->stream, c->st, c->f[x], c->f[x0], c->pitch, c->n, c->own0, c->own1, alpha, beta, c->d_scalar);
    TRY(reduce_to_host(c, out));
    return exchange(c, FLUID_XCHG_MAX, {}, 0, out);
}

int fluid_absmax_velocity(fluid_ctx* c, int u, int v, float* out)
{
    TRY(check_ctx(c));
    TRY(check_fields(c, {u, v}));
    if (!out) return fail(FLUID_E_INVALID, "null pointer");
    TRY(materialize(c, {u, v}));
    HIP_TRY(hipMemsetAsync(c->d_scalar, 0, sizeof(unsigned), c->stream));
    fluid::launch_absmax2(c->stream, c->st, c->f[u], c->f[v], c->pitch, c->n, c->own0, c->own1, c->d_scalar);
    TRY(reduce_to_host(c, out));
    return exchange(c, FLUID_XCHG_MAX, {}, 0, out);
}

// ---- timing -------------------------------------------------------------------
int fluid_timing_enable(fluid_ctx* c, int on)
{
    TRY(check_ctx(c));
    TRY(timing_collect(c));
    c->timing = on != 0;
    return FLUID_OK;
}

int fluid_timing_read(fluid_ctx* c, fluid_timing* out, int reset)
{
    TRY(check_ctx(c));
    if (!out) return fail(FLUID_E_INVALID, "null pointer");
    TRY(timing_collect(c));
    out->jacobi_ms = c->cat_ms[FLUID_TIME_DIFFUSION];
    out->sweeps = c->sweeps;
    out->jacobi_launches = c->launches;
    out->jacobi_field_launches = c->field_launches;
    out->pressure_ms = c->pressure_ms;
    out->pressure_sweeps = c->pressure_sweeps;
    out->solves = c->cat_calls[FLUID_TIME_DIFFUSION];
    for (int k = 0; k < FLUID_TIMING_CATEGORIES; ++k) {
        out->category_ms[k] = c->cat_ms[k];
        out->category_calls[k] = c->cat_calls[k];
    }
    if (reset) {
        for (int k = 0; k < FLUID_TIMING_CATEGORIES; ++k) {
            c->cat_ms[k] = 0.0;
            c->cat_calls[k] = 0;
        }
        c->sweeps = 0;
        c->launches = c->field_launches = 0;
        c->pressure_ms = 0.0;
        c->pressure_sweeps = 0;
    }
    return FLUID_OK;
}

// Opt-in, NOT the reference's behaviour (it always runs a fixed count,
// FluidSequential.c:91): Jacobi in blocks of `check_every` sweeps until the
// max-norm residual drops to `tol` or `max_iters` is reached.
int fluid_op_diffuse_tol(fluid_ctx* c, int b, int x, int x0, float alpha, float beta, float tol, int max_iters,
                         int check_every, int* iters_done, float* residual)
{
    TRY(check_ctx(c));
    TRY(check_fields(c, {x, x0}));
    if (b < 0 || b > 2) return fail(FLUID_E_INVALID, "b must be 0, 1 or 2");
    if (check_every < 2 || (check_every & 1) || max_iters < 0 || !(tol >= 0.f))
        return fail(FLUID_E_INVALID, "diffuse_tol: check_every must be even and >= 2, max_iters >= 0, tol >= 0");
    int done = 0;
    float res = 0.f;
    TRY(fluid_residual(c, x, x0, alpha, beta, &res));
    while (res > tol && done < max_iters) {
        const int blk = std::min(check_every, (max_iters - done) & ~1);
        if (blk <= 0) break;
        TRY(op_diffuse(c, b, x, x0, alpha, beta, blk));
        done += blk;
        TRY(fluid_residual(c, x, x0, alpha, beta, &res));
    }
    if (iters_done) *iters_done = done;
    if (residual) *residual = res;
    return FLUID_OK;
}

// ---- the reference's loop body on host arrays -----------------------------------
// The host arrays are ordinary pageable memory (the reference malloc()s them, FluidSequential.c:277-282) and travel
// at PCIe speed as they are: measured 8.7 ms per step() at 4096^2 against 7.1 ms for the same 384 MiB through pinned
// buffers plus 1.6 ms of compute (tools/step_timing.py).  All copies of a call are enqueued on the context's stream
// and waited for once.
int fluid_release_cached(void)
{
    fluid_ctx* c = g_cached;
    g_cached = nullptr;
    return fluid_destroy(c);
}

static int cached_ctx(int N, fluid_ctx** out)
{
    if (g_cached && g_cached->n != N) TRY(fluid_release_cached());
    if (!g_cached) TRY(fluid_create(N, &g_cached));
    *out = g_cached;
    return FLUID_OK;
}

int step_src(int N, float dt, float diff, float visc, int iters, float* u, float* v, float* dens, float* u_prev,
             float* v_prev, float* dens_prev)
{
    if (!u || !v || !dens || !u_prev || !v_prev || !dens_prev) return fail(FLUID_E_INVALID, "step_src: null field");
    if (iters < 0 || (iters & 1)) return fail(FLUID_E_INVALID, "sweep count must be even and >= 0 (got %d)", iters);
    fluid_ctx* c;
    TRY(cached_ctx(N, &c));
    float* host[6] = {u, v, dens, u_prev, v_prev, dens_prev};
    for (int k = 0; k < 6; ++k) TRY(copy_rows(c, k, nullptr, host[k], 0, c->w, true, false));
    TRY(fluid_step(c, dt, diff, visc, iters, 1, 1));
    for (int k = 0; k < 6; ++k) TRY(copy_rows(c, k, host[k], nullptr, 0, c->w, false, false));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return FLUID_OK;
}

int step(int N, float dt, float diff, float visc, float* u, float* v, float* dens)
{
    if (!u || !v || !dens) return fail(FLUID_E_INVALID, "step: null field");
    fluid_ctx* c;
    TRY(cached_ctx(N, &c));
    float* host[3] = {u, v, dens};
    for (int k = 0; k < 3; ++k) TRY(copy_rows(c, k, nullptr, host[k], 0, c->w, true, false));
    TRY(fluid_step(c, dt, diff, visc, 40, 1, 0));   // 40 sweeps: FluidSequential.c:91
    for (int k = 0; k < 3; ++k) TRY(copy_rows(c, k, host[k], nullptr, 0, c->w, false, false));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return FLUID_OK;
}

}  // extern "C"

namespace {
CachedCtx::~CachedCtx()
{
    fluid_ctx* mine = c;
    c = nullptr;
    if (mine) (void)fluid_destroy(mine);
}
}  // namespace
