// fluid_kernels.hip -- hand-written gfx950 (CDNA4) kernels for the
// Stable-Fluids step: boundary, sources, Jacobi sweep, advection, divergence,
// pressure-gradient subtraction, plus two wavefront reductions.
//
// Arithmetic contract (bit parity with the reference's
// project/sequential/FluidSequential.c): every expression keeps the reference's
// operand order, division is true IEEE division, and this file is compiled with
// -ffp-contract=off so no multiply-add is fused.
//
// Device field layout (see DESIGN.md "Data layout in HBM"): a field is W=n+2
// rows of `pitch` floats; column c of a row sits at float index c+XOFF with
// XOFF=63, so interior column 1 starts a 256-byte line and a wave's 64 float4
// accesses cover whole 128-byte lines.  Pad floats (index <63 and >n+64) are
// zero-initialised and never read as data.
//
// Every stencil kernel takes a global interior row range [row_lo,row_hi) so the
// same code serves one GPU (1..n) and a row slab of a multi-GPU run, and every
// kernel applies the reference's set_bnd (FluidSequential.c:62-75) itself: the
// thread that produces an interior cell next to a wall also writes the ghost
// cell(s) derived from it, so no separate boundary launch is needed.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <cstdlib>

#include "fluid_kernels.h"

namespace fluid {

// ---------------------------------------------------------------------------
// helpers
// ---------------------------------------------------------------------------
__device__ __forceinline__ float flip_if(bool c, float v) { return c ? -v : v; }

// Field storage type S: float (the reference's arithmetic, bit for bit) or
// _Float16 (BASELINE config 4: fp16 fields, every operation still in fp32 on the
// widened values, one round-to-nearest when a kernel stores).
// With fp16 storage hipcc likes to fold the widening / narrowing conversions into
// mixed-precision instructions (observed: fmul + fptrunc -> v_fma_mixlo_f16 x, y, 0),
// which rounds the exact product straight to fp16 and adds a +0 that flips -0
// results: neither is "the fp32 operation, then one rounding on store".  An empty
// asm on the value (no instruction) keeps the conversions separate.
__device__ __forceinline__ float keep_f32(float v) { asm("" : "+v"(v)); return v; }
__device__ __forceinline__ float ld1(const float* p) { return *p; }
__device__ __forceinline__ float ld1(const half_t* p) { return keep_f32((float)*p); }
__device__ __forceinline__ void st1(float* p, float v) { *p = v; }
__device__ __forceinline__ void st1(half_t* p, float v) { *p = (half_t)keep_f32(v); }
// a value as it reads back after a store in the storage type (fp16 storage: rounded once; fp32: itself)
template <typename S>
__device__ __forceinline__ float as_stored(float v)
{
    if constexpr (sizeof(S) == 2) return keep_f32((float)(S)keep_f32(v));
    else return v;
}
typedef half_t half4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 ld4(const half_t* p)
{
    const half4_t h = *reinterpret_cast<const half4_t*>(p);
    return make_float4(keep_f32((float)h.x), keep_f32((float)h.y), keep_f32((float)h.z), keep_f32((float)h.w));
}
__device__ __forceinline__ void st4(float* p, const float4& v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ void st4(half_t* p, const float4& v)
{
    half4_t h;
    h.x = (half_t)keep_f32(v.x); h.y = (half_t)keep_f32(v.y); h.z = (half_t)keep_f32(v.z); h.w = (half_t)keep_f32(v.w);
    *reinterpret_cast<half4_t*>(p) = h;
}

// Range-checked buffer loads/stores (buf_ldv / buf_stv below) for the fused kernel's steady state: the
// hardware range check (offset >= num_records: loads return 0, stores are
// dropped) replaces every `if` around a memory operation, so the loop body has a
// fixed number of them and hipcc can emit counted s_waitcnt vmcnt(N) instead of
// draining the queue (vmcnt(0)) once per iteration -- which is what keeps three
// rows of loads in flight.  kBufOff disables a lane: added to any in-field
// offset it stays below 2^32 and above num_records (fields are < 2 GiB).
constexpr unsigned kBufOff = 0x80000000u;
typedef unsigned uint4_t __attribute__((ext_vector_type(4)));
typedef unsigned uint2_t __attribute__((ext_vector_type(2)));

// value of lane-1 / lane+1 across the whole 64-wide wave (DPP wave shifts; one
// VALU op each, no LDS).  Lane 0 / 63 receive `edge`.
__device__ __forceinline__ float from_lane_below(float v, float edge)
{
    int r = __builtin_amdgcn_update_dpp(__float_as_int(edge), __float_as_int(v),
                                        0x138 /*wave_shr:1*/, 0xf, 0xf, false);
    return __int_as_float(r);
}
__device__ __forceinline__ float from_lane_above(float v, float edge)
{
    int r = __builtin_amdgcn_update_dpp(__float_as_int(edge), __float_as_int(v),
                                        0x130 /*wave_shl:1*/, 0xf, 0xf, false);
    return __int_as_float(r);
}

// Same shifts with bound_ctrl: lane 0 / 63 read 0.  No `old` operand to set up,
// so the compiler folds the shift into the consuming add (v_add_f32_dpp).
__device__ __forceinline__ float lane_below0(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float lane_above0(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130, 0xf, 0xf, true));
}

// Ghost cells that derive from interior cell (j,i) holding `val`
// (FluidSequential.c:65-74): across a vertical wall the ghost is -val when
// b==1, across a horizontal wall -val when b==2, else a copy; a corner is
// 0.5f*(horizontal neighbour + vertical neighbour), both of which are ghosts of
// the same corner-most interior cell.  Handles n==1 (a cell on several walls).
template <typename S>
__device__ __forceinline__ void emit_ghosts(S* __restrict__ f, size_t pitch, int n, int b, int j, int i, float val)
{
    const bool nx = (b == 1), ny = (b == 2);
    const bool left = (j == 1), right = (j == n), top = (i == 1), bot = (i == n);
    if (!(left | right | top | bot)) return;
    const float gx = flip_if(nx, val);
    const float gy = flip_if(ny, val);
    const float corner = 0.5f * (gy + gx);   // 0.5f*(x[horizontal nbr] + x[vertical nbr])
    S* r0 = f + XOFF;
    S* ri = f + (size_t)i * pitch + XOFF;
    S* rn = f + (size_t)(n + 1) * pitch + XOFF;
    if (left) st1(ri, gx);
    if (right) st1(ri + n + 1, gx);
    if (top) st1(r0 + j, gy);
    if (bot) st1(rn + j, gy);
    if (top & left) st1(r0, corner);
    if (top & right) st1(r0 + n + 1, corner);
    if (bot & left) st1(rn, corner);
    if (bot & right) st1(rn + n + 1, corner);
}

// ---------------------------------------------------------------------------
// a2  set_bnd as its own kernel (C-ABI operator + tests; the step itself uses
// the fused form).  One thread per edge index k in 1..n; thread k==1 / k==n
// also writes the corners from the values it just produced.
// ---------------------------------------------------------------------------
template <typename S>
__global__ __launch_bounds__(256) void k_set_bnd(S* __restrict__ f, int pitch, int n, int b)
{
    const int k = 1 + blockIdx.x * 256 + threadIdx.x;
    if (k > n) return;
    const size_t P = (size_t)pitch;
    const bool nx = (b == 1), ny = (b == 2);
    S* r0 = f + XOFF;
    S* rk = f + (size_t)k * P + XOFF;
    S* r1 = f + P + XOFF;
    S* rN = f + (size_t)n * P + XOFF;
    S* rn = f + (size_t)(n + 1) * P + XOFF;
    const float gl = flip_if(nx, ld1(rk + 1));      // x[0,k]
    const float gr = flip_if(nx, ld1(rk + n));      // x[n+1,k]
    const float gt = flip_if(ny, ld1(r1 + k));      // x[k,0]
    const float gb = flip_if(ny, ld1(rN + k));      // x[k,n+1]
    st1(rk, gl);
    st1(rk + n + 1, gr);
    st1(r0 + k, gt);
    st1(rn + k, gb);
    if (k == 1) {
        st1(r0, 0.5f * (gt + gl));                                  // x[1,0] + x[0,1]
        st1(rn, 0.5f * (gb + flip_if(nx, ld1(rN + 1))));            // x[1,n+1] + x[0,n]
    }
    if (k == n) {
        st1(r0 + n + 1, 0.5f * (gt + flip_if(nx, ld1(r1 + n))));    // x[n,0] + x[n+1,1]
        st1(rn + n + 1, 0.5f * (gb + gr));                          // x[n,n+1] + x[n+1,n]
    }
}

// ---------------------------------------------------------------------------
// a3  add_source: x += dt*s on every cell of rows [row_lo,row_hi), ghosts
// included (FluidSequential.c:78-82).  Streams whole padded rows as float4
// (pads are 0 and stay 0).
// ---------------------------------------------------------------------------
template <typename S>
__global__ __launch_bounds__(256) void k_add_source(S* __restrict__ x, const S* __restrict__ s, int pitch, int row_lo,
                                                    int row_hi, float dt)
{
    const int nvec = pitch >> 2;
    const size_t total = (size_t)(row_hi - row_lo) * nvec;
    S* xv = x + (size_t)row_lo * pitch;
    const S* sv = s + (size_t)row_lo * pitch;
    for (size_t t = (size_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (size_t)gridDim.x * 256) {
        float4 a = ld4(xv + 4 * t);
        const float4 c = ld4(sv + 4 * t);
        a.x = a.x + dt * c.x;
        a.y = a.y + dt * c.y;
        a.z = a.z + dt * c.z;
        a.w = a.w + dt * c.w;
        st4(xv + 4 * t, a);
    }
}

// add_source when the source field is known to be all +0: x += inc with inc = dt*(+0)
// formed on the host (so x = -0 still becomes +0 for dt >= 0, exactly as x + dt*0 does).
template <typename S>
__global__ __launch_bounds__(256) void k_add_zero_source(S* __restrict__ x, int pitch, int row_lo, int row_hi, float inc)
{
    const int nvec = pitch >> 2;
    const size_t total = (size_t)(row_hi - row_lo) * nvec;
    S* xv = x + (size_t)row_lo * pitch;
    for (size_t t = (size_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (size_t)gridDim.x * 256) {
        float4 a = ld4(xv + 4 * t);
        a.x = a.x + inc;
        a.y = a.y + inc;
        a.z = a.z + inc;
        a.w = a.w + inc;
        st4(xv + 4 * t, a);
    }
}

// x *= factor on rows [row_lo, row_hi) (factor a power of two: how a field kept scaled -- the pressure and its right-hand
// side with fp16 storage, see fluid_solver.hip: project -- goes back to its plain values for a reader that does not know)
template <typename S>
__global__ __launch_bounds__(256) void k_scale(S* __restrict__ x, int pitch, int row_lo, int row_hi, float factor)
{
    const int nvec = pitch >> 2;
    const size_t total = (size_t)(row_hi - row_lo) * nvec;
    S* xv = x + (size_t)row_lo * pitch;
    for (size_t t = (size_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (size_t)gridDim.x * 256) {
        float4 a = ld4(xv + 4 * t);
        a.x = a.x * factor;
        a.y = a.y * factor;
        a.z = a.z * factor;
        a.w = a.w * factor;
        st4(xv + 4 * t, a);
    }
}

// ---------------------------------------------------------------------------
// a4  Jacobi sweep, three variants.  All compute, for interior rows
// [row_lo,row_hi) and columns 1..n,
//     out = (x0 + alpha*(((L + R) + U) + D)) / beta     (FluidSequential.c:95-96)
// and the ghosts of `out` that derive from those cells.
// ---------------------------------------------------------------------------

// (i) naive-global: one thread per cell, five global loads.
template <typename S>
__global__ __launch_bounds__(256) void k_jacobi_naive(const S* __restrict__ x, const S* __restrict__ x0,
                                                      S* __restrict__ out, int pitch, int n, int row_lo,
                                                      int row_hi, float alpha, float beta, int b)
{
    const int j = 1 + blockIdx.x * 64 + (threadIdx.x & 63);
    const int i = row_lo + blockIdx.y * 4 + (threadIdx.x >> 6);
    if (j > n || i >= row_hi) return;
    const size_t P = (size_t)pitch;
    const S* c = x + (size_t)i * P + XOFF + j;
    float nb = ld1(c - 1) + ld1(c + 1);
    nb = nb + ld1(c - (ptrdiff_t)P);
    nb = nb + ld1(c + P);
    const float val = (ld1(x0 + (size_t)i * P + XOFF + j) + alpha * nb) / beta;
    st1(out + (size_t)i * P + XOFF + j, val);
    emit_ghosts(out, P, n, b, j, i, val);
}

// (ii) LDS-tiled: a (TY+2)x(TX+2) halo tile of x staged in LDS per workgroup
// (TX=64, TY=16, 256 threads, each thread 4 rows of one column).  Tile corners
// are not needed by a 5-point stencil and are not loaded.
constexpr int LT_X = 64, LT_Y = 16;
template <typename S>
__global__ __launch_bounds__(256) void k_jacobi_lds(const S* __restrict__ x, const S* __restrict__ x0,
                                                    S* __restrict__ out, int pitch, int n, int row_lo,
                                                    int row_hi, float alpha, float beta, int b)
{
    __shared__ float tile[LT_Y + 2][LT_X + 2 + 1];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;      // ty in 0..3
    const int j0 = 1 + blockIdx.x * LT_X, i0 = row_lo + blockIdx.y * LT_Y;
    const int j = j0 + tx;
    const size_t P = (size_t)pitch;
    const int rows = min(LT_Y, row_hi - i0);
    const int cols = min(LT_X, n - j0 + 1);
    // body + top/bottom halo rows: rows i0-1 .. i0+rows, 64 columns each
    for (int r = ty; r < rows + 2; r += 4)
        if (tx < cols) tile[r][tx + 1] = ld1(x + (size_t)(i0 - 1 + r) * P + XOFF + j);
    // left/right halo columns
    if (threadIdx.x < 2 * LT_Y) {
        const int r = threadIdx.x >> 1, side = threadIdx.x & 1;
        if (r < rows) {
            const int jj = side ? j0 + cols : j0 - 1;
            tile[r + 1][side ? cols + 1 : 0] = ld1(x + (size_t)(i0 + r) * P + XOFF + jj);
        }
    }
    __syncthreads();
    if (tx >= cols) return;
#pragma unroll
    for (int k = 0; k < LT_Y / 4; ++k) {
        const int r = ty + 4 * k;
        if (r >= rows) break;
        const int i = i0 + r;
        float nb = tile[r + 1][tx] + tile[r + 1][tx + 2];
        nb = nb + tile[r][tx + 1];
        nb = nb + tile[r + 2][tx + 1];
        const float val = (ld1(x0 + (size_t)i * P + XOFF + j) + alpha * nb) / beta;
        st1(out + (size_t)i * P + XOFF + j, val);
        emit_ghosts(out, P, n, b, j, i, val);
    }
}

// (iii) streaming: each lane owns one float4 (4 columns) and walks RB rows
// with a three-row register window, so every x row is fetched once per strip
// (+2 halo rows per RB); left/right neighbours come from the adjacent lanes by
// DPP wave shift, and only the wave's two edge lanes issue an extra dword load.
// 16 bytes per lane, 1 KiB per wave-instruction, 256-byte aligned.
template <int RB, typename S>
__global__ __launch_bounds__(256) void k_jacobi_stream(const S* __restrict__ x, const S* __restrict__ x0,
                                                       S* __restrict__ out, int pitch, int n, int row_lo,
                                                       int row_hi, float alpha, float beta, int b)
{
    const int lane = threadIdx.x & 63;
    const int vec = blockIdx.x * 256 + threadIdx.x;     // float4 index along the row
    const int nvec = (n + 3) >> 2;
    const int i0 = row_lo + blockIdx.y * RB;
    const int i1 = min(i0 + RB, row_hi);
    const bool active = vec < nvec;
    const int v = active ? vec : nvec - 1;              // clamp: inactive lanes load valid memory
    const int j = 1 + 4 * v;                            // first column of this lane
    const size_t P = (size_t)pitch;
    const S* xc = x + XOFF + j;
    const S* rc = x0 + XOFF + j;
    S* oc = out + XOFF + j;
    const bool edge_lo = (lane == 0), edge_hi = (lane == 63) | (vec >= nvec - 1);
    const bool nx = (b == 1), ny = (b == 2);

    float4 up = ld4(xc + (size_t)(i0 - 1) * P);
    float4 me = ld4(xc + (size_t)i0 * P);
    for (int i = i0; i < i1; ++i) {
        const float4 dn = ld4(xc + (size_t)(i + 1) * P);
        const float4 r = ld4(rc + (size_t)i * P);
        float le = 0.f, re = 0.f;
        if (edge_lo) le = ld1(xc + (size_t)i * P - 1);
        if (edge_hi) re = ld1(xc + (size_t)i * P + 4);
        float L = from_lane_below(me.w, le);
        float R = from_lane_above(me.x, re);
        if (edge_lo) L = le;
        if (edge_hi) R = re;
        float4 o;
        float nb;
        nb = L + me.y;     nb = nb + up.x; nb = nb + dn.x; o.x = (r.x + alpha * nb) / beta;
        nb = me.x + me.z;  nb = nb + up.y; nb = nb + dn.y; o.y = (r.y + alpha * nb) / beta;
        nb = me.y + me.w;  nb = nb + up.z; nb = nb + dn.z; o.z = (r.z + alpha * nb) / beta;
        nb = me.z + R;     nb = nb + up.w; nb = nb + dn.w; o.w = (r.w + alpha * nb) / beta;
        if (active) {
            S* orow = oc + (size_t)i * P;
            const int last = n - j;            // component index of column n (>=0)
            if (last >= 3) {
                st4(orow, o);
            } else {                           // ragged right end: columns j..n only
                st1(orow, o.x);
                if (last >= 1) st1(orow + 1, o.y);
                if (last >= 2) st1(orow + 2, o.z);
            }
            // ---- fused set_bnd (selects only: no runtime-indexed arrays) ----
            const float vn = last == 0 ? o.x : last == 1 ? o.y : last == 2 ? o.z : o.w;  // column n
            if (j == 1) st1(orow - 1, flip_if(nx, o.x));                   // x[0,i]
            if (last <= 3) st1(orow + last + 1, flip_if(nx, vn));          // x[n+1,i]
            if (i == 1 || i == n) {
                float4 g4;
                g4.x = flip_if(ny, o.x); g4.y = flip_if(ny, o.y);
                g4.z = flip_if(ny, o.z); g4.w = flip_if(ny, o.w);
                const float cl = 0.5f * (g4.x + flip_if(nx, o.x));          // corner next to column 1
                const float cr = 0.5f * (flip_if(ny, vn) + flip_if(nx, vn)); // corner next to column n
#pragma unroll
                for (int side = 0; side < 2; ++side) {
                    if (side == 0 ? i != 1 : i != n) continue;
                    S* g = oc + (side == 0 ? (size_t)0 : (size_t)(n + 1) * P);
                    if (last >= 3) {
                        st4(g, g4);
                    } else {
                        st1(g, g4.x);
                        if (last >= 1) st1(g + 1, g4.y);
                        if (last >= 2) st1(g + 2, g4.z);
                    }
                    if (j == 1) st1(g - 1, cl);
                    if (last <= 3) st1(g + last + 1, cr);
                }
            }
        }
        up = me;
        me = dn;
    }
}

// (iv) temporally blocked: T sweeps per launch, results identical to T launches
// of any variant above (the only way past 12 B/cell/sweep).
//
// One wave = one window of 64 lanes x NV columns (NV = 2: 128 columns, the
// default; NV = 4: 256) x one strip of `rb` output rows, marching down the rows
// with all T sweeps in flight: at step t it loads row t of x and x0 and, for
// s = 1..T, produces row t-s of "x after s sweeps" from the three rows of stage
// s-1 it still holds.  Everything lives in registers: per stage a three-row ring
// of its input, a queue of the last T x0 rows, and three rows of x/x0 prefetched
// ahead of use.  The time loop is unrolled by three so the rings rotate by
// renaming, not by moving.  Left/right neighbours come from the adjacent lanes
// by DPP; a wave has no one to ask at its two ends, so windows overlap by
// HL = ceil(T/NV) lanes per side and only the inner 64-2*HL lanes store (the
// wrong values creep inwards one column per sweep and never reach them).  Strips
// overlap by T rows per side the same way.  No LDS, no barriers.
//
// set_bnd is replayed inside the pipeline (it must be: sweep s+1 reads the
// ghosts of sweep s): the lanes holding ghost column 0 / n+1 overwrite that
// component from the neighbouring interior column after every stage (EDGE
// windows only), and ghost rows 0 / n+1 of a stage are the flipped rows 1 / n
// of the same stage (steps near a wall take the `general` path; all others a
// branch-free body).  Domain walls do not shrink the valid region.  Corner
// ghosts are never read by a 5-point stencil, so they are only formed in the
// final store.  Negation is a sign-bit XOR (bit-identical to the reference's
// unary minus, zeros and NaNs included).
//
// The division by the per-solve constant beta comes in three forms (DIVMODE):
// 0: (..)/beta, true division (about 20 VALU-op times on gfx950).
// 4: beta is a power of two and alpha == 1.0f (the pressure solve: 1, 4):
//   `beta` carries the exact reciprocal, (..)*rbeta is the same correctly rounded
//   result for every input, and x * 1.0f is x, so alpha is not applied at all.
// 2: (float)((double)(..) * yd) with yd = RN64(1/beta).  The double product is
//   within 2^-52 relative of the true quotient, while a quotient of two floats is
//   never that close to a float rounding boundary (the boundary would need an odd
//   25-bit significand times beta's to fit 24 bits), so the final conversion
//   rounds exactly as IEEE division does; zeros, denormals, inf and NaN need no
//   special case.  The one exception class -- quotients that land exactly on a
//   denormal midpoint, possible for a few even-integer-like beta -- is why the
//   solver proves each beta on the device before using modes 2 and 4:
//   k_validate_div compares them with a/beta for all 2^32 inputs.
// 3: the two-term reciprocal q = fma(x, hi, x*lo) with hi = RD32(1/beta), lo = RN32(1/beta - hi) > 0 -- two
//   PACKED instructions per pair of cells where mode 2 takes six scalar ones.  hi + lo carries 1/beta to
//   ~2^-48 relative, which rounds like the true quotient for every x that is zero or at least beta * 2^-98 in
//   magnitude (k_validate_div proves it for all such x; lo > 0 keeps -0 / beta = -0); below that x*lo loses bits
//   to underflow and q can be one ulp off.  The dividends here are x0 + alpha*nb, and a float sum is either
//   zero or no smaller than 2^-24 of its smaller operand, or than half its larger one: wherever the right-hand
//   side x0 is at least m in magnitude, EVERY dividend of EVERY sweep is 0 or >= 2^-25 * m, whatever the
//   neighbours hold.  So no per-value guard is needed, only a fact about x0: k_tile_min_abs reduces |x0| over
//   tiles of kTileRows x kTileCols cells once per solve (x0 does not change during a solve), and a wave takes
//   the two-term path if every tile its strip and window touch (overlaps included) is >= beta * 2^-72;
//   otherwise, and in the windows and strips on the domain's edge, it divides as mode 2 does.  Velocities
//   pass unless they have died out; a density with exact zeros outside its support falls back there.

// 5: Markstein's residual correction with the residual kept out of the underflow range by a power of two (round 3).
//   With r = RN32(1/beta) and S = 2^24:
//       q0 = D * r;   e = fma(q0, beta*S, D*(-S));   q = fma(e, -(r/S), q0)
//   -- four PACKED instructions per pair of cells where mode 2 takes six scalar ones.  e = -S*(D - q0*beta) is exact
//   (q0 is within an ulp of D/beta, so the residual has at most 24 significant bits, and the scale keeps its last bit
//   at or above 2^-148 even for denormal D: unscaled, it underflows for |D| < 2^-102 and the quotient comes out an ulp
//   off -- what DESIGN.md section 3 records of the unscaled form), and the last fma rounds q0 + (D - q0*beta)*r once,
//   which is RN(D/beta) by Markstein's theorem.  The signs are arranged so that -0 / beta = -0.  k_validate_div proves
//   it per beta for every |D| < 2^104.  Beyond that D*S overflows -- and then e, and q, are inf or NaN, never a finite
//   wrong number; every later sweep turns anything it computes from a non-finite value into a non-finite value, down
//   to the rows the wave stores.  So the wave keeps ONE sticky test on what it stores (fma(value, 0, acc): NaN as soon
//   as a value is inf or NaN), one packed instruction per time step, and a wave that stored anything non-finite runs
//   its strip again in mode 2 (stores are out of place, so a second pass simply overwrites the first).  Fields that
//   hold inf or NaN to begin with take that second pass too; nothing else ever does (|D| >= 2^104 ~ 2e31).
//   Unlike mode 3 this needs no fact about the data: the tiny and denormal ranges a decaying field crosses are exact.

// One lane's share of a row: NV consecutive columns (NV = 2 or 4).
template <int NV>
struct Vec {
    float c[NV];
};
typedef float v2f __attribute__((ext_vector_type(2)));
typedef half_t half2_t __attribute__((ext_vector_type(2)));

template <int NV>
__device__ __forceinline__ Vec<NV> buf_ldv(const float*, __amdgpu_buffer_rsrc_t r, unsigned off)
{
    Vec<NV> v;
    if constexpr (NV == 4) {
        const uint4_t u = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
        v.c[0] = __uint_as_float(u.x); v.c[1] = __uint_as_float(u.y); v.c[2] = __uint_as_float(u.z); v.c[3] = __uint_as_float(u.w);
    } else {
        const uint2_t u = __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, 0);
        v.c[0] = __uint_as_float(u.x); v.c[1] = __uint_as_float(u.y);
    }
    return v;
}
template <int NV>
__device__ __forceinline__ Vec<NV> buf_ldv(const half_t*, __amdgpu_buffer_rsrc_t r, unsigned off)
{
    Vec<NV> v;
    if constexpr (NV == 4) {
        const uint2_t u = __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, 0);
        half4_t h;
        __builtin_memcpy(&h, &u, 8);
        v.c[0] = keep_f32((float)h.x); v.c[1] = keep_f32((float)h.y); v.c[2] = keep_f32((float)h.z); v.c[3] = keep_f32((float)h.w);
    } else {
        const unsigned u = __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0);
        half2_t h;
        __builtin_memcpy(&h, &u, 4);
        v.c[0] = keep_f32((float)h.x); v.c[1] = keep_f32((float)h.y);
    }
    return v;
}
template <int NV>
__device__ __forceinline__ void buf_stv(float*, __amdgpu_buffer_rsrc_t r, unsigned off, const Vec<NV>& v)
{
    if constexpr (NV == 4) {
        uint4_t u;
        u.x = __float_as_uint(v.c[0]); u.y = __float_as_uint(v.c[1]); u.z = __float_as_uint(v.c[2]); u.w = __float_as_uint(v.c[3]);
        __builtin_amdgcn_raw_buffer_store_b128(u, r, off, 0, 0);
    } else {
        uint2_t u;
        u.x = __float_as_uint(v.c[0]); u.y = __float_as_uint(v.c[1]);
        __builtin_amdgcn_raw_buffer_store_b64(u, r, off, 0, 0);
    }
}
template <int NV>
__device__ __forceinline__ void buf_stv(half_t*, __amdgpu_buffer_rsrc_t r, unsigned off, const Vec<NV>& v)
{
    if constexpr (NV == 4) {
        half4_t h;
        h.x = (half_t)keep_f32(v.c[0]); h.y = (half_t)keep_f32(v.c[1]); h.z = (half_t)keep_f32(v.c[2]); h.w = (half_t)keep_f32(v.c[3]);
        uint2_t u;
        __builtin_memcpy(&u, &h, 8);
        __builtin_amdgcn_raw_buffer_store_b64(u, r, off, 0, 0);
    } else {
        half2_t h;
        h.x = (half_t)keep_f32(v.c[0]); h.y = (half_t)keep_f32(v.c[1]);
        unsigned u;
        __builtin_memcpy(&u, &h, 4);
        __builtin_amdgcn_raw_buffer_store_b32(u, r, off, 0, 0);
    }
}

// the per-solve constants of the division (DIVMODE above): what each mode reads
struct DivK {
    float beta;           // 0: beta; 4: the exact reciprocal; 5: r = RN32(1/beta); 2, 3: unused
    double yd;            // 2 (and 3's fallback): RN64(1/beta)
    float hi, lo;         // 3: hi = RD32(1/beta), lo = RN32(1/beta - hi); 5: hi = beta * 2^24, lo = -(r * 2^-24)
};
constexpr float kDiv5NegScale = -0x1p24f;

template <typename S, int NV>
struct TbArgs {
    const S* xc;          // x   + column offset of this lane
    const S* rc;          // x0  + column offset
    S* oc;                // out + column offset
    __amdgpu_buffer_rsrc_t bx, br, bo;   // the three fields as range-checked buffers
    unsigned ld_off;      // byte offset of this lane's vector in row 0 (kBufOff: lane loads nothing)
    unsigned st_off;      // same for the store (kBufOff unless the lane owns interior or ghost columns)
    unsigned row_bytes;
    unsigned m_lg;        // all ones in the lane whose last component is ghost column 0, else 0
    unsigned m_r[NV];     // all ones in the lane AND component that is ghost column n+1
    int n, q_lo, q_hi, cg;
    float alpha;
    DivK k;               // the division's constants
    S* dc;                // DIVSRC: the divergence field + column offset, its buffer, and -0.5f * h
    __amdgpu_buffer_rsrc_t bd;   // (ADDSRC: the field that receives x0 + dt * source, and dt in div_scale)
    float div_scale;
    int s_lo, s_hi;       // ADDSRC: rows of it this wave stores (its strip, plus the wall row next to it)
    unsigned sx, sy;      // sign masks: flip across vertical / horizontal walls
    float x0_inc;         // pending add_source increment of the right-hand side (-0.0f: none)
    bool st_rg_lane, is_lg;
};

__device__ __forceinline__ float fxor(float v, unsigned m) { return __uint_as_float(__float_as_uint(v) ^ m); }
template <int NV>
__device__ __forceinline__ Vec<NV> fxorv(const Vec<NV>& v, unsigned m)
{
    Vec<NV> o;
#pragma unroll
    for (int c = 0; c < NV; ++c) o.c[c] = fxor(v.c[c], m);
    return o;
}

// (r + alpha*nb) / beta for one cell (the validator) and for a pair of cells (the kernel; same
// operations, the compiler emits the packed form of each).
template <int DIVMODE>
__device__ __forceinline__ float tb_div(float num, const DivK& k)
{
    if (DIVMODE == 0) return num / k.beta;
    if (DIVMODE == 4) return num * k.beta;         // beta holds the exact reciprocal
    if (DIVMODE == 3) return __builtin_fmaf(num, k.hi, num * k.lo);
    if (DIVMODE == 5) {
        const float q0 = num * k.beta;
        const float e = __builtin_fmaf(q0, k.hi, num * kDiv5NegScale);
        return __builtin_fmaf(e, k.lo, q0);
    }
    return (float)((double)num * k.yd);
}
template <int DIVMODE>
__device__ __forceinline__ v2f tb_div2(v2f num, const DivK& k)
{
    if constexpr (DIVMODE == 4) return num * k.beta;
    else if constexpr (DIVMODE == 3) {             // v_pk_mul_f32 + v_pk_fma_f32 (four plain v_mul / v_fmac: measured no better)
        const v2f p = num * k.lo;
        return __builtin_elementwise_fma(num, (v2f){k.hi, k.hi}, p);
    } else if constexpr (DIVMODE == 5) {           // 2 x v_pk_mul_f32 + 2 x v_pk_fma_f32
        const v2f ds = num * kDiv5NegScale;
        const v2f q0 = num * k.beta;
        const v2f e = __builtin_elementwise_fma(q0, (v2f){k.hi, k.hi}, ds);
        return __builtin_elementwise_fma(e, (v2f){k.lo, k.lo}, q0);
    } else return (v2f){tb_div<DIVMODE>(num.x, k), tb_div<DIVMODE>(num.y, k)};
}

// One lane's vector of one stage.  Its columns are worked on in pairs (0,1), (2,3), which is how every
// row already sits in registers (a load fills aligned registers, each result is written as aligned
// pairs): the horizontal sum l+r is one scalar add per column -- the two at the ends of the vector
// reach into the neighbouring lane through DPP -- whose results land in adjacent registers, and
// everything after it is one packed instruction per pair (v_pk_add_f32 x3, v_pk_mul_f32 x1..2).
// 3 to 3.5 VALU instructions per cell instead of 6.5: a wave issues about one instruction per 4.6
// cycles whatever its kind (tools/ubench/valu_cost.hip), so instructions are what to save.
// Operand order and roundings are those of FluidSequential.c:93-94.
template <int DIVMODE, int NV>
__device__ __forceinline__ Vec<NV> tb_stencil(const Vec<NV>& up, const Vec<NV>& me, const Vec<NV>& dn, const Vec<NV>& r,
                                              float alpha, const DivK& k)
{
    float h[NV];
    h[0] = lane_below0(me.c[NV - 1]) + me.c[1];
#pragma unroll
    for (int c = 1; c < NV - 1; ++c) h[c] = me.c[c - 1] + me.c[c + 1];
    h[NV - 1] = me.c[NV - 2] + lane_above0(me.c[0]);
    Vec<NV> G;
#pragma unroll
    for (int p = 0; p < NV; p += 2) {
        v2f s = {h[p], h[p + 1]};
        s = s + (v2f){up.c[p], up.c[p + 1]};
        s = s + (v2f){dn.c[p], dn.c[p + 1]};
        if constexpr (DIVMODE != 4) s = s * alpha;       // mode 4: alpha == 1.0f, x * 1.0f is x
        s = (v2f){r.c[p], r.c[p + 1]} + s;
        s = tb_div2<DIVMODE>(s, k);
        G.c[p] = s.x;
        G.c[p + 1] = s.y;
    }
    return G;
}

// All 2^32 float bit patterns a: does tb_div<DIVMODE>(a) equal a / beta bit for bit
// (any NaN matches any NaN)?  Counts mismatches; the solver requires zero.
// Mode 3 (arg = hi): its fallback, mode 2, must be exact for every a, and the two-term quotient for every a
// that is zero or at least beta * 2^-98 in magnitude (the dividends the tile test lets through are zero or
// at least beta * 2^-97, see DIVMODE 3).
// Mode 5: exact for every a that the quotient of is finite ... or else NOT FINITE (inf / NaN, which the wave's sticky test
// on its stored rows catches, see DIVMODE 5) -- never a finite value that differs from a / beta.  In practice: exact for
// |a| < 2^104, non-finite beyond.
template <int DIVMODE>
__global__ __launch_bounds__(256) void k_validate_div(float beta, DivK k, unsigned long long* __restrict__ bad)
{
    unsigned long long n = 0;
    const unsigned long long stride = (unsigned long long)gridDim.x * 256;
    auto differs = [](float got, float ref) { return (ref != ref) ? !(got != got) : (__float_as_uint(got) != __float_as_uint(ref)); };
    DivK k2 = k;
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < (1ull << 32); i += stride) {
        const float a = __uint_as_float((unsigned)i);
        const float ref = a / beta;
        if constexpr (DIVMODE == 3) {
            const bool in_range = a == 0.0f || !(__builtin_fabsf(a) < beta * 0x1p-98f);
            n += differs(tb_div<2>(a, k2), ref) || (in_range && differs(tb_div<3>(a, k), ref));
        } else if constexpr (DIVMODE == 5) {
            const float got = tb_div<5>(a, k);
            const bool finite = __builtin_fabsf(got) < __builtin_inff();
            const bool must_be_exact = __builtin_fabsf(a) < 0x1p100f;       // (the claim is 2^104; what matters is: far beyond any field)
            n += differs(got, ref) && (finite || must_be_exact);
        } else {
            n += differs(tb_div<DIVMODE>(a, k), ref);
        }
    }
    if (n) atomicAdd(bad, n);
}

// ghost columns of one freshly computed row (set_bnd, FluidSequential.c:65-66), edge windows only.
// Branch-free: per-lane bit masks pick the one lane/component that is a ghost column and replace it
// by its (sign-flipped) interior neighbour; every other lane and component passes through unchanged.
// (Uniform branches here cost an edge-window wave ~2.7x an interior wave per step and made the two
// edge windows the tail of every launch.)  v1 / vn return column 1 / column n of the row, which the
// final stage needs for the corner cells.
__device__ __forceinline__ float bitsel(unsigned mask, float a, float b)      // mask ? a : b, per bit
{
    return __uint_as_float((__float_as_uint(a) & mask) | (__float_as_uint(b) & ~mask));
}

template <typename S, int NV>
__device__ __forceinline__ void tb_fix_columns(Vec<NV>& G, const TbArgs<S, NV>& a, float& v1, float& vn, bool need_corners)
{
    float o[NV];
#pragma unroll
    for (int c = 0; c < NV; ++c) o[c] = G.c[c];
    const float from_right = lane_above0(o[0]);          // column 1 as seen from the lane holding column 0
    const float from_left = lane_below0(o[NV - 1]);      // column n as seen from the next lane (n % NV == 0)
    G.c[0] = bitsel(a.m_r[0], fxor(from_left, a.sx), o[0]);
#pragma unroll
    for (int c = 1; c < NV - 1; ++c) G.c[c] = bitsel(a.m_r[c], fxor(o[c - 1], a.sx), o[c]);
    G.c[NV - 1] = bitsel(a.m_lg, fxor(from_right, a.sx), bitsel(a.m_r[NV - 1], fxor(o[NV - 2], a.sx), o[NV - 1]));
    if (need_corners) {          // folds to a constant once the stage loop is unrolled
        v1 = from_right;
        unsigned bits = __float_as_uint(from_left) & a.m_r[0];
#pragma unroll
        for (int c = 1; c < NV; ++c) bits |= __float_as_uint(o[c - 1]) & a.m_r[c];
        vn = __uint_as_float(bits);
    }
}

// final stage: store row q, its ghost columns, and ghost rows / corners -- WITHOUT branches.
// Every lane that holds anything to be stored writes its whole vector through the range-checked
// buffer: the ragged last vector and the two ghost-column lanes spill their surplus components
// into pad floats (never read as data), so edge windows issue the same single store per step as
// interior ones; wall strips add the two ghost rows, enabled by a select on the offset.  A fixed
// number of memory operations per step is what lets hipcc count them (see kBufOff).
template <bool EDGE, bool WALL, typename S, int NV>
__device__ __forceinline__ void tb_store_to(S* oc, __amdgpu_buffer_rsrc_t bo, const Vec<NV>& G, int q, bool mine, const TbArgs<S, NV>& a,
                                            float v1, float vn)
{
    auto on = [](bool c, unsigned off) { return c ? off : kBufOff; };
    buf_stv<NV>(oc, bo, on(mine, a.st_off) + (unsigned)q * a.row_bytes, G);
    if (WALL) {
        Vec<NV> g = fxorv<NV>(G, a.sy);                  // ghost row = flipped wall row ...
        if (EDGE) {                                      // ... except its two corner cells
            const float cl = 0.5f * (fxor(v1, a.sy) + fxor(v1, a.sx));
            const float cr = 0.5f * (fxor(vn, a.sy) + fxor(vn, a.sx));
            if (a.is_lg) g.c[NV - 1] = cl;
#pragma unroll
            for (int c = 0; c < NV; ++c)
                if (a.st_rg_lane && a.cg == c) g.c[c] = cr;
        }
        buf_stv<NV>(oc, bo, on(mine & (q == 1), a.st_off), g);                                        // row 0
        buf_stv<NV>(oc, bo, on(mine & (q == a.n), a.st_off) + (unsigned)(a.n + 1) * a.row_bytes, g);  // row n+1
    }
}
template <bool EDGE, bool WALL, typename S, int NV>
__device__ __forceinline__ void tb_store(const Vec<NV>& G, int q, bool mine, const TbArgs<S, NV>& a, float v1, float vn)
{
    tb_store_to<EDGE, WALL, S, NV>(a.oc, a.bo, G, q, mine, a, v1, vn);
}

// DIVSRC: the divergence of (u, v) on row t-1 (FluidSequential.c:151-152: (-0.5f*h) * (((uR - uL) + vD) - vU)) from the
// rows the wave holds -- u row t-1, v rows t-2 and t -- for the first launch of a pressure solve, which then needs
// no divergence kernel before it: the row becomes the launch's right-hand side and is stored for the later launches.
template <int NV>
__device__ __forceinline__ Vec<NV> tb_divergence(const Vec<NV>& u, const Vec<NV>& v_up, const Vec<NV>& v_dn, float scale)
{
    float g[NV];
#pragma unroll
    for (int c = 0; c < NV; ++c) {
        const float ur = c < NV - 1 ? u.c[c + 1] : lane_above0(u.c[0]);
        const float ul = c > 0 ? u.c[c - 1] : lane_below0(u.c[NV - 1]);
        g[c] = ur - ul;
    }
    Vec<NV> d;
#pragma unroll
    for (int p = 0; p < NV; p += 2) {
        v2f s = {g[p], g[p + 1]};
        s = s + (v2f){v_dn.c[p], v_dn.c[p + 1]};
        s = s - (v2f){v_up.c[p], v_up.c[p + 1]};
        s = s * scale;
        d.c[p] = s.x;
        d.c[p + 1] = s.y;
    }
    return d;
}

// the hand-over of a prefetched row into the pipeline, as register moves the compiler cannot see
// through (tb_step says why)
template <int NV>
__device__ __forceinline__ Vec<NV> take(const Vec<NV>& v)
{
    Vec<NV> o;
#pragma unroll
    for (int p = 0; p < NV; p += 2) {
        v2f in = {v.c[p], v.c[p + 1]}, out;
        asm("v_mov_b64 %0, %1" : "=v"(out) : "v"(in));
        o.c[p] = out.x;
        o.c[p + 1] = out.y;
    }
    return o;
}

// x0 queue shift as straight-line code (a loop here is recognised as memmove,
// which pins the whole queue in scratch memory).
template <int K, int N, int NV>
__device__ __forceinline__ void tb_qshift(Vec<NV> (&Q)[N])
{
    if constexpr (K >= 1) {
        Q[K] = Q[K - 1];
        tb_qshift<K - 1, N, NV>(Q);
    }
}

// DIVMODE 5's sticky test on a row of the last stage: nf turns NaN -- and stays NaN -- once a value is inf or NaN
// (x * 0 is +-0 for every finite x).  One packed instruction per pair and time step.
template <int DIVMODE, int NV>
__device__ __forceinline__ void tb_sticky(const Vec<NV>& G, v2f& nf)
{
    if constexpr (DIVMODE == 5) {
#pragma unroll
        for (int p = 0; p < NV; p += 2) nf = __builtin_elementwise_fma((v2f){G.c[p], G.c[p + 1]}, (v2f){0.f, 0.f}, nf);
    }
}

// One time step.  Ring s (s = 0..T-1) holds three consecutive rows of "x after
// s sweeps"; at phase PH its slots are up = PH, me = PH+1, fresh = PH+2 (mod 3).
// GEN = false: every row any stage touches at this step is interior -- a
// branch-free body.  GEN = true (the few steps of a wall strip that sit on rows
// 0 / n+1): per-stage checks, ghost rows of each stage regenerated from its rows 1 / n.
template <int T, int DIVMODE, bool EDGE, bool WALL, bool GEN, int PH, typename S, int NV, bool DIVSRC = false, bool ADDSRC = false>
__device__ __forceinline__ void tb_step(int t, Vec<NV> (&W)[T][3], Vec<NV> (&Q)[T + 1], Vec<NV> (&PX)[3], Vec<NV> (&PQ)[3],
                                        const TbArgs<S, NV>& a, Vec<NV> (&UR)[3], Vec<NV> (&VR)[3], v2f& nf)
{
    constexpr int UP = PH % 3, ME = (PH + 1) % 3, FR = (PH + 2) % 3;
    // stage 0: row t of x and x0, loaded three steps ago.  The hand-over is an opaque register move on
    // purpose: a plain assignment lets the allocator rename the prefetch slot into the ring and pay
    // for it with copies at the loop's back edge -- copies of rows still in flight, i.e. a wait for
    // every outstanding load once per iteration.
    __builtin_amdgcn_sched_barrier(0);                   // (and not hoisted into the previous step either)
    if constexpr (DIVSRC) {
        // the two prefetch slots carry u and v (the first guess is all +0 and is not read): rings of their last three rows
        UR[FR] = take<NV>(PX[PH]);
        VR[FR] = take<NV>(PQ[PH]);
    } else {
        W[0][FR] = take<NV>(PX[PH]);
        Q[0] = take<NV>(PQ[PH]);
        if constexpr (ADDSRC) {
            // add_source inside the solve's first launch (FluidSequential.c:78-82 with the SWAP of :181 / :201 / :209 behind
            // it): the first guess x IS the source field s and the right-hand side is x0 + dt * s -- formed here row by row,
            // in the reference's order (the product rounded, then the sum), handed to the stages and stored OUT OF PLACE for
            // the solve's later launches (neighbouring waves still read the raw rows of x0, so it cannot go back in place;
            // the solver swaps the two buffers afterwards).  Ghost cells included, as add_source touches every cell.
#pragma unroll
            for (int p = 0; p < NV; p += 2) {
                const v2f ds = (v2f){W[0][FR].c[p], W[0][FR].c[p + 1]} * a.div_scale;
                const v2f q = (v2f){Q[0].c[p], Q[0].c[p + 1]} + ds;
                Q[0].c[p] = as_stored<S>(q.x);           // fp16 storage: what a separate add_source pass would hand on
                Q[0].c[p + 1] = as_stored<S>(q.y);
            }
            buf_stv<NV>(a.dc, a.bd, (((t >= a.s_lo) & (t < a.s_hi)) ? a.st_off : kBufOff) + (unsigned)t * a.row_bytes, Q[0]);
        } else {
#pragma unroll
            for (int p = 0; p < NV; p += 2) {            // x0 + dt*0 where an add_source was deferred, else x0 + (-0) = x0
                const v2f q = (v2f){Q[0].c[p], Q[0].c[p + 1]} + a.x0_inc;
                Q[0].c[p] = q.x;
                Q[0].c[p + 1] = q.y;
            }
        }
    }
    {   // refill the slot with row t+3: three steps of arithmetic cover the memory latency.  Rows past
        // the field's end and lanes past its width fall outside the buffer and read as 0.
        const unsigned off = a.ld_off + (unsigned)(t + 3) * a.row_bytes;
        PX[PH] = buf_ldv<NV>(a.xc, a.bx, off);
        PQ[PH] = buf_ldv<NV>(a.rc, a.br, off);
    }
    // the loads must ISSUE here, three steps before their use: left to itself the scheduler sinks
    // them to the end of the unrolled iteration (shorter live ranges) and the wave then waits out a
    // full memory latency per iteration
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (DIVSRC) {
        // right-hand side row t-1 = divergence there; it is what stage 1 consumes at this step (Q[1]), and it is stored
        // (with the ghost cells set_bnd(0, div) would give it) for the solve's later launches and as the step's v_prev
        const int q = t - 1;
        Vec<NV> D = tb_divergence<NV>(UR[ME], VR[UP], VR[FR], a.div_scale);
#pragma unroll
        for (int c = 0; c < NV; ++c) D.c[c] = as_stored<S>(D.c[c]);     // fp16 storage: what a separate divergence pass would hand on
        Q[1] = D;
        float v1 = 0.f, vn = 0.f;
        if (EDGE) tb_fix_columns<S, NV>(D, a, v1, vn, WALL);
        tb_store_to<EDGE, (WALL || GEN), S, NV>(a.dc, a.bd, D, q, (q >= 1) & (q <= a.n) & (q >= a.q_lo) & (q < a.q_hi), a, v1, vn);
    }
    if constexpr (!GEN) {
#pragma unroll
        for (int s = 1; s <= T; ++s) {
            const int q = t - s;                         // row this stage produces now (wave-uniform)
            Vec<NV> G = tb_stencil<DIVMODE, NV>(W[s - 1][UP], W[s - 1][ME], W[s - 1][FR], Q[s], a.alpha, a.k);
            float v1 = 0.f, vn = 0.f;
            if (EDGE) tb_fix_columns<S, NV>(G, a, v1, vn, WALL && s == T);
            if (s < T) W[s][FR] = G;
            else {
                tb_sticky<DIVMODE, NV>(G, nf);
                tb_store<EDGE, WALL, S, NV>(G, q, (q >= a.q_lo) & (q < a.q_hi), a, v1, vn);
            }
        }
    } else {
        // ring writes stay unconditional (selected values), so the rings stay in registers
#pragma unroll
        for (int s = 1; s <= T; ++s) {
            const int q = t - s;
            const bool interior = (q >= 1 && q <= a.n);
            Vec<NV> G = tb_stencil<DIVMODE, NV>(W[s - 1][UP], W[s - 1][ME], W[s - 1][FR], Q[s], a.alpha, a.k);
            float v1 = 0.f, vn = 0.f;
            if (EDGE) tb_fix_columns<S, NV>(G, a, v1, vn, s == T);
            if (s < T) {
                const Vec<NV> me = W[s][ME];
                const Vec<NV> flipped_me = fxorv<NV>(me, a.sy);  // ghost row n+1 of this stage = flipped row n
                const Vec<NV> flipped_g = fxorv<NV>(G, a.sy);    // ghost row 0 of this stage = flipped row 1
                const bool bot_ghost = (q == a.n + 1), top_ghost = (q == 1);
#pragma unroll
                for (int c = 0; c < NV; ++c) {
                    W[s][FR].c[c] = bot_ghost ? flipped_me.c[c] : G.c[c];
                    W[s][ME].c[c] = top_ghost ? flipped_g.c[c] : me.c[c];
                }
            } else {
                tb_sticky<DIVMODE, NV>(G, nf);
                tb_store<EDGE, true, S, NV>(G, q, interior & (q >= a.q_lo) & (q < a.q_hi), a, v1, vn);
            }
        }
    }
    tb_qshift<T, T + 1, NV>(Q);
}

// returns (wave-uniform): did the last stage produce anything that is not finite (DIVMODE 5 only; false otherwise)
template <int T, int DIVMODE, bool EDGE, bool WALL, typename S, int NV, bool DIVSRC = false, bool ADDSRC = false>
__device__ __forceinline__ bool tb_march(int t0, int t1, const TbArgs<S, NV>& a)
{
    Vec<NV> W[T][3], Q[T + 1], PX[3], PQ[3], UR[3], VR[3];
    v2f nf = {0.f, 0.f};
    Vec<NV> zero;
#pragma unroll
    for (int c = 0; c < NV; ++c) zero.c[c] = 0.f;
#pragma unroll
    for (int s = 0; s < T; ++s) { W[s][0] = zero; W[s][1] = zero; W[s][2] = zero; }
#pragma unroll
    for (int s = 0; s <= T; ++s) Q[s] = zero;
#pragma unroll
    for (int d = 0; d < 3; ++d) { UR[d] = zero; VR[d] = zero; }
#pragma unroll
    for (int d = 0; d < 3; ++d) {                        // rows t0, t0+1, t0+2 in flight before the first step
        const unsigned off = a.ld_off + (unsigned)(t0 + d) * a.row_bytes;
        PX[d] = buf_ldv<NV>(a.xc, a.bx, off);
        PQ[d] = buf_ldv<NV>(a.rc, a.br, off);
        // a store that stores nothing (offset out of range): the loop is entered with the same
        // load, load, store sequence in flight that one of its iterations leaves behind, so the wait
        // for a row at the top of the loop counts 7 younger operations on both paths into it instead
        // of draining the queue
        buf_stv<NV>(a.oc, a.bo, kBufOff, zero);
        if constexpr (DIVSRC || ADDSRC) buf_stv<NV>(a.dc, a.bd, kBufOff, zero);      // (a step of these variants stores a second row)
        __builtin_amdgcn_sched_barrier(0);               // in this order
    }
    // whole triples only: up to two surplus steps load nothing (rows past the field) and store nothing
    // (q >= q_hi).  A step is "interior" when the rows its stages produce, t-T .. t-1, all lie in 1..n
    // and no stage below the last is on row 1 (whose ghost row it would have to regenerate):
    // T+1 <= t <= n+1.  Wall strips run the general body only for the triples that contain another
    // kind of step -- about T/3 of them per wall -- as three loops, so neither body pays for the
    // other's registers.
    int t = t0;
    if constexpr (WALL) {
        for (; t <= t1 && t < T + 1; t += 3) {
            tb_step<T, DIVMODE, EDGE, WALL, true, 0, S, NV, DIVSRC, ADDSRC>(t, W, Q, PX, PQ, a, UR, VR, nf);
            tb_step<T, DIVMODE, EDGE, WALL, true, 1, S, NV, DIVSRC, ADDSRC>(t + 1, W, Q, PX, PQ, a, UR, VR, nf);
            tb_step<T, DIVMODE, EDGE, WALL, true, 2, S, NV, DIVSRC, ADDSRC>(t + 2, W, Q, PX, PQ, a, UR, VR, nf);
        }
    }
    for (; t <= t1 && (!WALL || t + 1 <= a.n); t += 3) {
        tb_step<T, DIVMODE, EDGE, WALL, false, 0, S, NV, DIVSRC, ADDSRC>(t, W, Q, PX, PQ, a, UR, VR, nf);
        tb_step<T, DIVMODE, EDGE, WALL, false, 1, S, NV, DIVSRC, ADDSRC>(t + 1, W, Q, PX, PQ, a, UR, VR, nf);
        tb_step<T, DIVMODE, EDGE, WALL, false, 2, S, NV, DIVSRC, ADDSRC>(t + 2, W, Q, PX, PQ, a, UR, VR, nf);
    }
    if constexpr (WALL) {
        for (; t <= t1; t += 3) {
            tb_step<T, DIVMODE, EDGE, WALL, true, 0, S, NV, DIVSRC, ADDSRC>(t, W, Q, PX, PQ, a, UR, VR, nf);
            tb_step<T, DIVMODE, EDGE, WALL, true, 1, S, NV, DIVSRC, ADDSRC>(t + 1, W, Q, PX, PQ, a, UR, VR, nf);
            tb_step<T, DIVMODE, EDGE, WALL, true, 2, S, NV, DIVSRC, ADDSRC>(t + 2, W, Q, PX, PQ, a, UR, VR, nf);
        }
    }
    if constexpr (DIVMODE == 5) return __builtin_amdgcn_ballot_w64((nf.x != nf.x) | (nf.y != nf.y)) != 0ull;
    else return false;
}

// the four bodies of a march: edge windows replay the ghost columns, wall strips the ghost rows (both wave-uniform)
template <int T, int DIVMODE, typename S, int NV, bool DIVSRC = false, bool ADDSRC = false>
__device__ __forceinline__ bool tb_march_any(bool edge, bool wall, int t0, int t1, const TbArgs<S, NV>& a)
{
    if (edge) {
        if (wall) return tb_march<T, DIVMODE, true, true, S, NV, DIVSRC, ADDSRC>(t0, t1, a);
        return tb_march<T, DIVMODE, true, false, S, NV, DIVSRC, ADDSRC>(t0, t1, a);
    }
    if (wall) return tb_march<T, DIVMODE, false, true, S, NV, DIVSRC, ADDSRC>(t0, t1, a);
    return tb_march<T, DIVMODE, false, false, S, NV, DIVSRC, ADDSRC>(t0, t1, a);
}

// which (window, strip group) pairs exist (launch_jacobi_tb)
struct TbGrid {
    int inner_wins, inner_blocks, edge_wins, edge_blocks, first_right;
    int hole_lo, hole_hi;     // rows [hole_lo, hole_hi) inside [row_lo, row_hi) are left out (hole_lo >= hole_hi: none): the two
                              // edge parts of a launch split around a halo exchange run as ONE launch (launch_jacobi_tb)
};

// waves per SIMD the register allocator must leave room for (second launch-bound argument): the
// kernel hides its latencies (memory, dependent packed operations) only by running other waves
constexpr int tb_waves_per_simd(int T, int NV) { return NV == 2 ? (T <= 8 ? 4 : T <= 12 ? 3 : 2) : (T <= 4 ? 3 : 2); }

// blockIdx.z picks one of up to three independent solves of the same shape (u, v and density
// diffusion): more waves per launch, hence taller strips and less pipeline-fill redundancy.
template <int T, int DIVMODE, int NV, typename S, bool DIVSRC = false, bool ADDSRC = false>
__global__ __launch_bounds__(256, tb_waves_per_simd(T, NV)) void k_jacobi_tb(TbBatch batch, int pitch, int n, int row_lo,
                                                                             int row_hi, int rb, int rb_edge, TbGrid g)
{
    static_assert(!DIVSRC || DIVMODE == 4, "the divergence-sourced launch is the first launch of a pressure solve");
    static_assert(!(DIVSRC && ADDSRC) && (!ADDSRC || DIVMODE != 3), "one second store per launch; mode 3's tiles are taken from the summed field");
    // Workgroups are handed to the 8 XCDs round-robin by linear id, and each XCD has its own L2.  The
    // grid is one-dimensional, a multiple of 8 long, and renumbered so that XCD k works through the
    // k-th contiguous eighth of the (window, strip group) list, windows fastest: the blocks an XCD
    // runs at the same time are neighbouring windows of the same rows, and the columns they both
    // read (2*HL lanes per window, a third of the reads at T = 16) are served by that XCD's L2
    // instead of crossing the fabric twice.
    // Only blocks that have rows are enumerated: interior windows x their strip groups first, then the
    // edge windows (window 0 and the last one or two), which have more, shorter strips.
    const unsigned per_xcd = gridDim.x >> 3;
    unsigned vid = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
    if (vid >= (unsigned)(g.inner_blocks + g.edge_blocks)) return;
    int win, strip_group;
    if (vid < (unsigned)g.inner_blocks) {
        win = 1 + (int)(vid % (unsigned)g.inner_wins);
        strip_group = (int)(vid / (unsigned)g.inner_wins);
    } else {
        vid -= (unsigned)g.inner_blocks;
        const int e = (int)(vid % (unsigned)g.edge_wins);
        strip_group = (int)(vid / (unsigned)g.edge_wins);
        win = e == 0 ? 0 : g.first_right + e - 1;
    }
    const S* __restrict__ x = static_cast<const S*>(batch.x[blockIdx.z]);
    const S* __restrict__ x0 = static_cast<const S*>(batch.x0[blockIdx.z]);
    S* __restrict__ out = static_cast<S*>(batch.out[blockIdx.z]);
    const float alpha = batch.alpha[blockIdx.z], beta = batch.beta[blockIdx.z];
    const double yd = batch.yd[blockIdx.z];
    const int b = batch.b[blockIdx.z];
    constexpr int HL = (T + NV - 1) / NV;                // lanes of overlap per side: T columns
    constexpr int VS = 64 - 2 * HL;
    constexpr int C0 = (XOFF + 1) / NV;                  // vector index of column 1 within a row
    static_assert((XOFF + 1) % NV == 0 && HL <= C0, "window overlap must fit the row's left pad");
    const int lane = threadIdx.x & 63;
    // the wave index is uniform, but anything derived from threadIdx is divergent to hipcc: readfirstlane
    // makes the strip bounds (and with them the loop control and the store conditions) scalar
    const int strip = strip_group * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // the two windows that carry the ghost columns do extra work per stage; they get shorter strips
    // (rb_edge < rb, more of them) so that they do not finish long after everybody else
    const int kg = n / NV;                               // ghost column n+1 = component cg of vector kg
    const bool left_edge = (win == 0);                                                  // wave-uniform
    const bool right_edge = (kg >= win * VS - HL) && (kg < win * VS - HL + 64);         // wave-uniform
    const bool edge = left_edge || right_edge;
    const int rbw = edge ? rb_edge : rb;
    TbArgs<S, NV> a;
    a.k.yd = yd;
    a.k.lo = batch.lo[blockIdx.z];
    a.k.hi = batch.hi[blockIdx.z];
    a.x0_inc = batch.x0_inc[blockIdx.z];
    int seg_hi = row_hi;                                 // this wave's output rows [q_lo, q_hi): strips do not straddle the hole
    a.q_lo = row_lo + strip * rbw;
    if (g.hole_lo < g.hole_hi) {
        const int top = (g.hole_lo - row_lo + rbw - 1) / rbw;      // strips above the hole
        if (strip < top) seg_hi = g.hole_lo;
        else a.q_lo = g.hole_hi + (strip - top) * rbw;
    }
    if (a.q_lo >= seg_hi) return;                        // wave-uniform
    a.q_hi = min(a.q_lo + rbw, seg_hi);
    a.s_lo = a.q_lo == 1 ? 0 : a.q_lo;                   // (ADDSRC) the strips next to a wall store the wall row too
    a.s_hi = a.q_hi == n + 1 ? n + 2 : a.q_hi;
    const int k = win * VS - HL + lane;                  // vector index: columns 1+NV*k .. NV+NV*k
    const int nvec = (n + NV - 1) / NV;
    const bool ld_ok = k <= pitch / NV - C0 - 1;         // the whole vector lies inside the row (k >= -HL >= -C0 always)
    const ptrdiff_t cofs = (ptrdiff_t)(XOFF + 1) + NV * (ptrdiff_t)(ld_ok ? k : 0);
    a.xc = x + cofs;
    a.rc = x0 + cofs;
    a.oc = out + cofs;
    {
        const unsigned field_bytes = (unsigned)((size_t)(n + 2) * (size_t)pitch * sizeof(S));    // < 2 GiB (launch_jacobi_tb)
        // a first guess known to be all +0 (sources after step 0, the pressure) is never read: an
        // empty descriptor makes every load of it return 0
        // (DIVSRC: x and x0 are u and v, both read; the first guess is +0 by definition and lives nowhere)
        a.bx = __builtin_amdgcn_make_buffer_rsrc(const_cast<S*>(x), 0, (batch.x_zero[blockIdx.z] && !DIVSRC) ? 0u : field_bytes, 0x00020000);
        a.br = __builtin_amdgcn_make_buffer_rsrc(const_cast<S*>(x0), 0, field_bytes, 0x00020000);
        a.bo = __builtin_amdgcn_make_buffer_rsrc(out, 0, field_bytes, 0x00020000);
        S* dv = (DIVSRC || ADDSRC) ? static_cast<S*>(batch.div[blockIdx.z]) : out;
        a.dc = dv + cofs;
        a.bd = __builtin_amdgcn_make_buffer_rsrc(dv, 0, field_bytes, 0x00020000);
        a.div_scale = batch.div_scale;
        a.row_bytes = (unsigned)((size_t)pitch * sizeof(S));
        const unsigned col = (unsigned)((XOFF + 1 + NV * (ptrdiff_t)k) * (ptrdiff_t)sizeof(S));
        a.ld_off = ld_ok ? col : kBufOff;
    }
    a.n = n;
    a.cg = n % NV;
    a.is_lg = (k == -1);
    const bool is_rg = (k == kg);
    const bool own = (lane >= HL) && (lane < 64 - HL);
    const bool st_int = own && k >= 0 && k < nvec;       // stores interior columns
    a.st_rg_lane = is_rg && (own || k == nvec);          // stores ghost column n+1 (exactly one lane grid-wide)
    a.m_lg = a.is_lg ? 0xFFFFFFFFu : 0u;
#pragma unroll
    for (int c = 0; c < NV; ++c) a.m_r[c] = (is_rg && a.cg == c) ? 0xFFFFFFFFu : 0u;
    // lanes that store: owners of interior columns (the ragged last vector included), the lane whose
    // last component is ghost column 0 and the lane holding ghost column n+1; surplus components go to pads
    a.st_off = (ld_ok && (st_int || a.is_lg || a.st_rg_lane)) ? a.ld_off : kBufOff;
    a.sx = (b == 1) ? 0x80000000u : 0u;
    a.sy = (b == 2) ? 0x80000000u : 0u;
    a.alpha = alpha;
    a.k.beta = beta;
    const int t0 = max(0, a.q_lo - T), t1 = a.q_hi - 1 + T;
    // a strip whose input rows [q_lo-T, q_hi-1+T] all exist never needs a regenerated ghost row
    const bool wall = (a.q_lo < T) || (a.q_hi - 1 + T > n + 1);      // wave-uniform
    constexpr int DM = DIVMODE == 3 ? 2 : DIVMODE;                   // what mode 3 falls back to
    bool two_term = false;
    if constexpr (DIVMODE == 3) {
        // the two-term division if |x0| >= thr on every tile that this wave's part of the grid touches (DIVMODE 3
        // above): the x0 rows q_lo-T+1 .. q_hi+T-2 are the ones that reach a cell this wave stores, the columns are
        // those of its 64 * NV lanes; both clipped to the interior (ghost cells repeat interior values, and what
        // lanes past the row's ends compute is never stored)
        const unsigned* __restrict__ tiles = batch.tiles[blockIdx.z];
        const unsigned thr = batch.tile_thr[blockIdx.z];
        const int tr0 = (max(1, a.q_lo - T + 1) - 1) / kTileRows, tr1 = (min(n, a.q_hi + T - 2) - 1) / kTileRows;
        const int c0 = max(1, 1 + NV * (win * VS - HL)), c1 = min(n, NV * (win * VS - HL + 64));
        const int tc0 = (c0 - 1) / kTileCols, tc1 = (c1 - 1) / kTileCols, ntc = tc1 - tc0 + 1;
        const int count = (tr1 - tr0 + 1) * ntc;
        bool low = false;
        if (tiles != nullptr)
            for (int i = lane; i < count; i += 64)
                low |= tiles[(size_t)(tr0 + i / ntc) * batch.tile_pitch + (tc0 + i % ntc)] < thr;
        two_term = tiles != nullptr && __builtin_amdgcn_ballot_w64(low) == 0ull;
    }
    if (two_term) {
        tb_march_any<T, DIVMODE, S, NV>(edge, wall, t0, t1, a);
    } else {
        const bool again = tb_march_any<T, DM, S, NV, DIVSRC, ADDSRC>(edge, wall, t0, t1, a);
        // mode 5: something this wave stored is inf or NaN -- a dividend beyond 2^104, or a field that holds such values
        // to begin with.  The strip once more, dividing as mode 2 does; its stores replace the first pass's.
        if constexpr (DIVMODE == 5) {
            if (again) tb_march_any<T, 2, S, NV, false, ADDSRC>(edge, wall, t0, t1, a);
        }
    }
}

// |x0| minima over tiles of kTileRows x kTileCols interior cells (tile (r, c): rows 1 + r*kTileRows ..., columns
// 1 + c*kTileCols ...), as the bit pattern of a non-negative float (they order like unsigned integers), over the
// rows [row_lo, row_hi) only.  One wave per 32 rows x 256 columns: 16 lanes x 4 columns make one tile.
template <typename S>
__global__ __launch_bounds__(256) void k_tile_min_abs(TileBatch tb, int pitch, int n, int row_lo, int row_hi, int tile_row0, int tile_pitch)
{
    const S* __restrict__ f = static_cast<const S*>(tb.field[blockIdx.z]);
    unsigned* __restrict__ out = tb.tiles[blockIdx.z];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tr = tile_row0 + blockIdx.y;
    const int j = 1 + 4 * (lane + 64 * (blockIdx.x * 4 + wave));         // first of this lane's four columns
    const int i0 = max(row_lo, 1 + tr * kTileRows), i1 = min(row_hi, 1 + (tr + 1) * kTileRows);   // 1 <= row_lo, row_hi <= n+1
    const size_t P = (size_t)pitch;
    float m = __builtin_inff();
    if (j <= n)
        for (int i = i0; i < i1; ++i) {
            const float4 v = ld4(f + (size_t)i * P + XOFF + j);           // columns past n read ghost / pad floats: masked below
            m = fminf(m, fabsf(v.x));
            if (j + 1 <= n) m = fminf(m, fabsf(v.y));
            if (j + 2 <= n) m = fminf(m, fabsf(v.z));
            if (j + 3 <= n) m = fminf(m, fabsf(v.w));
        }
    // fminf drops NaNs: a NaN in x0 makes every dividend it enters NaN, which any division mode returns as NaN
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) m = fminf(m, __shfl_xor(m, off, 16));
    // (the 16 lanes of a tile agree on i0 < i1; lane 0 of the group holds the tile's first column)
    // A tile only partly inside [row_lo, row_hi) (a slab's ghost zone ends inside it) reads 0 = "divide the long way": rows
    // of it that become valid later -- an exchange inside the solve -- were never looked at, and a wave working on them must
    // not take the two-term path on the strength of the rows that were.
    const bool whole = row_lo <= 1 + tr * kTileRows && row_hi >= min(n + 1, 1 + (tr + 1) * kTileRows);
    if ((lane & 15) == 0 && j <= n) out[(size_t)tr * tile_pitch + (j - 1) / kTileCols] = whole ? __float_as_uint(m) : 0u;
}

// ---------------------------------------------------------------------------
// a5  advect (FluidSequential.c:107-141): the velocity streams in coalesced, the
// four bilinear taps of d0 per cell are gathered (served by L2 / Infinity Cache).
// ---------------------------------------------------------------------------
// What bounds these kernels is the gather, not HBM: a wave-wide load whose lanes are 16 bytes apart touches eight
// 128-byte lines to use a quarter of each, and a cell needs four taps per advected field.  So a wave works on 256
// consecutive cells of a row in four rounds of 64 -- lane l takes cells l, 64+l, 128+l, 192+l -- which makes every
// gather of every round lane-contiguous: the two horizontally adjacent taps of a cell are ONE 8-byte load (they are
// contiguous in memory; 4-byte aligned, which global loads allow), so a round's taps touch three lines per instruction,
// not eight.  The wave's own cells -- velocities in, results out -- travel as one 16-byte access per lane and change
// into that order through LDS (wave_cells_in / wave_cells_out; k_gradient_advect keeps one dword per lane and round,
// see there).  Four independent back-traces per thread keep 8 or 16 loads in flight.  Per cell the arithmetic is the
// reference's, in its order.
// Offsets are BYTE offsets of type IDX: unsigned for fields below 2^32 bytes (every size up to ~32700^2 fp32), so an
// access is a scalar base plus a 32-bit vector offset and no 64-bit vector arithmetic is spent on addresses; size_t beyond.
template <typename S, typename IDX>
__device__ __forceinline__ const S* at(const S* base, IDX byte_off) { return reinterpret_cast<const S*>(reinterpret_cast<const char*>(base) + byte_off); }
template <typename S, typename IDX>
__device__ __forceinline__ S* at(S* base, IDX byte_off) { return reinterpret_cast<S*>(reinterpret_cast<char*>(base) + byte_off); }

template <typename IDX>
struct AdvectTap {
    IDX o;                // byte offset of the top-left tap
    float s0, s1, t0, t1;
};

// P: row pitch in bytes, E: element size in bytes
template <typename IDX>
__device__ __forceinline__ AdvectTap<IDX> advect_trace(int j, int i, float uu, float vv, float dt0, int n, IDX P, IDX E)
{
    float px = (float)j - dt0 * uu;
    float py = (float)i - dt0 * vv;
    const float hi = (float)n + 0.5f;
    // FluidSequential.c:117-127: if (x < 0.5) x = 0.5; if (x > N + 0.5) x = N + 0.5 -- as max / min (the same value
    // for every x that is not NaN; a NaN velocity, which sends the reference's (int) cast into undefined behaviour,
    // lands on the lower bound here)
    px = __builtin_fminf(__builtin_fmaxf(px, 0.5f), hi);
    py = __builtin_fminf(__builtin_fmaxf(py, 0.5f), hi);
    const int j0 = (int)px, i0 = (int)py;
    AdvectTap<IDX> t;
    t.s1 = px - (float)j0;
    t.s0 = 1.0f - t.s1;
    t.t1 = py - (float)i0;
    t.t0 = 1.0f - t.t1;
    t.o = (IDX)i0 * P + (IDX)(XOFF + j0) * E;
    return t;
}

// two horizontally adjacent cells in one access
struct __attribute__((packed, aligned(4))) FloatPair { float a, b; };
__device__ __forceinline__ void ld_pair(const float* p, float& a, float& b)
{
    const FloatPair v = *reinterpret_cast<const FloatPair*>(p);
    a = v.a;
    b = v.b;
}
__device__ __forceinline__ void ld_pair(const half_t* p, float& a, float& b)
{
    a = ld1(p);
    b = ld1(p + 1);
}

template <typename S, typename IDX>
__device__ __forceinline__ float advect_sample(const S* __restrict__ d0, IDX P, const AdvectTap<IDX>& t)
{
    float q00, q10, q01, q11;            // q[col][row]
    ld_pair(at(d0, t.o), q00, q10);
    ld_pair(at(d0, t.o + P), q01, q11);
    const float a = t.t0 * q00 + t.t1 * q01;
    const float e = t.t0 * q10 + t.t1 * q11;
    return t.s0 * a + t.s1 * e;
}

constexpr int kAdvectRounds = 4;         // cells per thread, 64 apart
// first column of a thread's round-0 cell: a block covers 256 threads x 4 rounds = 1024 consecutive columns
__device__ __forceinline__ int advect_col0() { return 1 + (int)blockIdx.x * 1024 + (int)(threadIdx.x >> 6) * 256 + (int)(threadIdx.x & 63); }
// does this block hold a cell next to a wall (wave-uniform)?  Only those run the ghost-cell code at all.
__device__ __forceinline__ bool advect_wall_block(int i, int n) { return i == 1 || i == n || blockIdx.x == 0 || blockIdx.x == gridDim.x - 1; }

// A wave's own 256 cells of a row, in and out, as ONE 16-byte access per lane instead of four 4-byte ones: the memory
// side wants lane l to hold cells 4l .. 4l+3, the gathers want it to hold cells l, 64+l, 128+l, 192+l (above), and a
// wave-private 1 KiB tile of LDS turns one into the other.  What these kernels wait for is the number of vector-memory
// instructions the texture addresser has to take apart, about 21 cycles each per CU whatever their width: the
// velocity loads and the result stores were half of them.  LDS executes a wave's instructions in order, so no barrier.
// `p` points at the wave's first cell (16-byte aligned: column 1 sits on a 256-byte line and waves start 256 cells apart);
// `cells` of the 256 exist (the rest of the row's last vector is pad; lanes past it repeat the last vector).
template <typename S>
__device__ __forceinline__ void wave_cells_in(const S* __restrict__ p, int cells, float* __restrict__ tile, float (&out)[4])
{
    const int lane = threadIdx.x & 63;
    const int vec = min(lane, (cells - 1) >> 2);
    *reinterpret_cast<float4*>(tile + 4 * lane) = ld4(p + 4 * vec);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 0; k < 4; ++k) out[k] = tile[lane + 64 * k];
    __builtin_amdgcn_wave_barrier();
}
template <typename S>
__device__ __forceinline__ void wave_cells_out(S* __restrict__ p, int cells, float* __restrict__ tile, const float (&val)[4])
{
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int k = 0; k < 4; ++k) tile[lane + 64 * k] = val[k];
    __builtin_amdgcn_wave_barrier();
    const float4 y = *reinterpret_cast<const float4*>(tile + 4 * lane);
    __builtin_amdgcn_wave_barrier();
    const int left = cells - 4 * lane;                   // cells of this lane's vector that exist
    if (left >= 4) {
        st4(p + 4 * lane, y);
    } else if (left > 0) {
        st1(p + 4 * lane, y.x);
        if (left > 1) st1(p + 4 * lane + 1, y.y);
        if (left > 2) st1(p + 4 * lane + 2, y.z);
    }
}

// (Round 3 built the version DESIGN.md had sketched -- the taps staged in LDS: a block takes a tile of 8 or 16 rows x 256
// columns, computes all its back-traces, reduces their bounding box, loads the box with coalesced 16-byte accesses when it
// fits 12-24 rows x 272 columns per field and reads every tap from LDS, else gathers as here.  Bit-identical, 1.6 instead of 5
// vector-memory instructions per cell -- and slower wherever the back-traces are short or smooth, which is the workload:
// 8192^2, one field, smooth velocity: 196-210 us against 178-196 here (16-row tiles: 243-264); two fields inside a step: 329 us
// against 275 (decaying data), 546 against 555 (ordinary data).  It only won on large AND noisy velocities (13-cell
// back-traces of white noise: 366 against 490 us).  Two block-wide barriers, the staging burst and 34-60 KB of LDS per block
// (2-4 blocks per CU) cost more than the gathers they replace; the single-field kernel here already moves its 1074 MB at
// 6.0 TB/s.  Not kept.)
// (A block that walks eight rows and loads the next row's velocity ahead of the current row's taps was measured slower:
// 324 against 294 us at 8192^2.  The texture addresser is busy 94 % of k_advect2's time -- rocprofv3 TA_BUSY_avr --
// so what these kernels wait for is the address path of their gathers, neither HBM nor latency.)
template <typename S, typename IDX>
__global__ __launch_bounds__(256) void k_advect(S* __restrict__ d, const S* __restrict__ d0, const S* __restrict__ u,
                                                const S* __restrict__ v, int pitch, int n, int row_lo, int row_hi,
                                                float dt0, int b)
{
    __shared__ __attribute__((aligned(16))) float tiles[4][2][256];
    const int j0 = advect_col0();
    const int i = row_lo + blockIdx.y;
    if (i >= row_hi) return;
    const IDX E = (IDX)sizeof(S), P = (IDX)pitch * E;
    const IDX r = (IDX)i * P + (IDX)XOFF * E;
    const bool wall = advect_wall_block(i, n);
    const int wave = threadIdx.x >> 6;
    const int c0 = j0 - (int)(threadIdx.x & 63);         // the wave's first column (wave-uniform)
    const int cells = min(256, n + 1 - c0);              // how many of its 256 columns exist
    if (cells <= 0) return;
    float uu[kAdvectRounds], vv[kAdvectRounds], val[kAdvectRounds];
    wave_cells_in(at(u, r + (IDX)c0 * E), cells, tiles[wave][0], uu);
    wave_cells_in(at(v, r + (IDX)c0 * E), cells, tiles[wave][1], vv);
#pragma unroll
    for (int k = 0; k < kAdvectRounds; ++k) val[k] = advect_sample(d0, P, advect_trace<IDX>(min(j0 + 64 * k, n), i, uu[k], vv[k], dt0, n, P, E));
    wave_cells_out(at(d, r + (IDX)c0 * E), cells, tiles[wave][0], val);
    if (wall) {
#pragma unroll
        for (int k = 0; k < kAdvectRounds; ++k) {
            const int j = j0 + 64 * k;
            if (j <= n) emit_ghosts(d, (size_t)pitch, n, b, j, i, val[k]);
        }
    }
}

// Two advections along the same velocity field in one pass (vel_step advects u and v, both along
// (u0, v0), FluidSequential.c:213-214): the velocity is read and the back-trace computed once, and
// the eight taps of the two sources sit at the same offsets.
template <typename S, typename IDX>
__global__ __launch_bounds__(256) void k_advect2(S* __restrict__ da, const S* __restrict__ d0a, int ba, S* __restrict__ db,
                                                 const S* __restrict__ d0b, int bb, const S* __restrict__ u,
                                                 const S* __restrict__ v, int pitch, int n, int row_lo, int row_hi, float dt0)
{
    __shared__ __attribute__((aligned(16))) float tiles[4][2][256];
    const int j0 = advect_col0();
    const int i = row_lo + blockIdx.y;
    if (i >= row_hi) return;
    const IDX E = (IDX)sizeof(S), P = (IDX)pitch * E;
    const IDX r = (IDX)i * P + (IDX)XOFF * E;
    const bool wall = advect_wall_block(i, n);
    const int wave = threadIdx.x >> 6;
    const int c0 = j0 - (int)(threadIdx.x & 63);         // the wave's first column (wave-uniform)
    const int cells = min(256, n + 1 - c0);              // how many of its 256 columns exist
    if (cells <= 0) return;
    float uu[kAdvectRounds], vv[kAdvectRounds], va[kAdvectRounds], vb[kAdvectRounds];
    wave_cells_in(at(u, r + (IDX)c0 * E), cells, tiles[wave][0], uu);
    wave_cells_in(at(v, r + (IDX)c0 * E), cells, tiles[wave][1], vv);
#pragma unroll
    for (int k = 0; k < kAdvectRounds; ++k) {
        const AdvectTap<IDX> t = advect_trace<IDX>(min(j0 + 64 * k, n), i, uu[k], vv[k], dt0, n, P, E);
        va[k] = advect_sample(d0a, P, t);
        vb[k] = advect_sample(d0b, P, t);
    }
    wave_cells_out(at(da, r + (IDX)c0 * E), cells, tiles[wave][0], va);
    wave_cells_out(at(db, r + (IDX)c0 * E), cells, tiles[wave][1], vb);
    if (wall) {
#pragma unroll
        for (int k = 0; k < kAdvectRounds; ++k) {
            const int j = j0 + 64 * k;
            if (j <= n) {
                emit_ghosts(da, (size_t)pitch, n, ba, j, i, va[k]);
                emit_ghosts(db, (size_t)pitch, n, bb, j, i, vb[k]);
            }
        }
    }
}

// ---------------------------------------------------------------------------
// a6  divergence + pressure clear (FluidSequential.c:143-158).
// div = (-0.5f*h) * (((uR - uL) + vD) - vU), p = 0, set_bnd(0) on both: the
// ghosts of p are 0 as well, so p is zeroed on whole rows incl. ghost rows.
// ---------------------------------------------------------------------------
template <typename S>
__global__ __launch_bounds__(256) void k_divergence(const S* __restrict__ u, const S* __restrict__ v, S* __restrict__ p,
                                                    S* __restrict__ div, int pitch, int n, int row_lo, int row_hi,
                                                    float h, int write_p, float pscale)
{
    const int j = 1 + blockIdx.x * 256 + threadIdx.x;
    const int i = row_lo + blockIdx.y;
    if (j > n || i >= row_hi) return;
    const size_t P = (size_t)pitch;
    const size_t c = (size_t)i * P + XOFF + j;
    const float scale = (-0.5f * h) * pscale;            // pscale: a power of two (1 unless the solver keeps a scaled pressure, fp16 storage)
    float g = ld1(u + c + 1) - ld1(u + c - 1);
    g = g + ld1(v + c + P);
    g = g - ld1(v + c - P);
    const float val = scale * g;
    st1(div + c, val);
    emit_ghosts(div, P, n, 0, j, i, val);
    if (write_p) {          // the solver usually just marks p as "all zero" instead
        st1(p + c, 0.0f);
        emit_ghosts(p, P, n, 0, j, i, 0.0f);
    }
}

// 64-lane butterfly maximum (the reductions further down, and k_subtract_gradient_max)
__device__ __forceinline__ float wave_max(float m)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    return m;
}

// ---------------------------------------------------------------------------
// a7  pressure-gradient subtraction (FluidSequential.c:161-173):
// u -= (0.5f*(pR-pL))/h ; v -= (0.5f*(pD-pU))/h ; set_bnd(1,u), set_bnd(2,v).
// ---------------------------------------------------------------------------
template <typename S>
__global__ __launch_bounds__(256) void k_subtract_gradient(S* __restrict__ u, S* __restrict__ v, const S* __restrict__ p,
                                                           int pitch, int n, int row_lo, int row_hi, float h, float pinv)
{
    const int j = 1 + blockIdx.x * 256 + threadIdx.x;
    const int i = row_lo + blockIdx.y;
    if (j > n || i >= row_hi) return;
    const size_t P = (size_t)pitch;
    const size_t c = (size_t)i * P + XOFF + j;
    const float gx = 0.5f * (ld1(p + c + 1) - ld1(p + c - 1));
    const float gy = 0.5f * (ld1(p + c + P) - ld1(p + c - P));
    // pinv: 1 / (the power of two the pressure field is scaled by): (2^k x) / h * 2^-k is x / h, bit for bit
    const float nu = ld1(u + c) - (gx / h) * pinv;
    const float nv = ld1(v + c) - (gy / h) * pinv;
    st1(u + c, nu);
    st1(v + c, nv);
    emit_ghosts(u, P, n, 1, j, i, nu);
    emit_ghosts(v, P, n, 2, j, i, nv);
}

// The same with max(|u|, |v|) of what it stores (k_absmax2's result over the same rows, interior cells) reduced on the
// way: on row slabs the advection that follows needs that bound on the host before it can start, and a reduction pass of
// its own costs as much as this whole kernel.  Every block leaves its maximum in partials[]; k_max_partials (one block)
// folds them into the result word.  No atomics (thousands of them on one word serialise at ~12 ns each) and no memset.
template <typename S>
__global__ __launch_bounds__(256) void k_subtract_gradient_max(S* __restrict__ u, S* __restrict__ v, const S* __restrict__ p,
                                                               int pitch, int n, int row_lo, int row_hi, float h,
                                                               float* __restrict__ partials, float pinv)
{
    const int j = 1 + blockIdx.x * 256 + threadIdx.x;
    const size_t P = (size_t)pitch;
    float m = 0.0f;
    if (j <= n)
        for (int i = row_lo + blockIdx.y; i < row_hi; i += gridDim.y) {
            const size_t c = (size_t)i * P + XOFF + j;
            const float gx = 0.5f * (ld1(p + c + 1) - ld1(p + c - 1));
            const float gy = 0.5f * (ld1(p + c + P) - ld1(p + c - P));
            const float nu = ld1(u + c) - (gx / h) * pinv;
            const float nv = ld1(v + c) - (gy / h) * pinv;
            st1(u + c, nu);
            st1(v + c, nv);
            emit_ghosts(u, P, n, 1, j, i, nu);
            emit_ghosts(v, P, n, 2, j, i, nv);
            m = fmaxf(m, fmaxf(fabsf(as_stored<S>(nu)), fabsf(as_stored<S>(nv))));
        }
    __shared__ float part[4];
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) partials[blockIdx.y * gridDim.x + blockIdx.x] = fmaxf(fmaxf(part[0], part[1]), fmaxf(part[2], part[3]));
}

__global__ __launch_bounds__(256) void k_max_partials(const float* __restrict__ partials, int count, unsigned int* __restrict__ result)
{
    float m = 0.0f;
    for (int k = threadIdx.x; k < count; k += 256) m = fmaxf(m, partials[k]);
    __shared__ float part[4];
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) *result = __float_as_uint(fmaxf(fmaxf(part[0], part[1]), fmaxf(part[2], part[3])));
}

// The last two operators of a step in one pass (FluidSequential.c:240 + :185): the gradient subtraction
// of the second projection, then the density advection along the velocity it has just produced.  The
// back-trace of cell (i, j) needs u and v at (i, j) only -- the values this thread holds in registers
// (as stored: rounded to the storage type first) -- so the advection need not read the two fields back.
// Four cells per thread (64 apart, as in k_advect), per cell the arithmetic of k_subtract_gradient and k_advect.
// (Its own cells move as one dword per lane and round: the 16-byte accesses through LDS that pay in k_advect2 -- 44 -> 18
// memory instructions per wave and row here -- were built and measured for this kernel too, and gained nothing.)
template <typename S, typename IDX>
__global__ __launch_bounds__(256) void k_gradient_advect(S* __restrict__ u, S* __restrict__ v, const S* __restrict__ p,
                                                         S* __restrict__ d, const S* __restrict__ d0, int pitch, int n,
                                                         int row_lo, int row_hi, float h, float dt0, int b, float pinv)
{
    const int j0 = advect_col0();
    const int i = row_lo + blockIdx.y;
    if (i >= row_hi) return;
    const IDX E = (IDX)sizeof(S), P = (IDX)pitch * E;
    const IDX r = (IDX)i * P + (IDX)XOFF * E;
    const bool wall = advect_wall_block(i, n);
    float nu[kAdvectRounds], nv[kAdvectRounds], val[kAdvectRounds];
#pragma unroll
    for (int k = 0; k < kAdvectRounds; ++k) {            // lane-contiguous dwords; the row's left/right neighbours are L1 hits
        const IDX c = r + (IDX)min(j0 + 64 * k, n) * E;
        const float gx = 0.5f * (ld1(at(p, c + E)) - ld1(at(p, c - E)));
        const float gy = 0.5f * (ld1(at(p, c + P)) - ld1(at(p, c - P)));
        nu[k] = ld1(at(u, c)) - (gx / h) * pinv;
        nv[k] = ld1(at(v, c)) - (gy / h) * pinv;
    }
#pragma unroll
    for (int k = 0; k < kAdvectRounds; ++k) {
        const int j = j0 + 64 * k;
        if (j <= n) {
            st1(at(u, r + (IDX)j * E), nu[k]);
            st1(at(v, r + (IDX)j * E), nv[k]);
            if (wall) {
                emit_ghosts(u, (size_t)pitch, n, 1, j, i, nu[k]);
                emit_ghosts(v, (size_t)pitch, n, 2, j, i, nv[k]);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < kAdvectRounds; ++k)
        val[k] = advect_sample(d0, P, advect_trace<IDX>(min(j0 + 64 * k, n), i, as_stored<S>(nu[k]), as_stored<S>(nv[k]), dt0, n, P, E));
#pragma unroll
    for (int k = 0; k < kAdvectRounds; ++k) {
        const int j = j0 + 64 * k;
        if (j <= n) {
            st1(at(d, r + (IDX)j * E), val[k]);
            if (wall) emit_ghosts(d, (size_t)pitch, n, b, j, i, val[k]);
        }
    }
}

// ---------------------------------------------------------------------------
// wavefront reductions (diagnostics + the advect halo bound of the slab path).
// 64-lane __shfl_xor butterflies, one atomic per wave.  Neither feeds back
// into the fields, so they cannot change results.
//   k_absmax2 : max(|u|,|v|) over interior cells of rows [row_lo,row_hi)
//               (non-negative floats order like their bit patterns => atomicMax
//               on the uint view is exact and order independent).
//   k_residual: max |beta*x - alpha*(L+R+U+D) - x0| over the same cells.
// ---------------------------------------------------------------------------
// wave maxima -> block maximum (LDS) -> ONE atomic per block: contended atomics on a single word
// serialise at ~12 ns each, so one per wave from thousands of waves costs more than the reduction.
__device__ __forceinline__ void block_max_to(unsigned int* __restrict__ result, float m)
{
    __shared__ float part[4];
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(part[0], part[1]), fmaxf(part[2], part[3]));
        atomicMax(result, __float_as_uint(m));
    }
}

// one workgroup per group of rows, whole float4 vectors (columns beyond 1..n masked out)
template <typename S>
__global__ __launch_bounds__(256) void k_absmax2(const S* __restrict__ u, const S* __restrict__ v, int pitch, int n,
                                                 int row_lo, int row_hi, unsigned int* __restrict__ result)
{
    const size_t P = (size_t)pitch;
    const int nvec = (n + 3) >> 2;
    float m = 0.0f;
    for (int i = row_lo + blockIdx.x; i < row_hi; i += gridDim.x)
        for (int k = threadIdx.x; k < nvec; k += 256) {
            const size_t c = (size_t)i * P + XOFF + 1 + 4 * (size_t)k;
            const float4 a = ld4(u + c), b = ld4(v + c);
            const int last = n - (1 + 4 * k);            // components 0..last are columns <= n
            float t = fmaxf(fabsf(a.x), fabsf(b.x));
            if (last >= 1) t = fmaxf(t, fmaxf(fabsf(a.y), fabsf(b.y)));
            if (last >= 2) t = fmaxf(t, fmaxf(fabsf(a.z), fabsf(b.z)));
            if (last >= 3) t = fmaxf(t, fmaxf(fabsf(a.w), fabsf(b.w)));
            m = fmaxf(m, t);
        }
    block_max_to(result, m);
}

template <typename S>
__global__ __launch_bounds__(256) void k_residual(const S* __restrict__ x, const S* __restrict__ x0, int pitch, int n,
                                                  int row_lo, int row_hi, float alpha, float beta,
                                                  unsigned int* __restrict__ result)
{
    const size_t P = (size_t)pitch;
    float m = 0.0f;
    for (int i = row_lo + blockIdx.x; i < row_hi; i += gridDim.x)
        for (int j = 1 + threadIdx.x; j <= n; j += 256) {
            const size_t c = (size_t)i * P + XOFF + j;
            const float nb = ld1(x + c - 1) + ld1(x + c + 1) + ld1(x + c - P) + ld1(x + c + P);
            m = fmaxf(m, fabsf(beta * ld1(x + c) - alpha * nb - ld1(x0 + c)));
        }
    block_max_to(result, m);
}

// ---------------------------------------------------------------------------
// launch wrappers (host).  Shapes are validated by the caller (fluid_solver).
// `st` selects the field storage type the untyped pointers refer to.
// ---------------------------------------------------------------------------
static inline unsigned cdiv(unsigned a, unsigned b) { return (a + b - 1) / b; }

// element offsets into a field fit 32 bits (with a byte offset below 2^32 for the hardware's address add)
static inline bool narrow_index(int st, int pitch, int n) { return (size_t)(n + 2) * (size_t)pitch * storage_bytes(st) < (1ull << 32); }

#define FLUID_BY_STORAGE(st, CALL)            \
    do {                                      \
        if ((st) == STORAGE_F16) {            \
            using S = half_t;                 \
            CALL;                             \
        } else {                              \
            using S = float;                  \
            CALL;                             \
        }                                     \
    } while (0)

void launch_set_bnd(hipStream_t s, int st, void* f, int pitch, int n, int b)
{
    FLUID_BY_STORAGE(st, hipLaunchKernelGGL(k_set_bnd<S>, dim3(cdiv(n, 256)), dim3(256), 0, s, (S*)f, pitch, n, b));
}

// src == nullptr: the source is known to be all +0 (dt is then the pre-multiplied increment dt*0)
void launch_add_source(hipStream_t s, int st, void* x, const void* src, int pitch, int row_lo, int row_hi, float dt)
{
    const size_t total = (size_t)(row_hi - row_lo) * (pitch >> 2);
    const unsigned blocks = (unsigned)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    if (!src) {
        FLUID_BY_STORAGE(st, hipLaunchKernelGGL(k_add_zero_source<S>, dim3(blocks ? blocks : 1), dim3(256), 0, s, (S*)x,
                                                pitch, row_lo, row_hi, dt));
        return;
    }
    FLUID_BY_STORAGE(st, hipLaunchKernelGGL(k_add_source<S>, dim3(blocks ? blocks : 1), dim3(256), 0, s, (S*)x,
                                            (const S*)src, pitch, row_lo, row_hi, dt));
}

void launch_scale(hipStream_t s, int st, void* x, int pitch, int row_lo, int row_hi, float factor)
{
    const size_t total = (size_t)(row_hi - row_lo) * (pitch >> 2);
    const unsigned blocks = (unsigned)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    FLUID_BY_STORAGE(st, hipLaunchKernelGGL(k_scale<S>, dim3(blocks ? blocks : 1), dim3(256), 0, s, (S*)x, pitch, row_lo, row_hi, factor));
}

void launch_jacobi(hipStream_t s, int st, int variant, const void* x, const void* x0, void* out, int pitch, int n,
                   int row_lo, int row_hi, float alpha, float beta, int b)
{
    const int rows = row_hi - row_lo;
    if (rows <= 0) return;
    switch (variant) {
    case JACOBI_NAIVE:
        FLUID_BY_STORAGE(st, hipLaunchKernelGGL(k_jacobi_naive<S>, dim3(cdiv(n, 64), cdiv(rows, 4)), dim3(256), 0, s,
                                                (const S*)x, (const S*)x0, (S*)out, pitch, n, row_lo, row_hi, alpha,
                                                beta, b));
        break;
    case JACOBI_LDS:
        FLUID_BY_STORAGE(st, hipLaunchKernelGGL(k_jacobi_lds<S>, dim3(cdiv(n, LT_X), cdiv(rows, LT_Y)), dim3(256), 0, s,
                                                (const S*)x, (const S*)x0, (S*)out, pitch, n, row_lo, row_hi, alpha,
                                                beta, b));
        break;
    default: {
        constexpr int RB = 8;
        const unsigned nvec = (n + 3) / 4;
        FLUID_BY_STORAGE(st, hipLaunchKernelGGL((k_jacobi_stream<RB, S>), dim3(cdiv(nvec, 256), cdiv(rows, RB)), dim3(256),
                                                0, s, (const S*)x, (const S*)x0, (S*)out, pitch, n, row_lo, row_hi,
                                                alpha, beta, b));
    }
    }
}

// T in {8,4,2} sweeps per launch, nv in {2,4} columns per lane; batch.count solves per launch.
// divmode 0: beta; 2: beta unused, yd = RN64(1/beta); 4: beta = exact reciprocal of a power of two and alpha == 1;
// 3: hi, lo = the two-term reciprocal where the tiles of |x0| minima allow it, yd elsewhere;
// 5: beta = RN32(1/beta), hi = beta * 2^24, lo = -(RN32(1/beta) * 2^-24), yd for the second pass of a wave that met inf / NaN.
void launch_jacobi_tb(hipStream_t s, int st, int T, int divmode, int nv, const TbBatch& batch, int pitch, int n, int row_lo,
                      int row_hi, int rb, int rb_edge, bool divsrc, bool addsrc, int hole_lo, int hole_hi)
{
    const int rows = row_hi - row_lo;
    if (rows <= 0 || batch.count <= 0) return;
    if (hole_lo < row_lo) hole_lo = row_lo;
    if (hole_hi > row_hi) hole_hi = row_hi;
    const bool hole = hole_lo < hole_hi;
    if (hole && hole_lo == row_lo && hole_hi == row_hi) return;
    if (T == 16 || T == 12) nv = 2;
    const int HL = (T + nv - 1) / nv, VS = 64 - 2 * HL;
    const unsigned nvec = (n + nv - 1) / nv;
    rb_edge = rb_edge < 1 ? rb : (rb_edge > rb ? rb : rb_edge);
    // edge windows: those whose 64 lanes include vector -1 (window 0) or vector kg = n / nv (the last one or two)
    const int nwin = (int)cdiv(nvec, VS), kg = n / nv;
    int first_right = nwin - 1;
    while (first_right > 1 && kg < (first_right - 1) * VS - HL + 64) --first_right;
    if (first_right < 1) first_right = 1;                // a single window is window 0
    TbGrid g;
    g.first_right = first_right;
    g.inner_wins = first_right - 1;
    g.edge_wins = 1 + (nwin - first_right);
    // strips of a window: over the rows, or over the part above the hole and the part below it (none straddles it)
    auto strips = [&](int r) { return hole ? cdiv(hole_lo - row_lo, r) + cdiv(row_hi - hole_hi, r) : cdiv(rows, r); };
    g.hole_lo = hole ? hole_lo : 0;
    g.hole_hi = hole ? hole_hi : 0;
    g.inner_blocks = g.inner_wins * (int)cdiv(strips(rb), 4);
    g.edge_blocks = g.edge_wins * (int)cdiv(strips(rb_edge), 4);
    const dim3 grid(8 * cdiv(g.inner_blocks + g.edge_blocks, 8), 1, batch.count), block(256);
#define FLUID_TB2(TT, DD, NN) \
    FLUID_BY_STORAGE(st, hipLaunchKernelGGL((k_jacobi_tb<TT, DD, NN, S>), grid, block, 0, s, batch, pitch, n, row_lo, row_hi, rb, rb_edge, g))
#define FLUID_TB1(TT, DD)            \
    if (nv == 2) FLUID_TB2(TT, DD, 2); \
    else FLUID_TB2(TT, DD, 4)
#define FLUID_TB(TT)                          \
    if (divmode == 4) { FLUID_TB1(TT, 4); }      \
    else if (divmode == 5) { FLUID_TB1(TT, 5); } \
    else if (divmode == 3) { FLUID_TB1(TT, 3); } \
    else if (divmode == 2) { FLUID_TB1(TT, 2); } \
    else { FLUID_TB1(TT, 0); }
#define FLUID_TBD(TT) \
    FLUID_BY_STORAGE(st, hipLaunchKernelGGL((k_jacobi_tb<TT, 4, 2, S, true>), grid, block, 0, s, batch, pitch, n, row_lo, row_hi, rb, rb_edge, g))
#define FLUID_TBA2(TT, DD) \
    FLUID_BY_STORAGE(st, hipLaunchKernelGGL((k_jacobi_tb<TT, DD, 2, S, false, true>), grid, block, 0, s, batch, pitch, n, row_lo, row_hi, rb, rb_edge, g))
#define FLUID_TBA(TT)                       \
    if (divmode == 5) { FLUID_TBA2(TT, 5); }      \
    else if (divmode == 2) { FLUID_TBA2(TT, 2); } \
    else { FLUID_TBA2(TT, 0); }
    if (addsrc) {                                        // first launch of a diffusion: right-hand side = x0 + dt * x, stored out of place
        if (T == 16) { FLUID_TBA(16) }
        else if (T == 12) { FLUID_TBA(12) }
        else { FLUID_TBA(8) }
    }
    else if (divsrc) {                                   // first launch of a pressure solve, right-hand side computed from (u, v)
        if (T == 16) { FLUID_TBD(16); }
        else if (T == 12) { FLUID_TBD(12); }
        else { FLUID_TBD(8); }
    }
    else if (T == 16) {                                  // 2-column lanes only (4-column ones would need > 256 registers)
        if (divmode == 4) { FLUID_TB2(16, 4, 2); }
        else if (divmode == 5) { FLUID_TB2(16, 5, 2); }
        else if (divmode == 3) { FLUID_TB2(16, 3, 2); }
        else if (divmode == 2) { FLUID_TB2(16, 2, 2); }
        else { FLUID_TB2(16, 0, 2); }
    }
    else if (T == 12) {                                  // 2-column lanes, three waves per SIMD
        if (divmode == 4) { FLUID_TB2(12, 4, 2); }
        else if (divmode == 5) { FLUID_TB2(12, 5, 2); }
        else if (divmode == 3) { FLUID_TB2(12, 3, 2); }
        else if (divmode == 2) { FLUID_TB2(12, 2, 2); }
        else { FLUID_TB2(12, 0, 2); }
    }
    else if (T == 8) { FLUID_TB(8) }
    else if (T == 4) { FLUID_TB(4) }
    else { FLUID_TB(2) }
#undef FLUID_TBA
#undef FLUID_TBA2
#undef FLUID_TBD
#undef FLUID_TB
#undef FLUID_TB1
#undef FLUID_TB2
}

// tiles of |x0| minima for division mode 3, rows [row_lo, row_hi) within 1..n+1; tb.tiles[k] holds tile_rows(n) x tile_pitch words
void launch_tile_min_abs(hipStream_t s, int st, const TileBatch& tb, int count, int pitch, int n, int row_lo, int row_hi, int tile_pitch)
{
    if (row_lo < 1) row_lo = 1;
    if (row_hi > n + 1) row_hi = n + 1;
    if (row_hi <= row_lo || count <= 0) return;
    const int tr0 = (row_lo - 1) / kTileRows, tr1 = (row_hi - 2) / kTileRows;
    const dim3 grid(cdiv(cdiv(n, 4), 256), tr1 - tr0 + 1, count);
    FLUID_BY_STORAGE(st, hipLaunchKernelGGL(k_tile_min_abs<S>, grid, dim3(256), 0, s, tb, pitch, n, row_lo, row_hi, tr0, tile_pitch));
}

// mismatches of division mode `divmode` against a/beta over all 2^32 inputs are added to *bad; kbeta / yd / hi / lo: the
// constants the mode reads (DivK)
void launch_validate_div(hipStream_t s, int divmode, float beta, float kbeta, double yd, float hi, float lo, unsigned long long* bad)
{
    DivK k;
    k.beta = kbeta;
    k.yd = yd;
    k.hi = hi;
    k.lo = lo;
    if (divmode == 4) hipLaunchKernelGGL((k_validate_div<4>), dim3(8192), dim3(256), 0, s, beta, k, bad);
    else if (divmode == 5) hipLaunchKernelGGL((k_validate_div<5>), dim3(8192), dim3(256), 0, s, beta, k, bad);
    else if (divmode == 3) hipLaunchKernelGGL((k_validate_div<3>), dim3(8192), dim3(256), 0, s, beta, k, bad);
    else hipLaunchKernelGGL((k_validate_div<2>), dim3(8192), dim3(256), 0, s, beta, k, bad);
}

void launch_advect(hipStream_t s, int st, void* d, const void* d0, const void* u, const void* v, int pitch, int n,
                   int row_lo, int row_hi, float dt0, int b)
{
    if (row_hi <= row_lo) return;
    const dim3 grid(cdiv(n, 1024), row_hi - row_lo);
    if (narrow_index(st, pitch, n))
        FLUID_BY_STORAGE(st, hipLaunchKernelGGL((k_advect<S, unsigned>), grid, dim3(256), 0, s, (S*)d, (const S*)d0, (const S*)u,
                                                (const S*)v, pitch, n, row_lo, row_hi, dt0, b));
    else
        FLUID_BY_STORAGE(st, hipLaunchKernelGGL((k_advect<S, size_t>), grid, dim3(256), 0, s, (S*)d, (const S*)d0, (const S*)u,
                                                (const S*)v, pitch, n, row_lo, row_hi, dt0, b));
}

void launch_advect2(hipStream_t s, int st, void* da, const void* d0a, int ba, void* db, const void* d0b, int bb, const void* u,
                    const void* v, int pitch, int n, int row_lo, int row_hi, float dt0)
{
    if (row_hi <= row_lo) return;
    const dim3 grid(cdiv(n, 1024), row_hi - row_lo);
    if (narrow_index(st, pitch, n))
        FLUID_BY_STORAGE(st, hipLaunchKernelGGL((k_advect2<S, unsigned>), grid, dim3(256), 0, s, (S*)da, (const S*)d0a, ba, (S*)db,
                                                (const S*)d0b, bb, (const S*)u, (const S*)v, pitch, n, row_lo, row_hi, dt0));
    else
        FLUID_BY_STORAGE(st, hipLaunchKernelGGL((k_advect2<S, size_t>), grid, dim3(256), 0, s, (S*)da, (const S*)d0a, ba, (S*)db,
                                                (const S*)d0b, bb, (const S*)u, (const S*)v, pitch, n, row_lo, row_hi, dt0));
}

void launch_divergence(hipStream_t s, int st, const void* u, const void* v, void* p, void* div, int pitch, int n,
                       int row_lo, int row_hi, float h, int write_p, float pscale)
{
    if (row_hi <= row_lo) return;
    FLUID_BY_STORAGE(st, hipLaunchKernelGGL(k_divergence<S>, dim3(cdiv(n, 256), row_hi - row_lo), dim3(256), 0, s,
                                            (const S*)u, (const S*)v, (S*)p, (S*)div, pitch, n, row_lo, row_hi, h,
                                            write_p, pscale));
}

void launch_subtract_gradient(hipStream_t s, int st, void* u, void* v, const void* p, int pitch, int n, int row_lo,
                              int row_hi, float h, float* partials, unsigned int* max_out, float pinv)
{
    if (row_hi <= row_lo) return;
    if (max_out) {
        const unsigned per_col = cdiv(n, 256), rows = (unsigned)(row_hi - row_lo);
        const unsigned gy = std::max(1u, std::min(rows, (unsigned)kMaxPartials / per_col));     // a few rows per block
        FLUID_BY_STORAGE(st, hipLaunchKernelGGL(k_subtract_gradient_max<S>, dim3(per_col, gy), dim3(256), 0, s, (S*)u, (S*)v,
                                                (const S*)p, pitch, n, row_lo, row_hi, h, partials, pinv));
        hipLaunchKernelGGL(k_max_partials, dim3(1), dim3(256), 0, s, partials, (int)(per_col * gy), max_out);
        return;
    }
    FLUID_BY_STORAGE(st, hipLaunchKernelGGL(k_subtract_gradient<S>, dim3(cdiv(n, 256), row_hi - row_lo), dim3(256), 0, s,
                                            (S*)u, (S*)v, (const S*)p, pitch, n, row_lo, row_hi, h, pinv));
}

void launch_gradient_advect(hipStream_t s, int st, void* u, void* v, const void* p, void* d, const void* d0, int pitch, int n,
                            int row_lo, int row_hi, float h, float dt0, int b, float pinv)
{
    if (row_hi <= row_lo) return;
    const dim3 grid(cdiv(n, 1024), row_hi - row_lo);
    if (narrow_index(st, pitch, n))
        FLUID_BY_STORAGE(st, hipLaunchKernelGGL((k_gradient_advect<S, unsigned>), grid, dim3(256), 0, s, (S*)u, (S*)v, (const S*)p,
                                                (S*)d, (const S*)d0, pitch, n, row_lo, row_hi, h, dt0, b, pinv));
    else
        FLUID_BY_STORAGE(st, hipLaunchKernelGGL((k_gradient_advect<S, size_t>), grid, dim3(256), 0, s, (S*)u, (S*)v, (const S*)p,
                                                (S*)d, (const S*)d0, pitch, n, row_lo, row_hi, h, dt0, b, pinv));
}

void launch_absmax2(hipStream_t s, int st, const void* u, const void* v, int pitch, int n, int row_lo, int row_hi,
                    unsigned int* result)
{
    if (row_hi <= row_lo) return;
    const unsigned blocks = (unsigned)(row_hi - row_lo) < 1024u ? (unsigned)(row_hi - row_lo) : 1024u;
    FLUID_BY_STORAGE(st, hipLaunchKernelGGL(k_absmax2<S>, dim3(blocks), dim3(256), 0, s, (const S*)u, (const S*)v, pitch,
                                            n, row_lo, row_hi, result));
}

void launch_residual(hipStream_t s, int st, const void* x, const void* x0, int pitch, int n, int row_lo, int row_hi,
                     float alpha, float beta, unsigned int* result)
{
    if (row_hi <= row_lo) return;
    const unsigned blocks = (unsigned)(row_hi - row_lo) < 1024u ? (unsigned)(row_hi - row_lo) : 1024u;
    FLUID_BY_STORAGE(st, hipLaunchKernelGGL(k_residual<S>, dim3(blocks), dim3(256), 0, s, (const S*)x, (const S*)x0, pitch,
                                            n, row_lo, row_hi, alpha, beta, result));
}

}  // namespace fluid
