// fluid_kernels.h -- launch interface of the gfx950 kernels (fluid_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace fluid {

// column c of a device row lives at element index c + XOFF (column 1 => a 256-B (fp32) / 128-B (fp16) line start)
constexpr int XOFF = 63;

enum JacobiVariant { JACOBI_STREAM = 0, JACOBI_LDS = 1, JACOBI_NAIVE = 2, JACOBI_TB = 3, JACOBI_VARIANTS = 4 };

// field storage: fp32 (the reference's type; bit parity) or fp16 (fp32 arithmetic, rounded on store)
enum Storage { STORAGE_F32 = 0, STORAGE_F16 = 1 };
typedef _Float16 half_t;
inline size_t storage_bytes(int st) { return st == STORAGE_F16 ? 2 : 4; }

// pitch (elements) for interior size n: room for XOFF, ceil(n/4) 4-vectors and the
// right ghost, rounded to a multiple of 64 elements.
inline int pitch_for(int n) { return ((XOFF + 1 + 4 * ((n + 3) / 4) + 1) + 63) / 64 * 64; }

// up to three independent solves of the same shape, one per blockIdx.z of the fused Jacobi kernel
struct TbBatch {
    const void* x[3];
    const void* x0[3];
    void* out[3];
    float alpha[3], beta[3];     // beta: divisor; its exact reciprocal in division mode 4; RN32(1/beta) in mode 5
    double yd[3];                // RN64(1/beta) for division mode 2 (and the fall-backs of modes 3 and 5)
    float hi[3], lo[3];          // division mode 3: hi = RD32(1/beta), lo = RN32(1/beta - hi); mode 5: beta * 2^24, -(RN32(1/beta) * 2^-24)
    const unsigned* tiles[3];    // division mode 3: |x0| minima per tile (k_tile_min_abs), tile_pitch words per tile row
    unsigned tile_thr[3];        //   ... and the bit pattern of beta * 2^-72 they must reach
    int tile_pitch;
    void* div[3];                // divergence-sourced launch (launch_jacobi_tb divsrc): x / x0 hold u / v, the divergence is
    float div_scale;             //   written here; div_scale = -0.5f * h.  Source-adding launch (addsrc): x is the source
                                 //   field s, x0 + div_scale * s (div_scale = dt) is the right-hand side and is written here
    int b[3];
    int x_zero[3];               // first guess known to be all +0: never read
    float x0_inc[3];             // added to every x0 value as it is loaded (-0.0f: nothing pending)
    int count;
};

void launch_set_bnd(hipStream_t s, int st, void* f, int pitch, int n, int b);
void launch_add_source(hipStream_t s, int st, void* x, const void* src, int pitch, int row_lo, int row_hi, float dt);
void launch_jacobi(hipStream_t s, int st, int variant, const void* x, const void* x0, void* out, int pitch, int n,
                   int row_lo, int row_hi, float alpha, float beta, int b);
void launch_jacobi_tb(hipStream_t s, int st, int T, int divmode, int nv, const TbBatch& batch, int pitch, int n, int row_lo,
                      int row_hi, int rb, int rb_edge, bool divsrc = false, bool addsrc = false, int hole_lo = 0, int hole_hi = 0);
// the launches that exist with addsrc: 2-column lanes, 8 / 12 / 16 sweeps, division modes 0, 2 and 5
inline bool jacobi_tb_addsrc_exists(int T, int divmode, int nv) { return nv == 2 && (T == 8 || T == 12 || T == 16) && (divmode == 0 || divmode == 2 || divmode == 5); }
// tiles of kTileRows x kTileCols interior cells, tile (r, c) = rows 1 + r*kTileRows.., columns 1 + c*kTileCols..
constexpr int kTileRows = 32, kTileCols = 64;
inline int tile_rows(int n) { return (n + kTileRows - 1) / kTileRows; }
inline int tile_pitch(int n) { return (n + kTileCols - 1) / kTileCols; }
struct TileBatch {
    const void* field[3];
    unsigned* tiles[3];
};
void launch_tile_min_abs(hipStream_t s, int st, const TileBatch& tb, int count, int pitch, int n, int row_lo, int row_hi, int tile_pitch);
void launch_validate_div(hipStream_t s, int divmode, float beta, float kbeta, double yd, float hi, float lo, unsigned long long* bad);
void launch_advect(hipStream_t s, int st, void* d, const void* d0, const void* u, const void* v, int pitch, int n,
                   int row_lo, int row_hi, float dt0, int b);
void launch_advect2(hipStream_t s, int st, void* da, const void* d0a, int ba, void* db, const void* d0b, int bb, const void* u,
                    const void* v, int pitch, int n, int row_lo, int row_hi, float dt0);
// pscale: power of two the divergence is stored multiplied by (1: plain)
void launch_divergence(hipStream_t s, int st, const void* u, const void* v, void* p, void* div, int pitch, int n,
                       int row_lo, int row_hi, float h, int write_p, float pscale = 1.0f);
void launch_scale(hipStream_t s, int st, void* x, int pitch, int row_lo, int row_hi, float factor);
// max_out != nullptr: also leaves max(|u|, |v|) of the stored interior values in *max_out (the bit pattern of a
// non-negative float; what launch_absmax2 would produce for the same rows), via `partials` (kMaxPartials floats of scratch)
constexpr int kMaxPartials = 8192;
void launch_subtract_gradient(hipStream_t s, int st, void* u, void* v, const void* p, int pitch, int n, int row_lo,
                              int row_hi, float h, float* partials = nullptr, unsigned int* max_out = nullptr, float pinv = 1.0f);
void launch_gradient_advect(hipStream_t s, int st, void* u, void* v, const void* p, void* d, const void* d0, int pitch, int n,
                            int row_lo, int row_hi, float h, float dt0, int b, float pinv = 1.0f);
void launch_absmax2(hipStream_t s, int st, const void* u, const void* v, int pitch, int n, int row_lo, int row_hi,
                    unsigned int* result);
void launch_residual(hipStream_t s, int st, const void* x, const void* x0, int pitch, int n, int row_lo, int row_hi,
                     float alpha, float beta, unsigned int* result);

}  // namespace fluid
