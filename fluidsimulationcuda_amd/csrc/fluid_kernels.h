// fluid_kernels.h -- launch interface of the gfx950 kernels (fluid_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace fluid {

// column c of a device row lives at float index c + XOFF (column 1 => 256-B line start)
constexpr int XOFF = 63;

enum JacobiVariant { JACOBI_STREAM = 0, JACOBI_LDS = 1, JACOBI_NAIVE = 2, JACOBI_TB = 3, JACOBI_VARIANTS = 4 };

// pitch (floats) for interior size n: room for XOFF, ceil(n/4) float4s and the
// right ghost, rounded to a 256-byte multiple.
inline int pitch_for(int n) { return ((XOFF + 1 + 4 * ((n + 3) / 4) + 1) + 63) / 64 * 64; }

void launch_set_bnd(hipStream_t s, float* f, int pitch, int n, int b);
void launch_add_source(hipStream_t s, float* x, const float* src, int pitch, int row_lo, int row_hi, float dt);
void launch_jacobi(hipStream_t s, int variant, const float* x, const float* x0, float* out, int pitch, int n,
                   int row_lo, int row_hi, float alpha, float beta, int b);
// up to three independent solves of the same shape, one per blockIdx.z of the fused Jacobi kernel
struct TbBatch {
    const float* x[3];
    const float* x0[3];
    float* out[3];
    float alpha[3], beta[3];     // beta: divisor, or its exact reciprocal in division mode 1
    double yd[3];                // RN64(1/beta) for division mode 2
    int b[3];
    int count;
};
void launch_jacobi_tb(hipStream_t s, int T, int divmode, const TbBatch& batch, int pitch, int n, int row_lo,
                      int row_hi, int rb);
void launch_validate_div(hipStream_t s, int divmode, float beta, float arg, double yd, unsigned long long* bad);
void launch_advect(hipStream_t s, float* d, const float* d0, const float* u, const float* v, int pitch, int n,
                   int row_lo, int row_hi, float dt0, int b);
void launch_divergence(hipStream_t s, const float* u, const float* v, float* p, float* div, int pitch, int n,
                       int row_lo, int row_hi, float h);
void launch_subtract_gradient(hipStream_t s, float* u, float* v, const float* p, int pitch, int n, int row_lo,
                              int row_hi, float h);
void launch_absmax2(hipStream_t s, const float* u, const float* v, int pitch, int n, int row_lo, int row_hi,
                    unsigned int* result);
void launch_residual(hipStream_t s, const float* x, const float* x0, int pitch, int n, int row_lo, int row_hi,
                     float alpha, float beta, unsigned int* result);

}  // namespace fluid
