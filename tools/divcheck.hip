// divcheck.hip -- exhaustive check + throughput of constant-divisor division candidates.
//   A: a / beta                         (IEEE, the reference's operation)
//   B: q0=a*y; r=fma(-beta,q0,a); q1=fma(r,y,q0)   (fp32 residual correction; range-guarded)
//   C: (float)((double)a * yd)          (f64 reciprocal multiply)
// For every one of the 2^32 float bit patterns `a`, B and C are compared bitwise with A
// (NaN == NaN).  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/divcheck.hip -o divcheck
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

__device__ __forceinline__ float divB(float a, float beta, float y)
{
    const float q0 = a * y;
    const float r = __builtin_fmaf(-beta, q0, a);
    return __builtin_fmaf(r, y, q0);
}
__device__ __forceinline__ float divC(float a, double yd) { return (float)((double)a * yd); }

__global__ void check(float beta, float y, double yd, unsigned long long* bad)  // bad[0]=B all, bad[1]=B in range, bad[2]=C
{
    const unsigned long long tid = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long nb = 0, nbr = 0, nc = 0;
    for (unsigned long long k = tid; k < (1ull << 32); k += (unsigned long long)gridDim.x * blockDim.x) {
        const float a = __uint_as_float((unsigned)k);
        const float ref = a / beta;
        const float b = divB(a, beta, y), c = divC(a, yd);
        const bool refnan = ref != ref;
        const bool okb = refnan ? (b != b) : (__float_as_uint(b) == __float_as_uint(ref));
        const bool okc = refnan ? (c != c) : (__float_as_uint(c) == __float_as_uint(ref));
        const float aa = fabsf(a);
        nb += !okb;
        nbr += (!okb) && (aa >= 0x1p-100f) && (aa <= 0x1p100f);
        nc += !okc;
    }
    if (nb) atomicAdd(&bad[0], nb);
    if (nbr) atomicAdd(&bad[1], nbr);
    if (nc) atomicAdd(&bad[2], nc);
}

template <int MODE>
__global__ void bench(float* out, float beta, float y, double yd, int iters)
{
    float a0 = threadIdx.x * 1e-3f + 1.0f, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f;
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) { a0 = a0 / beta + 1.f; a1 = a1 / beta + 1.f; a2 = a2 / beta + 1.f; a3 = a3 / beta + 1.f; }
        if (MODE == 1) { a0 = divB(a0, beta, y) + 1.f; a1 = divB(a1, beta, y) + 1.f; a2 = divB(a2, beta, y) + 1.f; a3 = divB(a3, beta, y) + 1.f; }
        if (MODE == 2) { a0 = divC(a0, yd) + 1.f; a1 = divC(a1, yd) + 1.f; a2 = divC(a2, yd) + 1.f; a3 = divC(a3, yd) + 1.f; }
        if (MODE == 3) { a0 = a0 * y + 1.f; a1 = a1 * y + 1.f; a2 = a2 * y + 1.f; a3 = a3 * y + 1.f; }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3;
}

int main(int argc, char** argv)
{
    unsigned long long* bad;
    hipMalloc(&bad, 24);
    float* out;
    hipMalloc(&out, 4 * 1024 * 256 * 8);
    const float betas[] = {4.0f, 1.00016f, 102.606407f, 671.8304f, 26828.27f, 107376.3f, 3.3f, 0.75f, 1.0f + 4.0f * 0.016f * 0.0025f * 4094 * 4094};
    for (float beta : betas) {
        const float y = 1.0f / beta;
        const double yd = 1.0 / (double)beta;
        hipMemset(bad, 0, 24);
        hipLaunchKernelGGL(check, dim3(4096), dim3(256), 0, 0, beta, y, yd, bad);
        unsigned long long h[3];
        hipMemcpy(h, bad, 24, hipMemcpyDeviceToHost);
        printf("beta=%-14.9g  B mismatches: %llu (in [2^-100,2^100]: %llu)   C mismatches: %llu\n", beta, h[0], h[1], h[2]);
    }
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 4096;
    const char* names[] = {"A true div", "B fp32 residual", "C f64 mul", "D fp32 mul (pow2 only)"};
    for (int m = 0; m < 4; ++m) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (m == 0) hipLaunchKernelGGL(bench<0>, dim3(1024 * 8), dim3(256), 0, 0, out, 3.3f, 1 / 3.3f, 1 / 3.3, iters);
            if (m == 1) hipLaunchKernelGGL(bench<1>, dim3(1024 * 8), dim3(256), 0, 0, out, 3.3f, 1 / 3.3f, 1 / 3.3, iters);
            if (m == 2) hipLaunchKernelGGL(bench<2>, dim3(1024 * 8), dim3(256), 0, 0, out, 3.3f, 1 / 3.3f, 1 / 3.3, iters);
            if (m == 3) hipLaunchKernelGGL(bench<3>, dim3(1024 * 8), dim3(256), 0, 0, out, 3.3f, 1 / 3.3f, 1 / 3.3, iters);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
        }
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double ops = 4.0 * iters * 1024 * 8 * 256;
        printf("%-24s %.3f ms  -> %.2f Tdiv+add/s\n", names[m], ms, ops / ms / 1e9);
    }
    return 0;
}
