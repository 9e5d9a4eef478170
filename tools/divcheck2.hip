// divcheck2.hip -- exhaustive check of the two-term reciprocal division  q = fma(a, hi, a*lo),  hi + lo ~ 1/beta,
// against a / beta for all 2^32 float inputs a, for several ways of choosing (hi, lo):
//   N: hi = RN32(1/beta),             lo = RN32(1/beta - hi)   (lo of either sign)
//   D: hi = RD32(1/beta) (round down), lo = RN32(1/beta - hi)  (lo >= 0: -0 * hi + -0 * lo = -0, as -0 / beta)
// Reports mismatches over all inputs, mismatches whose reference quotient is zero or at least 2^-100 in
// magnitude (the range the fused Jacobi kernel's guard lets through), and the largest |quotient| that mismatches.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/divcheck2.hip -o tools/divcheck2 && tools/divcheck2 [beta ...]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

__global__ void check(float beta, float hi, float lo, unsigned long long* bad, unsigned* worst)
{
    const unsigned long long tid = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long n_all = 0, n_guard = 0;
    unsigned w = 0;
    for (unsigned long long k = tid; k < (1ull << 32); k += (unsigned long long)gridDim.x * blockDim.x) {
        const float a = __uint_as_float((unsigned)k);
        const float ref = a / beta;
        const float p = a * lo;
        const float q = __builtin_fmaf(a, hi, p);
        const bool ok = (ref != ref) ? (q != q) : (__float_as_uint(q) == __float_as_uint(ref));
        if (!ok) {
            ++n_all;
            const float m = fabsf(ref);
            if (m == 0.0f || m >= 0x1p-100f) ++n_guard;
            const unsigned mb = __float_as_uint(m);
            if (mb > w) w = mb;
        }
    }
    if (n_all) atomicAdd(&bad[0], n_all);
    if (n_guard) atomicAdd(&bad[1], n_guard);
    if (w) atomicMax(worst, w);
}

static float round_down(double v)
{
    float f = (float)v;
    if ((double)f > v) f = nextafterf(f, -INFINITY);
    return f;
}

int main(int argc, char** argv)
{
    std::vector<float> betas;
    for (int i = 1; i < argc; ++i) betas.push_back((float)atof(argv[i]));
    if (betas.empty()) {
        auto coef = [](int n, float dt, float c) { volatile float a = dt * c; a = a * (float)n; a = a * (float)n; volatile float f = 4.0f * a; return 1.0f + f; };
        for (int n : {126, 1022, 4094, 8190, 16382}) { betas.push_back(coef(n, 0.016f, 0.0025f)); betas.push_back(coef(n, 0.016f, 0.1f)); }
        for (float b : {3.0f, 6.0f, 12.0f, 10.0f, 1.00016f, 102.606407f, 2682.734f, 0.75f, 3.3f, 5e-5f, 7e5f, 2.2f, 1e10f, 1e-10f, 4.0f})
            betas.push_back(b);
    }
    unsigned long long* bad;
    unsigned* worst;
    hipMalloc(&bad, 16);
    hipMalloc(&worst, 4);
    for (float beta : betas) {
        const double y = 1.0 / (double)beta;
        for (int variant = 0; variant < 2; ++variant) {
            const float hi = variant == 0 ? (float)y : round_down(y);
            const float lo = (float)(y - (double)hi);
            hipMemset(bad, 0, 16);
            hipMemset(worst, 0, 4);
            hipLaunchKernelGGL(check, dim3(4096), dim3(256), 0, 0, beta, hi, lo, bad, worst);
            unsigned long long h[2];
            unsigned w;
            hipMemcpy(h, bad, 16, hipMemcpyDeviceToHost);
            hipMemcpy(&w, worst, 4, hipMemcpyDeviceToHost);
            float wf;
            memcpy(&wf, &w, 4);
            printf("beta=%-14.9g %c hi=%-14.9g lo=%-14.9g mismatches: %-10llu zero-or->=2^-100: %-6llu largest |q| wrong: %g (2^%d)\n", beta,
                   variant == 0 ? 'N' : 'D', hi, lo, h[0], h[1], wf, w ? ilogbf(wf) : 0);
        }
    }
    return 0;
}
