// div_flag.hip -- is "fma(x, hi, x*lo) unless the wave's sticky UNDERFLOW flag came up" an exact a/beta?
// For every one of the 2^32 float inputs (64 consecutive bit patterns per wave and trial): clear
// TRAPSTS.EXCP, compute the cheap form (scalar and packed encodings), read TRAPSTS.EXCP; the result must
// equal a/beta bit for bit in every wave whose underflow bit stayed clear.  Also reports how many waves
// were flagged (the fraction of steps that would take the slow path on such data).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/ubench/div_flag.hip -o tools/ubench/div_flag
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
constexpr int TRAPSTS_EXCP = 3 | (0 << 6) | (8 << 11);      // hwreg(HW_REG_TRAPSTS, 0, 9)
constexpr unsigned UNDERFLOW = 1u << 4;

template <bool PACKED>
__global__ void check(float beta, float ch, float cl, unsigned long long* res)   // res: 0 bad-in-clear, 1 flagged waves, 2 waves, 3 excp OR
{
    const unsigned long long wave = ((unsigned long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const unsigned long long nwaves = ((unsigned long long)gridDim.x * blockDim.x) >> 6;
    const unsigned lane = threadIdx.x & 63;
    unsigned long long bad = 0, flagged = 0, total = 0;
    unsigned seen = 0;
    for (unsigned long long w = wave; w < (1ull << 26); w += nwaves) {
        const float a = __uint_as_float((unsigned)(w * 64 + lane));
        const float ref = a / beta;
        asm volatile("s_nop 4");
        __builtin_amdgcn_s_setreg(TRAPSTS_EXCP, 0);
        asm volatile("s_nop 4");
        float got;
        if (PACKED) {
            v2f x = {a, a}, u, q;
            asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(u) : "v"(x), "v"((v2f){cl, cl}));
            asm volatile("s_nop 1\n v_pk_fma_f32 %0, %1, %2, %3" : "=v"(q) : "v"(x), "v"((v2f){ch, ch}), "v"(u));
            got = q.y;
        } else {
            float u;
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(u) : "v"(a), "v"(cl));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(got) : "v"(a), "v"(ch), "v"(u));
        }
        asm volatile("s_nop 4");
        const unsigned excp = __builtin_amdgcn_s_getreg(TRAPSTS_EXCP);
        seen |= excp;
        const bool ok = (ref != ref) ? (got != got) : (__float_as_uint(got) == __float_as_uint(ref));
        const bool flag = (excp & UNDERFLOW) != 0;
        if (!flag && !ok) ++bad;
        if (lane == 0) { ++total; flagged += flag; }
    }
    if (bad) atomicAdd(&res[0], bad);
    if (lane == 0) { atomicAdd(&res[1], flagged); atomicAdd(&res[2], total); atomicOr(&res[3], (unsigned long long)seen); }
}

int main()
{
    unsigned long long* res;
    hipMalloc(&res, 32);
    const float betas[] = {1.00016f, 102.606407f, 671.8304f, 26828.27f, 3.3f, 0.75f, 3.0f, 5.0f, 1.0f + 4.0f * 0.016f * 0.0025f * 4094 * 4094,
                           1.0f + 4.0f * 0.016f * 0.1f * 4094 * 4094, 1.0f + 4.0f * 0.016f * 0.0025f * 8190 * 8190, 1.0f + 4.0f * 0.016f * 0.1f * 8190 * 8190, 1e-3f, 7e5f};
    for (float beta : betas) {
        const double yd = 1.0 / (double)beta;
        float ch = (float)yd;
        if ((double)ch > yd) ch = nextafterf(ch, 0.f);
        const float cl = (float)(yd - (double)ch);
        for (int packed = 0; packed < 2; ++packed) {
            hipMemset(res, 0, 32);
            if (packed) hipLaunchKernelGGL(check<true>, dim3(4096), dim3(256), 0, 0, beta, ch, cl, res);
            else hipLaunchKernelGGL(check<false>, dim3(4096), dim3(256), 0, 0, beta, ch, cl, res);
            unsigned long long h[4];
            hipMemcpy(h, res, 32, hipMemcpyDeviceToHost);
            printf("beta=%-14.9g %s  wrong in unflagged waves: %llu   flagged waves: %llu / %llu (%.2f%%)   EXCP bits seen 0x%llx\n", beta,
                   packed ? "packed" : "scalar", h[0], h[1], h[2], 100.0 * h[1] / h[2], h[3]);
        }
    }
    return 0;
}
