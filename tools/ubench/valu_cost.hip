// valu_cost.hip -- cycles per VALU instruction on gfx950, as one wave sees them (s_memtime around an
// unrolled block of 256 instructions, repeated).  Variants: independent / dependent chains of
// v_add_f32, v_pk_add_f32, v_pk_mul_f32, v_add_f32 with a DPP wave shift, v_mov_b32, v_cvt/v_mul_f64;
// with 1, 2 and 4 waves per SIMD.   hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_cost.hip -o valu_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v2f __attribute__((ext_vector_type(2)));

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP32(x) REP16(x) REP16(x)

template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* cyc, int iters)
{
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    v2f p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6};
    const float c = 1.0001f;
    const v2f c2 = {c, c};
    double d0 = a0, d1 = a1, d2 = a2, d3 = a3;
    asm volatile("s_waitcnt lgkmcnt(0)");
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) {   // 8 independent v_add_f32 chains
            REP32(asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
                               "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));)
        } else if (KIND == 1) {   // one dependent v_add_f32 chain
            REP32(asm volatile("v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n"
                               "v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1" : "+v"(a0) : "v"(c));)
        } else if (KIND == 2) {   // 8 independent v_pk_add_f32
            REP32(asm volatile("v_pk_add_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %8\n v_pk_add_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %8\n"
                               "v_pk_add_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %8\n v_pk_add_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %8"
                               : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(c2));)
        } else if (KIND == 3) {   // dependent v_pk_add_f32 chain (s_nop as the compiler inserts it)
            REP32(asm volatile("v_pk_add_f32 %0, %0, %1\n s_nop 0\n v_pk_add_f32 %0, %0, %1\n s_nop 0\n v_pk_add_f32 %0, %0, %1\n s_nop 0\n v_pk_add_f32 %0, %0, %1\n s_nop 0\n"
                               "v_pk_add_f32 %0, %0, %1\n s_nop 0\n v_pk_add_f32 %0, %0, %1\n s_nop 0\n v_pk_add_f32 %0, %0, %1\n s_nop 0\n v_pk_add_f32 %0, %0, %1\n s_nop 0" : "+v"(p0) : "v"(c2));)
        } else if (KIND == 4) {   // 8 independent DPP adds
            REP32(asm volatile("v_add_f32_dpp %0, %0, %8 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %1, %1, %8 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                               "v_add_f32_dpp %2, %2, %8 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %3, %3, %8 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                               "v_add_f32_dpp %4, %4, %8 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %5, %5, %8 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                               "v_add_f32_dpp %6, %6, %8 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %7, %7, %8 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));)
        } else if (KIND == 5) {   // 8 independent v_mov_b32 (pairs swapped so nothing folds)
            REP32(asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %4\n v_mov_b32 %4, %5\n v_mov_b32 %5, %6\n v_mov_b32 %6, %7\n v_mov_b32 %7, %0"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
        } else if (KIND == 6) {   // f64 division form: cvt, mul_f64, cvt on 4 independent values (12 instr) x2 = 24... use 8 groups of 3
            REP32(asm volatile("v_cvt_f64_f32 %4, %0\n v_cvt_f64_f32 %5, %1\n v_mul_f64 %4, %4, %8\n v_mul_f64 %5, %5, %8\n v_cvt_f32_f64 %0, %4\n v_cvt_f32_f64 %1, %5\n"
                               "v_cvt_f64_f32 %6, %2\n v_cvt_f64_f32 %7, %3\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"((double)c));)
        } else if (KIND == 7) {   // 8 independent v_pk_mul_f32
            REP32(asm volatile("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n"
                               "v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8"
                               : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(c2));)
        } else if (KIND == 8) {   // v_mov_b64 x8
            REP32(asm volatile("v_mov_b64 %0, %1\n v_mov_b64 %1, %2\n v_mov_b64 %2, %3\n v_mov_b64 %3, %4\n v_mov_b64 %4, %5\n v_mov_b64 %5, %6\n v_mov_b64 %6, %7\n v_mov_b64 %7, %0"
                               : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7));)
        } else if (KIND == 9) {   // 2 interleaved dependent pk chains (no nops needed?) -- pairs of chains
            REP32(asm volatile("v_pk_add_f32 %0, %0, %2\n v_pk_add_f32 %1, %1, %2\n v_pk_add_f32 %0, %0, %2\n v_pk_add_f32 %1, %1, %2\n"
                               "v_pk_add_f32 %0, %0, %2\n v_pk_add_f32 %1, %1, %2\n v_pk_add_f32 %0, %0, %2\n v_pk_add_f32 %1, %1, %2" : "+v"(p0), "+v"(p1) : "v"(c2));)
        }
    }
    asm volatile("s_nop 0");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p2.x + p3.x + p4.x + p5.x + p6.x + p7.x +
                                                 p1.y + p2.y + p3.y + (float)(d0 + d1 + d2 + d3);
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int KIND>
void run(const char* name, int instr_per_rep)
{
    float* out;
    unsigned long long* cyc;
    hipMalloc(&out, 4 * 256 * 1024 * 4);
    hipMalloc(&cyc, 8 * 4 * 1024 * 4);
    const int iters = 64;
    printf("%-44s", name);
    for (int wps : {1, 2, 4}) {       // waves per SIMD: blocks of 256 threads (1 wave per SIMD each), `wps` blocks per CU
        const int blocks = 256 * wps;
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
        std::vector<unsigned long long> h(blocks * 4);
        hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
        double mean = 0;
        for (auto v : h) mean += (double)v;
        mean /= h.size();
        printf("  %dw/SIMD: %6.2f cyc/instr", wps, mean / ((double)iters * 32 * instr_per_rep));
    }
    printf("\n");
    hipFree(out);
    hipFree(cyc);
}

int main()
{
    run<0>("v_add_f32, 8 independent chains", 8);
    run<1>("v_add_f32, one dependent chain", 8);
    run<2>("v_pk_add_f32, 8 independent", 8);
    run<3>("v_pk_add_f32 dependent (+s_nop 0 each)", 8);
    run<9>("v_pk_add_f32, 2 interleaved chains", 8);
    run<7>("v_pk_mul_f32, 8 independent", 8);
    run<4>("v_add_f32_dpp wave_shr/shl, 8 independent", 8);
    run<5>("v_mov_b32 x8", 8);
    run<8>("v_mov_b64 x8", 8);
    run<6>("cvt_f64_f32/mul_f64/cvt_f32_f64 mix (8 instr)", 8);
    return 0;
}
