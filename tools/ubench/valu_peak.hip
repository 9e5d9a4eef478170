// valu_peak.hip -- what one gfx950 SIMD actually sustains, in wave-instructions per microsecond (wall clock,
// HIP events), for the instruction kinds of the fused Jacobi kernel, at 1/2/4/8 waves per SIMD; and what one
// s_memtime tick is in nanoseconds.  These are the numbers behind bench.py's `roofline_actual` (valu bound).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_peak.hip -o tools/ubench/valu_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v2f __attribute__((ext_vector_type(2)));

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))

// KIND 0: v_add_f32   1: v_pk_add_f32   2: v_pk_fma_f32   3: v_add_f32_dpp   4: v_min3_f32   5: f64 division triple
// 6: one pair-stage of the general-form stencil, division mode 2 (12 instr)   7: the same, mode 3 (10 instr)
// 8: pressure form (6 instr)
template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* ticks, int iters)
{
    float a[8];
    v2f p[8];
    for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x + i; p[i] = (v2f){a[i], a[i] + 0.5f}; }
    const float c = 1.0001f;
    const v2f c2 = {c, c};
    const double cd = 1.0001;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (KIND == 0) { REP16(asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(c));) }
        if (KIND == 1) { REP16(asm volatile("v_pk_add_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %8\n v_pk_add_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %8\n v_pk_add_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %8\n v_pk_add_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %8" : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7]) : "v"(c2));) }
        if (KIND == 2) { REP16(asm volatile("v_pk_fma_f32 %0, %0, %8, %8\n v_pk_fma_f32 %1, %1, %8, %8\n v_pk_fma_f32 %2, %2, %8, %8\n v_pk_fma_f32 %3, %3, %8, %8\n v_pk_fma_f32 %4, %4, %8, %8\n v_pk_fma_f32 %5, %5, %8, %8\n v_pk_fma_f32 %6, %6, %8, %8\n v_pk_fma_f32 %7, %7, %8, %8" : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7]) : "v"(c2));) }
        if (KIND == 3) { REP16(asm volatile("v_add_f32_dpp %0, %0, %8 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %1, %1, %8 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %2, %2, %8 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %3, %3, %8 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %4, %4, %8 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %5, %5, %8 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %6, %6, %8 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %7, %7, %8 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(c));) }
        if (KIND == 4) { REP16(asm volatile("v_min3_f32 %0, %0, |%8|, |%1|\n v_min3_f32 %1, %1, |%8|, |%2|\n v_min3_f32 %2, %2, |%8|, |%3|\n v_min3_f32 %3, %3, |%8|, |%4|\n v_min3_f32 %4, %4, |%8|, |%5|\n v_min3_f32 %5, %5, |%8|, |%6|\n v_min3_f32 %6, %6, |%8|, |%7|\n v_min3_f32 %7, %7, |%8|, |%0|" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(c));) }
        if (KIND == 5) {
            double d0, d1, d2, d3;
            REP16(asm volatile("v_cvt_f64_f32 %4, %0\n v_cvt_f64_f32 %5, %1\n v_cvt_f64_f32 %6, %2\n v_cvt_f64_f32 %7, %3\n v_mul_f64 %4, %4, %8\n v_mul_f64 %5, %5, %8\n v_mul_f64 %6, %6, %8\n v_mul_f64 %7, %7, %8\n"
                               "v_cvt_f32_f64 %0, %4\n v_cvt_f32_f64 %1, %5\n v_cvt_f32_f64 %2, %6\n v_cvt_f32_f64 %3, %7" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3) : "v"(cd));)
        }
        if (KIND == 6 || KIND == 7 || KIND == 8) {
            // four independent pair-stages per repetition (as four waves' worth of ILP inside one wave)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    v2f& s = p[j];
                    v2f& up = p[4 + j];
                    float h0, h1;
                    asm volatile("v_add_f32_dpp %0, %2, %3 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %1, %3, %2 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=&v"(h0), "=&v"(h1) : "v"(s.y), "v"(s.x));
                    v2f t = {h0, h1};
                    asm volatile("v_pk_add_f32 %0, %0, %1\n v_pk_add_f32 %0, %0, %1" : "+v"(t) : "v"(up));
                    if (KIND == 8) {
                        asm volatile("v_pk_add_f32 %0, %0, %1\n v_pk_mul_f32 %0, %0, %2" : "+v"(t) : "v"(up), "v"(c2));
                    } else {
                        asm volatile("v_pk_mul_f32 %0, %0, %2\n v_pk_add_f32 %0, %0, %1" : "+v"(t) : "v"(up), "v"(c2));
                        if (KIND == 6) {
                            double d0, d1;
                            asm volatile("v_cvt_f64_f32 %1, %0\n v_mul_f64 %1, %1, %2\n v_cvt_f32_f64 %0, %1" : "+v"(t.x), "=&v"(d0) : "v"(cd));
                            asm volatile("v_cvt_f64_f32 %1, %0\n v_mul_f64 %1, %1, %2\n v_cvt_f32_f64 %0, %1" : "+v"(t.y), "=&v"(d1) : "v"(cd));
                        } else {
                            v2f q;
                            asm volatile("v_pk_mul_f32 %1, %0, %2\n v_pk_fma_f32 %0, %0, %2, %1" : "+v"(t), "=&v"(q) : "v"(c2));
                            asm volatile("v_min3_f32 %0, %0, |%1|, |%2|" : "+v"(a[j]) : "v"(t.x), "v"(t.y));
                            asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(a[4 + j]) : "v"(t.x), "v"(t.y));
                        }
                    }
                    s = t;
                }
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float acc = 0;
    for (int i = 0; i < 8; ++i) acc += a[i] + p[i].x + p[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if ((threadIdx.x & 63) == 0) ticks[blockIdx.x * 4 + threadIdx.x / 64] = t1 - t0;
}

template <int KIND>
void run(const char* name, int instr_per_iter)
{
    float* out;
    unsigned long long* ticks;
    hipMalloc(&out, sizeof(float) * 256 * 256 * 8);
    hipMalloc(&ticks, 8 * 4 * 256 * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 20000;
    printf("%-46s", name);
    for (int wps : {1, 2, 4, 8}) {
        const int blocks = 256 * wps;
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, ticks, 200);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, ticks, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(blocks * 4);
        hipMemcpy(h.data(), ticks, h.size() * 8, hipMemcpyDeviceToHost);
        double mean = 0;
        for (auto v : h) mean += (double)v;
        mean /= h.size();
        // wave-instructions per SIMD per microsecond; ns per s_memtime tick
        const double per_simd = (double)iters * instr_per_iter * wps / (ms * 1e3);
        printf(" %dw: %6.0f/us (tick %.2f ns)", wps, per_simd, ms * 1e6 / mean);
    }
    printf("\n");
    hipFree(out);
    hipFree(ticks);
}

int main()
{
    run<0>("v_add_f32", 128);
    run<1>("v_pk_add_f32", 128);
    run<2>("v_pk_fma_f32", 128);
    run<3>("v_add_f32_dpp", 128);
    run<4>("v_min3_f32", 128);
    run<5>("cvt_f64_f32 / mul_f64 / cvt_f32_f64", 192);
    run<6>("general-form pair-stage, division mode 2 (12)", 16 * 12);
    run<7>("general-form pair-stage, division mode 3 (10)", 16 * 10);
    run<8>("pressure-form pair-stage (6)", 16 * 6);
    return 0;
}
