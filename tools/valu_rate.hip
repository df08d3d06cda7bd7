// valu_rate.hip -- issue-rate microbenchmark: cycles per wave64 VALU instruction on gfx950 (standalone tool).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define REP16(x) x x x x x x x x x x x x x x x x

template <int MODE>
__global__ __launch_bounds__(256) void k_rate(float *out, int iters)
{
    float a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7;
    float b = 1.0001f, c = 0.5f;
    int i0 = threadIdx.x, i1 = 1, i2 = 2, i3 = 3, i4 = 4, i5 = 5, i6 = 6, i7 = 7;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {1, 2}, p1 = {3, 4}, p2 = {5, 6}, p3 = {7, 8}, pb = {1.0001f, 1.0002f}, pc = {0.5f, 0.25f};
    for (int it = 0; it < iters; it++) {
        if (MODE == 0) {   // 8 independent v_fma_f32 x 2
            REP16(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                               "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));)
        } else if (MODE == 1) {   // 8 independent v_cvt_f32_i32
            REP16(asm volatile("v_cvt_f32_i32 %0, %8\n v_cvt_f32_i32 %1, %9\n v_cvt_f32_i32 %2, %10\n v_cvt_f32_i32 %3, %11\n"
                               "v_cvt_f32_i32 %4, %12\n v_cvt_f32_i32 %5, %13\n v_cvt_f32_i32 %6, %14\n v_cvt_f32_i32 %7, %15\n"
                               : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3), "=v"(a4), "=v"(a5), "=v"(a6), "=v"(a7)
                               : "v"(i0), "v"(i1), "v"(i2), "v"(i3), "v"(i4), "v"(i5), "v"(i6), "v"(i7));)
        } else {   // 4 independent v_pk_fma_f32 (2 fmas each) x 2 = 8 instrs
            REP16(asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                               "v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                               : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb), "v"(pc));)
        }
    }
    float r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y + (float)(i0 + i7);
    if (r == 12345.678f) out[0] = r;
}

template <int MODE>
static void run(const char *name, int waves_per_simd)
{
    float *out;
    hipMalloc(&out, 4);
    const int iters = 2000;
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    const int blocks = 256 * waves_per_simd;   // 256 CUs x (waves_per_simd blocks of 4 waves)
    k_rate<MODE><<<blocks, 256>>>(out, 10);
    hipEventRecord(a);
    k_rate<MODE><<<blocks, 256>>>(out, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    const double instrs_per_wave = (double)iters * 16 * 8;
    const double ns_per_instr_per_simd = ms * 1e6 / (instrs_per_wave * waves_per_simd);
    printf("%-14s waves/SIMD=%d  %.3f ms  %.3f ns per wave-instr per SIMD  (= %.2f cycles at 2.4 GHz)\n", name, waves_per_simd, ms,
           ns_per_instr_per_simd, ns_per_instr_per_simd * 2.4);
    hipFree(out);
}

int main()
{
    for (int w : {1, 2, 4}) {
        run<0>("v_fma_f32", w);
        run<1>("v_cvt_f32_i32", w);
        run<2>("v_pk_fma_f32", w);
    }
    return 0;
}
