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
    unsigned long long q0 = threadIdx.x, q1 = 1, q2 = 2, q3 = 3;
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
        } else if (MODE == 3) {   // 8 independent v_lshlrev_b64
            REP16(asm volatile("v_lshlrev_b64 %0, 23, %0\n v_lshlrev_b64 %1, 23, %1\n v_lshlrev_b64 %2, 23, %2\n v_lshlrev_b64 %3, 23, %3\n"
                               "v_lshlrev_b64 %0, 23, %0\n v_lshlrev_b64 %1, 23, %1\n v_lshlrev_b64 %2, 23, %2\n v_lshlrev_b64 %3, 23, %3\n"
                               : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3));)
        } else if (MODE == 4) {   // 8 v_alignbit_b32
            REP16(asm volatile("v_alignbit_b32 %0, %0, %8, 9\n v_alignbit_b32 %1, %1, %8, 9\n v_alignbit_b32 %2, %2, %8, 9\n v_alignbit_b32 %3, %3, %8, 9\n"
                               "v_alignbit_b32 %4, %4, %8, 9\n v_alignbit_b32 %5, %5, %8, 9\n v_alignbit_b32 %6, %6, %8, 9\n v_alignbit_b32 %7, %7, %8, 9\n"
                               : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7) : "v"(i0));)
        } else if (MODE == 5) {   // 8 v_lshl_add_u64 (the 64-bit add)
            REP16(asm volatile("v_lshl_add_u64 %0, %0, 0, %1\n v_lshl_add_u64 %1, %1, 0, %2\n v_lshl_add_u64 %2, %2, 0, %3\n v_lshl_add_u64 %3, %3, 0, %0\n"
                               "v_lshl_add_u64 %0, %0, 0, %1\n v_lshl_add_u64 %1, %1, 0, %2\n v_lshl_add_u64 %2, %2, 0, %3\n v_lshl_add_u64 %3, %3, 0, %0\n"
                               : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3));)
        } else if (MODE == 6) {   // 8 v_xor3_b32
            REP16(asm volatile("v_bitop3_b32 %0, %0, %8, %1 bitop3:0x96\n v_bitop3_b32 %1, %1, %8, %2 bitop3:0x96\n v_bitop3_b32 %2, %2, %8, %3 bitop3:0x96\n v_bitop3_b32 %3, %3, %8, %4 bitop3:0x96\n"
                               "v_bitop3_b32 %4, %4, %8, %5 bitop3:0x96\n v_bitop3_b32 %5, %5, %8, %6 bitop3:0x96\n v_bitop3_b32 %6, %6, %8, %7 bitop3:0x96\n v_bitop3_b32 %7, %7, %8, %0 bitop3:0x96\n"
                               : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7) : "v"(i0));)
        } else if (MODE == 7) {
            REP16(asm volatile("v_xor_b32 %0, %0, %8\n v_xor_b32 %1, %1, %8\n v_xor_b32 %2, %2, %8\n v_xor_b32 %3, %3, %8\n"
                               "v_xor_b32 %4, %4, %8\n v_xor_b32 %5, %5, %8\n v_xor_b32 %6, %6, %8\n v_xor_b32 %7, %7, %8\n"
                               : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7) : "v"(i0));)
        } else if (MODE == 8) {
            REP16(asm volatile("v_cvt_f32_i32_sdwa %0, sext(%8) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0\n v_cvt_f32_i32_sdwa %1, sext(%9) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1\n"
                               "v_cvt_f32_i32_sdwa %2, sext(%10) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2\n v_cvt_f32_i32_sdwa %3, sext(%11) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3\n"
                               "v_cvt_f32_i32_sdwa %4, sext(%12) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0\n v_cvt_f32_i32_sdwa %5, sext(%13) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1\n"
                               "v_cvt_f32_i32_sdwa %6, sext(%14) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2\n v_cvt_f32_i32_sdwa %7, sext(%15) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3\n"
                               : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3), "=v"(a4), "=v"(a5), "=v"(a6), "=v"(a7)
                               : "v"(i0), "v"(i1), "v"(i2), "v"(i3), "v"(i4), "v"(i5), "v"(i6), "v"(i7));)
        } else if (MODE == 9) {
            REP16(asm volatile("v_cvt_i32_f32_sdwa %0, %8 dst_sel:BYTE_0 dst_unused:UNUSED_PRESERVE src0_sel:DWORD\n v_cvt_i32_f32_sdwa %1, %8 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD\n"
                               "v_cvt_i32_f32_sdwa %2, %8 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:DWORD\n v_cvt_i32_f32_sdwa %3, %8 dst_sel:BYTE_3 dst_unused:UNUSED_PRESERVE src0_sel:DWORD\n"
                               "v_cvt_i32_f32_sdwa %4, %8 dst_sel:BYTE_0 dst_unused:UNUSED_PRESERVE src0_sel:DWORD\n v_cvt_i32_f32_sdwa %5, %8 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD\n"
                               "v_cvt_i32_f32_sdwa %6, %8 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:DWORD\n v_cvt_i32_f32_sdwa %7, %8 dst_sel:BYTE_3 dst_unused:UNUSED_PRESERVE src0_sel:DWORD\n"
                               : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7) : "v"(b));)
        } else if (MODE == 10) {
            REP16(asm volatile("v_bfi_b32 %0, %8, %0, %1\n v_bfi_b32 %1, %8, %1, %2\n v_bfi_b32 %2, %8, %2, %3\n v_bfi_b32 %3, %8, %3, %4\n"
                               "v_bfi_b32 %4, %8, %4, %5\n v_bfi_b32 %5, %8, %5, %6\n v_bfi_b32 %6, %8, %6, %7\n v_bfi_b32 %7, %8, %7, %0\n"
                               : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7) : "v"(i0));)
        } else if (MODE == 11) {
            REP16(asm volatile("v_max3_f32 %0, %0, %8, %9\n v_max3_f32 %1, %1, %8, %9\n v_max3_f32 %2, %2, %8, %9\n v_max3_f32 %3, %3, %8, %9\n"
                               "v_max3_f32 %4, %4, %8, %9\n v_max3_f32 %5, %5, %8, %9\n v_max3_f32 %6, %6, %8, %9\n v_max3_f32 %7, %7, %8, %9\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));)
        } else if (MODE == 12) {   // 4-bit signed -> float/16 through the interpolation-offset conversion
            REP16(asm volatile("v_cvt_off_f32_i4 %0, %8\n v_cvt_off_f32_i4 %1, %9\n v_cvt_off_f32_i4 %2, %10\n v_cvt_off_f32_i4 %3, %11\n"
                               "v_cvt_off_f32_i4 %4, %12\n v_cvt_off_f32_i4 %5, %13\n v_cvt_off_f32_i4 %6, %14\n v_cvt_off_f32_i4 %7, %15\n"
                               : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3), "=v"(a4), "=v"(a5), "=v"(a6), "=v"(a7)
                               : "v"(i0), "v"(i1), "v"(i2), "v"(i3), "v"(i4), "v"(i5), "v"(i6), "v"(i7));)
        } else if (MODE == 13) {
            REP16(asm volatile("v_cvt_off_f32_i4_sdwa %0, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0\n v_cvt_off_f32_i4_sdwa %1, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1\n"
                               "v_cvt_off_f32_i4_sdwa %2, %10 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2\n v_cvt_off_f32_i4_sdwa %3, %11 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3\n"
                               "v_cvt_off_f32_i4_sdwa %4, %12 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0\n v_cvt_off_f32_i4_sdwa %5, %13 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1\n"
                               "v_cvt_off_f32_i4_sdwa %6, %14 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2\n v_cvt_off_f32_i4_sdwa %7, %15 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3\n"
                               : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3), "=v"(a4), "=v"(a5), "=v"(a6), "=v"(a7)
                               : "v"(i0), "v"(i1), "v"(i2), "v"(i3), "v"(i4), "v"(i5), "v"(i6), "v"(i7));)
        } else if (MODE == 14) {
            REP16(asm volatile("v_cvt_f32_ubyte0 %0, %8\n v_cvt_f32_ubyte1 %1, %9\n v_cvt_f32_ubyte2 %2, %10\n v_cvt_f32_ubyte3 %3, %11\n"
                               "v_cvt_f32_ubyte0 %4, %12\n v_cvt_f32_ubyte1 %5, %13\n v_cvt_f32_ubyte2 %6, %14\n v_cvt_f32_ubyte3 %7, %15\n"
                               : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3), "=v"(a4), "=v"(a5), "=v"(a6), "=v"(a7)
                               : "v"(i0), "v"(i1), "v"(i2), "v"(i3), "v"(i4), "v"(i5), "v"(i6), "v"(i7));)
        } else if (MODE == 15) {
            REP16(asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                               "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));)
        } else if (MODE == 16) {   // VOP2 fmac
            REP16(asm volatile("v_fmac_f32 %0, %8, %9\n v_fmac_f32 %1, %8, %9\n v_fmac_f32 %2, %8, %9\n v_fmac_f32 %3, %8, %9\n"
                               "v_fmac_f32 %4, %8, %9\n v_fmac_f32 %5, %8, %9\n v_fmac_f32 %6, %8, %9\n v_fmac_f32 %7, %8, %9\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));)
        } else if (MODE == 17) {   // v_pk_mul_f32
            REP16(asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n"
                               "v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n"
                               : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb));)
        } else if (MODE == 18) {   // v_and_b32 with a literal
            REP16(asm volatile("v_and_b32 %0, 0xf0f0f0f0, %8\n v_and_b32 %1, 0xf0f0f0f0, %9\n v_and_b32 %2, 0xf0f0f0f0, %10\n v_and_b32 %3, 0xf0f0f0f0, %11\n"
                               "v_and_b32 %4, 0xf0f0f0f0, %12\n v_and_b32 %5, 0xf0f0f0f0, %13\n v_and_b32 %6, 0xf0f0f0f0, %14\n v_and_b32 %7, 0xf0f0f0f0, %15\n"
                               : "=v"(i0), "=v"(i1), "=v"(i2), "=v"(i3), "=v"(i4), "=v"(i5), "=v"(i6), "=v"(i7)
                               : "v"(i0), "v"(i1), "v"(i2), "v"(i3), "v"(i4), "v"(i5), "v"(i6), "v"(i7));)
        } else {   // 4 independent v_pk_fma_f32 (2 fmas each) x 2 = 8 instrs
            REP16(asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                               "v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                               : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb), "v"(pc));)
        }
    }
    float r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y + (float)(i0 + i7) + (float)(q0 + q1 + q2 + q3);
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
    for (int w : {2, 4}) {
        run<0>("v_fma_f32", w);
        run<1>("v_cvt_f32_i32", w);
        run<2>("v_pk_fma_f32", w);
        run<3>("v_lshlrev_b64", w);
        run<4>("v_alignbit_b32", w);
        run<5>("v_lshl_add_u64", w);
        run<6>("v_bitop3_b32 (xor3)", w);
        run<7>("v_xor_b32", w);
        run<8>("v_cvt_f32_i32_sdwa", w);
        run<9>("v_cvt_i32_f32_sdwa(dst)", w);
        run<10>("v_bfi_b32", w);
        run<11>("v_max3_f32", w);
        run<12>("v_cvt_off_f32_i4", w);
        run<13>("v_cvt_off_f32_i4_sdwa", w);
        run<14>("v_cvt_f32_ubyteN", w);
        run<15>("v_mul_f32", w);
        run<16>("v_fmac_f32 (VOP2)", w);
        run<17>("v_pk_mul_f32", w);
        run<18>("v_and_b32 literal", w);
    }
    return 0;
}
