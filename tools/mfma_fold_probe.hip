// mfma_fold_probe.hip -- how many fold instructions hide beside v_mfma_scale_f32_32x32x64_f8f6f4 (FP6 operands) on gfx950.
//   build: hipcc --offload-arch=gfx950 -O3 -o tools/mfma_fold_probe tools/mfma_fold_probe.hip      (experiment, not product)
//
// The GEMM's K-loop is, per 32x32 tile and K-block, one MFMA (8 passes = 32 cycles of the matrix pipe) and 16 fmas per lane
// that fold its result into the accumulator.  Both go through the SIMD's one VALU issue port.  This probe times
// hand-written instruction streams -- the whole loop is ONE asm statement, so hipcc cannot re-schedule it -- for different
// fold shapes and 1..4 waves per SIMD, in shader cycles (s_memtime) per MFMA.
//
// Registers (named literally, listed as clobbers): v[0:63] four accumulators, v[64:79] / v[80:95] / v[96:111] result sets,
// v[112:117] / v[118:123] FP6 fragments, v124 the E8M0 scale word, s[40:41] the fold factor.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define MFMA(dst) "v_mfma_scale_f32_32x32x64_f8f6f4 " dst ", v[112:117], v[118:123], 0, v124, v124 op_sel_hi:[0,0,0] cbsz:2 blgp:2\n"
// 16 scalar fmas: acc[a..a+15] += c * r[r..r+15]
#define F1(a, r) "v_fma_f32 v" #a ", s40, v" #r ", v" #a "\n"
#define P1(a0, a1, r0, r1) "v_pk_fma_f32 v[" #a0 ":" #a1 "], s[40:41], v[" #r0 ":" #r1 "], v[" #a0 ":" #a1 "] op_sel_hi:[0,1,1]\n"

#define FOLD16_A0_R0 F1(0,64) F1(1,65) F1(2,66) F1(3,67) F1(4,68) F1(5,69) F1(6,70) F1(7,71) F1(8,72) F1(9,73) F1(10,74) F1(11,75) F1(12,76) F1(13,77) F1(14,78) F1(15,79)
#define FOLD16_A1_R1 F1(16,80) F1(17,81) F1(18,82) F1(19,83) F1(20,84) F1(21,85) F1(22,86) F1(23,87) F1(24,88) F1(25,89) F1(26,90) F1(27,91) F1(28,92) F1(29,93) F1(30,94) F1(31,95)
#define FOLDPK_A0_R0 P1(0,1,64,65) P1(2,3,66,67) P1(4,5,68,69) P1(6,7,70,71) P1(8,9,72,73) P1(10,11,74,75) P1(12,13,76,77) P1(14,15,78,79)
#define FOLDPK_A1_R1 P1(16,17,80,81) P1(18,19,82,83) P1(20,21,84,85) P1(22,23,86,87) P1(24,25,88,89) P1(26,27,90,91) P1(28,29,92,93) P1(30,31,94,95)
// mixed: NS scalar fmas first (they sit in the MFMA's shadow), the rest packed
#define FOLDMIX4_A0_R0 F1(0,64) F1(1,65) F1(2,66) F1(3,67) P1(4,5,68,69) P1(6,7,70,71) P1(8,9,72,73) P1(10,11,74,75) P1(12,13,76,77) P1(14,15,78,79)
#define FOLDMIX4_A1_R1 F1(16,80) F1(17,81) F1(18,82) F1(19,83) P1(20,21,84,85) P1(22,23,86,87) P1(24,25,88,89) P1(26,27,90,91) P1(28,29,92,93) P1(30,31,94,95)
#define FOLDMIX6_A0_R0 F1(0,64) F1(1,65) F1(2,66) F1(3,67) F1(4,68) F1(5,69) P1(6,7,70,71) P1(8,9,72,73) P1(10,11,74,75) P1(12,13,76,77) P1(14,15,78,79)
#define FOLDMIX6_A1_R1 F1(16,80) F1(17,81) F1(18,82) F1(19,83) F1(20,84) F1(21,85) P1(22,23,86,87) P1(24,25,88,89) P1(26,27,90,91) P1(28,29,92,93) P1(30,31,94,95)
// packed first, scalar last
#define FOLDMIXR_A0_R0 P1(6,7,70,71) P1(8,9,72,73) P1(10,11,74,75) P1(12,13,76,77) P1(14,15,78,79) F1(0,64) F1(1,65) F1(2,66) F1(3,67) F1(4,68) F1(5,69)
#define FOLDMIXR_A1_R1 P1(22,23,86,87) P1(24,25,88,89) P1(26,27,90,91) P1(28,29,92,93) P1(30,31,94,95) F1(16,80) F1(17,81) F1(18,82) F1(19,83) F1(20,84) F1(21,85)

#define CLOBBERS                                                                                                               \
    "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20",   \
        "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71",    \
        "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90",    \
        "v91", "v92", "v93", "v94", "v95", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123",      \
        "v124", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "scc", "memory"

#define PROLOGUE                                                                                   \
    "s_mov_b32 s40, %[c]\n s_mov_b32 s41, %[c]\n s_mov_b32 s42, %[iters]\n"                      \
    "v_mov_b32 v124, 0x82828282\n"                                                               \
    "v_mov_b32 v112, %[f]\n v_mov_b32 v113, %[f]\n v_mov_b32 v114, %[f]\n v_mov_b32 v115, %[f]\n v_mov_b32 v116, %[f]\n v_mov_b32 v117, %[f]\n" \
    "v_mov_b32 v118, %[g]\n v_mov_b32 v119, %[g]\n v_mov_b32 v120, %[g]\n v_mov_b32 v121, %[g]\n v_mov_b32 v122, %[g]\n v_mov_b32 v123, %[g]\n" \
    "v_mov_b32 v0, 0\n v_mov_b32 v1, 0\n v_mov_b32 v2, 0\n v_mov_b32 v3, 0\n v_mov_b32 v4, 0\n v_mov_b32 v5, 0\n v_mov_b32 v6, 0\n v_mov_b32 v7, 0\n" \
    "v_mov_b32 v8, 0\n v_mov_b32 v9, 0\n v_mov_b32 v10, 0\n v_mov_b32 v11, 0\n v_mov_b32 v12, 0\n v_mov_b32 v13, 0\n v_mov_b32 v14, 0\n v_mov_b32 v15, 0\n" \
    "v_mov_b32 v16, 0\n v_mov_b32 v17, 0\n v_mov_b32 v18, 0\n v_mov_b32 v19, 0\n v_mov_b32 v20, 0\n v_mov_b32 v21, 0\n v_mov_b32 v22, 0\n v_mov_b32 v23, 0\n" \
    "v_mov_b32 v24, 0\n v_mov_b32 v25, 0\n v_mov_b32 v26, 0\n v_mov_b32 v27, 0\n v_mov_b32 v28, 0\n v_mov_b32 v29, 0\n v_mov_b32 v30, 0\n v_mov_b32 v31, 0\n" \
    "s_nop 4\n" MFMA("v[64:79]") MFMA("v[80:95]") "s_nop 15\n"                                   \
    "s_memtime s[44:45]\n s_waitcnt lgkmcnt(0)\n"                                                 \
    "1:\n"

#define EPILOGUE                                                                                   \
    "s_sub_u32 s42, s42, 1\n s_cmp_lg_u32 s42, 0\n s_cbranch_scc1 1b\n"                            \
    "s_nop 15\n s_memtime s[46:47]\n s_waitcnt lgkmcnt(0)\n"                                      \
    "s_sub_u32 %[t0], s46, s44\n s_subb_u32 %[t1], s47, s45\n"                                    \
    "v_add_f32 %[o], v0, v16\n v_add_f32 %[o], %[o], v15\n v_add_f32 %[o], %[o], v31\n"

// one loop iteration = 2 MFMAs (into R0 then R1) and the folds of the PREVIOUS results of the same registers:
//   MFMA -> R0 | fold R1 | MFMA -> R1 | fold R0 -- wait: the fold of R0 must sit >= 12 states behind the MFMA that writes R0.
// Layout used: fold(R0 of last iteration's first MFMA) ... see BODY macros: [MFMA->R0'][fold R1][MFMA->R1'][fold R0'] is illegal
// for short folds, so the streams below fold the OLDER result: [fold R0][MFMA->R0][fold R1][MFMA->R1], i.e. every result has
// one full fold + one MFMA (>= 9 instructions, padded to 12 with s_nop where needed) between its MFMA and its fold.
#define BODY(FA, FB, PAD) FA MFMA("v[64:79]") PAD FB MFMA("v[80:95]") PAD

template <int V>
__global__ __launch_bounds__(256) void k_probe(float *out, uint32_t *cycles, int iters, float c, int f, int g)
{
    float o;
    uint32_t t0, t1;
    if (V == 0)
        asm volatile(PROLOGUE BODY("", "", "") EPILOGUE : [o] "=v"(o), [t0] "=s"(t0), [t1] "=s"(t1) : [c] "s"(c), [iters] "s"(iters), [f] "v"(f), [g] "v"(g) : CLOBBERS);
    else if (V == 1)
        asm volatile(PROLOGUE BODY(FOLD16_A0_R0, FOLD16_A1_R1, "") EPILOGUE : [o] "=v"(o), [t0] "=s"(t0), [t1] "=s"(t1) : [c] "s"(c), [iters] "s"(iters), [f] "v"(f), [g] "v"(g) : CLOBBERS);
    else if (V == 2)
        asm volatile(PROLOGUE BODY(FOLDPK_A0_R0, FOLDPK_A1_R1, "s_nop 2\n") EPILOGUE : [o] "=v"(o), [t0] "=s"(t0), [t1] "=s"(t1) : [c] "s"(c), [iters] "s"(iters), [f] "v"(f), [g] "v"(g) : CLOBBERS);
    else if (V == 3)
        asm volatile(PROLOGUE BODY(FOLDMIX4_A0_R0, FOLDMIX4_A1_R1, "s_nop 0\n") EPILOGUE : [o] "=v"(o), [t0] "=s"(t0), [t1] "=s"(t1) : [c] "s"(c), [iters] "s"(iters), [f] "v"(f), [g] "v"(g) : CLOBBERS);
    else if (V == 4)
        asm volatile(PROLOGUE BODY(FOLDMIX6_A0_R0, FOLDMIX6_A1_R1, "") EPILOGUE : [o] "=v"(o), [t0] "=s"(t0), [t1] "=s"(t1) : [c] "s"(c), [iters] "s"(iters), [f] "v"(f), [g] "v"(g) : CLOBBERS);
    else if (V == 5)
        asm volatile(PROLOGUE BODY(FOLDMIXR_A0_R0, FOLDMIXR_A1_R1, "") EPILOGUE : [o] "=v"(o), [t0] "=s"(t0), [t1] "=s"(t1) : [c] "s"(c), [iters] "s"(iters), [f] "v"(f), [g] "v"(g) : CLOBBERS);
    else if (V == 6)      // MFMA first, then the fold of the OTHER register set's result (the fold sits in this MFMA's shadow)
        asm volatile(PROLOGUE MFMA("v[64:79]") FOLD16_A1_R1 MFMA("v[80:95]") FOLD16_A0_R0 EPILOGUE : [o] "=v"(o), [t0] "=s"(t0), [t1] "=s"(t1) : [c] "s"(c), [iters] "s"(iters), [f] "v"(f), [g] "v"(g) : CLOBBERS);
    else if (V == 7)
        asm volatile(PROLOGUE MFMA("v[64:79]") FOLDPK_A1_R1 "s_nop 2\n" MFMA("v[80:95]") FOLDPK_A0_R0 "s_nop 2\n" EPILOGUE : [o] "=v"(o), [t0] "=s"(t0), [t1] "=s"(t1) : [c] "s"(c), [iters] "s"(iters), [f] "v"(f), [g] "v"(g) : CLOBBERS);
    else if (V == 8)
        asm volatile(PROLOGUE MFMA("v[64:79]") FOLDMIX6_A1_R1 MFMA("v[80:95]") FOLDMIX6_A0_R0 EPILOGUE : [o] "=v"(o), [t0] "=s"(t0), [t1] "=s"(t1) : [c] "s"(c), [iters] "s"(iters), [f] "v"(f), [g] "v"(g) : CLOBBERS);
    else if (V == 9)      // the fold alone, no MFMA: the VALU floor of 16 fmas / 8 packed fmas
        asm volatile(PROLOGUE FOLD16_A0_R0 FOLD16_A1_R1 EPILOGUE : [o] "=v"(o), [t0] "=s"(t0), [t1] "=s"(t1) : [c] "s"(c), [iters] "s"(iters), [f] "v"(f), [g] "v"(g) : CLOBBERS);
    else if (V == 10)
        asm volatile(PROLOGUE FOLDPK_A0_R0 FOLDPK_A1_R1 EPILOGUE : [o] "=v"(o), [t0] "=s"(t0), [t1] "=s"(t1) : [c] "s"(c), [iters] "s"(iters), [f] "v"(f), [g] "v"(g) : CLOBBERS);
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    out[gid] = o;
    if ((threadIdx.x & 63) == 0) cycles[gid >> 6] = t0;
    (void)t1;
}

template <int V>
static void run(const char *name, float *dout, uint32_t *dcyc, int wg_per_cu)
{
    const int iters = 4000, blocks = 256 * wg_per_cu, waves = blocks * 4;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k_probe<V>, dim3(blocks), dim3(256), 0, 0, dout, dcyc, 50, 0.5f, 0x05030107, 0x02060401);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_probe<V>, dim3(blocks), dim3(256), 0, 0, dout, dcyc, iters, 0.5f, 0x05030107, 0x02060401);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    uint32_t *h = (uint32_t *)malloc(waves * 4);
    hipMemcpy(h, dcyc, waves * 4, hipMemcpyDeviceToHost);
    double sum = 0;
    uint32_t mx = 0;
    for (int i = 0; i < waves; i++) { sum += h[i]; mx = h[i] > mx ? h[i] : mx; }
    free(h);
    const double per_wave = sum / waves / iters / 2.0;          // cycles per MFMA slot as one wave sees it
    const double per_simd = per_wave / wg_per_cu;               // wg_per_cu waves share a SIMD
    printf("%-44s waves/SIMD=%d  %.3f ms  cycles per MFMA-unit: per wave %.1f, per SIMD %.1f  (wall: %.1f ns per unit per SIMD)  -> 8192^3: %.3f ms at 2.4 GHz\n", name,
           wg_per_cu, ms, per_wave, per_simd, ms * 1e6 / (iters * 2.0 * wg_per_cu), per_simd * 8192.0 / 2.4e9 * 1e3 * 1.0);
    (void)mx;
}

int main()
{
    float *dout;
    uint32_t *dcyc;
    hipMalloc(&dout, 256 * 8 * 256 * 4);
    hipMalloc(&dcyc, 256 * 8 * 4 * 4);
    for (int w = 1; w <= 4; w++) {
        run<0>("MFMA only", dout, dcyc, w);
        run<9>("fold only: 16 v_fma_f32", dout, dcyc, w);
        run<10>("fold only: 8 v_pk_fma_f32", dout, dcyc, w);
        run<1>("fold16 scalar | MFMA", dout, dcyc, w);
        run<2>("fold8 packed (+nop2) | MFMA", dout, dcyc, w);
        run<3>("4 scalar + 6 packed (+nop0) | MFMA", dout, dcyc, w);
        run<4>("6 scalar + 5 packed | MFMA", dout, dcyc, w);
        run<5>("5 packed + 6 scalar | MFMA", dout, dcyc, w);
        run<6>("MFMA | fold16 scalar (other set)", dout, dcyc, w);
        run<7>("MFMA | fold8 packed (other set, +nop2)", dout, dcyc, w);
        run<8>("MFMA | 6 scalar + 5 packed (other set)", dout, dcyc, w);
        printf("\n");
    }
    return 0;
}
