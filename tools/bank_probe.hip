// bank_probe.hip -- does the fp32 fold of the GEMM loop (v_fma_f32 acc, s_c, v_result, acc) run slower because accumulator and result
// registers share a VGPR bank?  Explicit physical registers, 16 fmas per unit as in tools/gen_gemm6_loop256.py.  Standalone tool:
//   hipcc --offload-arch=gfx950 -O2 tools/bank_probe.hip -o tools/bank_probe && tools/bank_probe
#include <hip/hip_runtime.h>
#include <stdio.h>

#define CLOB "v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17", \
             "v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75","v76","v77","v78","v79","v80","v81","v82","v83","v100","v101","v102","v103","s4","s5"

// R = base register of the "result set", C = operand 1 (the factor)
#define FMA(i, R, C) "v_fma_f32 v" #i ", " C ", v[" #R "+" #i "], v" #i "\n"
#define FOLD16(R, C) FMA(0,R,C) FMA(1,R,C) FMA(2,R,C) FMA(3,R,C) FMA(4,R,C) FMA(5,R,C) FMA(6,R,C) FMA(7,R,C) \
                     FMA(8,R,C) FMA(9,R,C) FMA(10,R,C) FMA(11,R,C) FMA(12,R,C) FMA(13,R,C) FMA(14,R,C) FMA(15,R,C)
#define PK(i, R, C) "v_pk_fma_f32 v[" #i ":" #i "+1], " C ", v[" #R "+" #i ":" #R "+" #i "+1], v[" #i ":" #i "+1] op_sel_hi:[0,1,1]\n"
#define PKV(i, R, C) "v_pk_fma_f32 v[" #i ":" #i "+1], " C ", v[" #R "+" #i ":" #R "+" #i "+1], v[" #i ":" #i "+1]\n"
#define FOLD8PK(R, C) PK(0,R,C) PK(2,R,C) PK(4,R,C) PK(6,R,C) PK(8,R,C) PK(10,R,C) PK(12,R,C) PK(14,R,C)
#define FOLD8PKV(R, C) PKV(0,R,C) PKV(2,R,C) PKV(4,R,C) PKV(6,R,C) PKV(8,R,C) PKV(10,R,C) PKV(12,R,C) PKV(14,R,C)
#define FMAC(i, R, C) "v_fmac_f32 v" #i ", " C ", v[" #R "+" #i "]\n"
#define FOLD16MAC(R, C) FMAC(0,R,C) FMAC(1,R,C) FMAC(2,R,C) FMAC(3,R,C) FMAC(4,R,C) FMAC(5,R,C) FMAC(6,R,C) FMAC(7,R,C) \
                        FMAC(8,R,C) FMAC(9,R,C) FMAC(10,R,C) FMAC(11,R,C) FMAC(12,R,C) FMAC(13,R,C) FMAC(14,R,C) FMAC(15,R,C)

// ---- the fold beside the MFMA that feeds the NEXT fold (two result sets, fold one unit behind, as in the product) -----------------
#define CLOB2 CLOB,"v84","v85","v86","v87","v88","v89","v90","v91","v92","v93","v94","v95","v96","v97","v98","v99","v104","v105","v106","v107", \
              "v108","v109","v110","v111","v112","v113","v120"
#define MFMA(D) "v_mfma_scale_f32_32x32x64_f8f6f4 v[" #D ":" #D "+15], v[96:101], v[108:113], 0, v120, v120 op_sel_hi:[0,0,0] cbsz:2 blgp:2\n"
#define SC(i, R, C) FMA(i, R, C)
#define F_8S4P(R, CS, CP) FMA(0,R,CS) FMA(1,R,CS) FMA(2,R,CS) FMA(3,R,CS) FMA(4,R,CS) FMA(5,R,CS) FMA(6,R,CS) FMA(7,R,CS) PK(8,R,CP) PK(10,R,CP) PK(12,R,CP) PK(14,R,CP)
#define F_12S2P(R, CS, CP) FMA(0,R,CS) FMA(1,R,CS) FMA(2,R,CS) FMA(3,R,CS) FMA(4,R,CS) FMA(5,R,CS) FMA(6,R,CS) FMA(7,R,CS) FMA(8,R,CS) FMA(9,R,CS) FMA(10,R,CS) FMA(11,R,CS) PK(12,R,CP) PK(14,R,CP)
#define F_4S6P(R, CS, CP) FMA(0,R,CS) FMA(1,R,CS) FMA(2,R,CS) FMA(3,R,CS) PK(4,R,CP) PK(6,R,CP) PK(8,R,CP) PK(10,R,CP) PK(12,R,CP) PK(14,R,CP)
// one unit = MFMA into set X, fold of set Y (the previous MFMA's)
#define UNIT2(FOLDA, FOLDB) MFMA(64) FOLDB MFMA(80) FOLDA

template <int MODE>
__global__ __launch_bounds__(256) void km(float *out, int iters)
{
    asm volatile("s_mov_b32 s4, 0x3f800001\n s_mov_b32 s5, 0x3f800001\n v_mov_b32 v100, 0x3f800001\n v_mov_b32 v101, 0x3f800001\n v_mov_b32 v120, 0x82828282\n"
                 "v_mov_b32 v96, 0x12345678\n v_mov_b32 v97, 0x9abcdef0\n v_mov_b32 v98, 0x0fedcba9\n v_mov_b32 v99, 0x87654321\n v_mov_b32 v108, 0x13572468\n v_mov_b32 v109, 0xa5a5c3c3\n"
                 "v_mov_b32 v110, 0x5a5a3c3c\n v_mov_b32 v111, 0x0f1e2d3c\n v_mov_b32 v112, 0x4b5a6978\n v_mov_b32 v113, 0x8796a5b4\n" ::: CLOB2);
    for (int it = 0; it < iters; it++) {
        if (MODE == 0) asm volatile(UNIT2(F_8S4P(64, "s4", "s[4:5]"), F_8S4P(80, "s4", "s[4:5]")) ::: CLOB2);              // the product
        if (MODE == 1) asm volatile(UNIT2(F_8S4P(64, "v100", "s[4:5]"), F_8S4P(80, "v100", "s[4:5]")) ::: CLOB2);          // scalar fmas take the factor from a VGPR
        if (MODE == 2) asm volatile(UNIT2(FOLD16(64, "v100"), FOLD16(80, "v100")) ::: CLOB2);
        if (MODE == 3) asm volatile(UNIT2(FOLD16(64, "s4"), FOLD16(80, "s4")) ::: CLOB2);
        if (MODE == 4) asm volatile(UNIT2(FOLD8PK(64, "s[4:5]"), FOLD8PK(80, "s[4:5]")) ::: CLOB2);
        if (MODE == 5) asm volatile(UNIT2(F_12S2P(64, "v100", "s[4:5]"), F_12S2P(80, "v100", "s[4:5]")) ::: CLOB2);
        if (MODE == 6) asm volatile(UNIT2(F_4S6P(64, "v100", "s[4:5]"), F_4S6P(80, "v100", "s[4:5]")) ::: CLOB2);
        if (MODE == 7) asm volatile(MFMA(64) MFMA(80) ::: CLOB2);
    }
    float r;
    asm volatile("v_add_f32 %0, v0, v15" : "=v"(r) :: CLOB2);
    if (r == 12345.678f) out[0] = r;
}

template <int MODE>
static void runm(const char *name, int waves_per_simd)
{
    float *out;
    (void)hipMalloc(&out, 4);
    const int iters = 4000;
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    const int blocks = 256 * waves_per_simd;
    km<MODE><<<blocks, 256>>>(out, 10);
    (void)hipEventRecord(a);
    km<MODE><<<blocks, 256>>>(out, iters);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms;
    (void)hipEventElapsedTime(&ms, a, b);
    const double ns = ms * 1e6 / ((double)iters * 2 * waves_per_simd);
    printf("%-58s waves/SIMD=%d  %.2f ns per MFMA unit per SIMD -> 8192^3 arithmetic %.3f ms\n", name, waves_per_simd, ns, ns * 8192 * 1e-6);
    (void)hipFree(out);
}

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters)
{
    asm volatile("s_mov_b32 s4, 0x3f800001\n s_mov_b32 s5, 0x3f800001\n v_mov_b32 v100, 0x3f800001\n v_mov_b32 v101, 0x3f800001\n v_mov_b32 v102, 0x3f800001\n v_mov_b32 v103, 0x3f800001\n" ::: CLOB);
    for (int it = 0; it < iters; it++) {
        if (MODE == 0) asm volatile(FOLD16(64, "s4") FOLD16(64, "s4") ::: CLOB);          // the product's form: acc v0.., result v64..: same bank
        if (MODE == 1) asm volatile(FOLD16(66, "s4") FOLD16(66, "s4") ::: CLOB);          // result set two registers on: banks differ by 2
        if (MODE == 2) asm volatile(FOLD16(65, "s4") FOLD16(65, "s4") ::: CLOB);          // by 1
        if (MODE == 3) asm volatile(FOLD16(64, "v100") FOLD16(64, "v100") ::: CLOB);      // factor in a VGPR of the same bank as both
        if (MODE == 4) asm volatile(FOLD16(66, "v101") FOLD16(66, "v101") ::: CLOB);      // three different banks
        if (MODE == 5) asm volatile(FOLD8PK(64, "s[4:5]") FOLD8PK(64, "s[4:5]") ::: CLOB);
        if (MODE == 6) asm volatile(FOLD8PK(66, "s[4:5]") FOLD8PK(66, "s[4:5]") ::: CLOB);
        if (MODE == 7) asm volatile(FOLD8PKV(66, "v[100:101]") FOLD8PKV(66, "v[100:101]") ::: CLOB);
        if (MODE == 8) asm volatile(FOLD16MAC(64, "s4") FOLD16MAC(64, "s4") ::: CLOB);
        if (MODE == 9) asm volatile(FOLD16MAC(66, "s4") FOLD16MAC(66, "s4") ::: CLOB);
        if (MODE == 10) asm volatile(FOLD16MAC(66, "v101") FOLD16MAC(66, "v101") ::: CLOB);
    }
    float r;
    asm volatile("v_add_f32 %0, v0, v15" : "=v"(r) :: CLOB);
    if (r == 12345.678f) out[0] = r;
}

template <int MODE>
static void run(const char *name, int waves_per_simd)
{
    float *out;
    (void)hipMalloc(&out, 4);
    const int iters = 4000;
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    const int blocks = 256 * waves_per_simd;
    k<MODE><<<blocks, 256>>>(out, 10);
    (void)hipEventRecord(a);
    k<MODE><<<blocks, 256>>>(out, iters);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms;
    (void)hipEventElapsedTime(&ms, a, b);
    const double folds = (double)iters * 2;                         // folds (of 16 fmas) per wave
    const double ns = ms * 1e6 / (folds * waves_per_simd);
    printf("%-58s waves/SIMD=%d  %.2f ns per fold per SIMD = %.1f cycles at 2.4 GHz (%.2f per fma)\n", name, waves_per_simd, ns, ns * 2.4, ns * 2.4 / 16);
    (void)hipFree(out);
}

int main()
{
    for (int w : {4}) {
        runm<7>("MFMA only", w);
        runm<0>("MFMA | 8 v_fma(s) + 4 v_pk_fma(s)   [the product]", w);
        runm<1>("MFMA | 8 v_fma(v) + 4 v_pk_fma(s)", w);
        runm<5>("MFMA | 12 v_fma(v) + 2 v_pk_fma(s)", w);
        runm<6>("MFMA | 4 v_fma(v) + 6 v_pk_fma(s)", w);
        runm<2>("MFMA | 16 v_fma(v)", w);
        runm<3>("MFMA | 16 v_fma(s)", w);
        runm<4>("MFMA | 8 v_pk_fma(s)", w);
    }
    for (int w : {4}) {
        run<0>("16 v_fma acc, s, res(v64..: same bank as acc), acc", w);
        run<1>("16 v_fma acc, s, res(v66..: bank + 2), acc", w);
        run<2>("16 v_fma acc, s, res(v65..: bank + 1), acc", w);
        run<3>("16 v_fma acc, v100, res(v64..), acc", w);
        run<4>("16 v_fma acc, v101, res(v66..), acc", w);
        run<5>("8 v_pk_fma acc, s[4:5], res(v64..), acc", w);
        run<6>("8 v_pk_fma acc, s[4:5], res(v66..), acc", w);
        run<7>("8 v_pk_fma acc, v[100:101], res(v66..), acc", w);
        run<8>("16 v_fmac acc, s, res(v64..)", w);
        run<9>("16 v_fmac acc, s, res(v66..)", w);
        run<10>("16 v_fmac acc, v101, res(v66..)", w);
    }
    return 0;
}
