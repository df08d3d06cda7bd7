// gemm4.hip -- CloverMatrix4 x CloverMatrix4^T -> fp32 on the gfx950 matrix cores, int8 MFMA version.
// NOT the default any more: gemm6.hip (FP6 block-scaled MFMA on exactly representable operands, one fma per element) is
// a third faster; this kernel runs under CLV_GEMM_KERNEL=i8 for A/B measurements and keeps its parity test.
//
// The reference has no GEMM; semantics are defined in oracle/clover4_oracle.h / DESIGN.md 6:
//   C[i][j] = fold_b fmaf(c_b, (float)S_b, C),  S_b = exact int32 sum of the 64 nibble products of K-block b,
//   c_b = f32(f32(sA[i>>6][b] * 1/49) * sB[j>>6][b]).
// One K-block (64 elements, one Clover scale block) is exactly one v_mfma_i32_16x16x64_i8.
//
// Mapping
//   workgroup 512 threads = 2x4 waves, tile 128x128; wave tile 64x32 = 4x2 MFMA tiles inside ONE scale tile
//   of A and of B, so c_b is wave-uniform.
//   staging: global (packed nibbles, 64 B per row per stage = 2 K-blocks) -> registers -> unpack -> LDS int8,
//   double-buffered.  Unpacking needs no sign extension: (w & 0xF0F0F0F0) holds 16*q of the high nibbles and
//   ((w << 4) & 0xF0F0F0F0) 16*q of the low nibbles as int8, so the MFMA returns 256*S_b exactly; the 2^-8
//   is folded into c_b (exact power-of-two scaling; guarded against underflow).  The K order inside a
//   block is permuted (hi nibbles, then lo nibbles per dword) identically for A and B: the integer sum is
//   order-free.
//   LDS: [kblock][row][64 B] int8, 16-byte slots XOR-swizzled with f(row>>2) = {0,2,3,1} so that each
//   ds_read_b128 lane group (MI355X_MICROARCH.md LDS table) touches 16 distinct slots: conflict-free.
//   epilogue per K-block and accumulator element: the MFMA accumulates onto the constant 0x4B400000 (the bit
//   pattern of 12582912.0f = 1.5 * 2^23, where one ulp is 1), so its int32 result, read as fp32, IS the float
//   12582912 + 256*S_b exactly (|256*S_b| <= 802816 < 2^20 keeps the exponent).  One exact v_sub_f32 replaces
//   the half-rate v_cvt_f32_i32 (measured on this chip: cvt 4.3 cycles, fma/sub 2.8 cycles per wave64), then
//   an fma folds it in; both as packed fp32 pairs (fold4): 4 VALU per MFMA instead of 4 half-rate + 4 full-rate.
#include "common.h"

typedef int i32x4 __attribute__((ext_vector_type(4)));

#define GM_BIAS_BITS 0x4B400000          // 12582912.0f
#define GM_BIAS_F 12582912.0f
#define GM_BIAS4 (i32x4{GM_BIAS_BITS, GM_BIAS_BITS, GM_BIAS_BITS, GM_BIAS_BITS})

// fold the 4 results one lane holds of a 16x16 MFMA tile: acc[t] = fma(c8, raw[t] - bias, acc[t]) as two packed-fp32
// pairs (v_pk_add_f32 + v_pk_fma_f32: 4 VALU instructions per MFMA instead of 8, each component IEEE-exact as before)
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void fold4(float (&acc)[4], const i32x4 raw, float c8)
{
    const f32x2 bias = {GM_BIAS_F, GM_BIAS_F}, cc = {c8, c8};
    const f32x2 lo = {__int_as_float(raw.x), __int_as_float(raw.y)}, hi = {__int_as_float(raw.z), __int_as_float(raw.w)};
    f32x2 a0 = {acc[0], acc[1]}, a1 = {acc[2], acc[3]};
    a0 = __builtin_elementwise_fma(cc, lo - bias, a0);
    a1 = __builtin_elementwise_fma(cc, hi - bias, a1);
    acc[0] = a0.x; acc[1] = a0.y; acc[2] = a1.x; acc[3] = a1.y;
}

#define GM_TILE 128

// f(row>>2) = {0,2,3,1} makes every ds_read_b128 lane group conflict-free; the extra XOR with the K-block index
// (constant per read) separates the kb=0 / kb=1 pieces that one 8-lane ds_write_b128 group stores together.
__device__ __forceinline__ int swz(int row, int kb) { return ((0x1320 >> (4 * ((row >> 2) & 3))) ^ kb) & 3; }

// 16 packed bytes (32 nibbles) -> two 16-byte int8 slots (each nibble as 16*q)
__device__ __forceinline__ void unpack32(const u32x4 p, u32x4 &s0, u32x4 &s1)
{
    const uint32_t M = 0xF0F0F0F0u;
    s0 = u32x4{p.x & M, (p.x << 4) & M, p.y & M, (p.y << 4) & M};
    s1 = u32x4{p.z & M, (p.z << 4) & M, p.w & M, (p.w << 4) & M};
}

// One LDS stage = 2 K-blocks (one barrier per stage), single result set, 4 waves per SIMD.  Variants measured and dropped in
// round 1: two result sets ping-ponged at 2 waves/SIMD (slower), 4 K-blocks per barrier (no gain), a PACKED
// LDS tile widened after the fragment read (0.90 vs 0.81 ms: every wave widens its own fragments, VALU-bound).
#define GM_KBS 2
__global__ __launch_bounds__(512, 4) void k_m4_gemm_mfma(const uint8_t *__restrict__ A, const float *__restrict__ sA,
                                                         const uint8_t *__restrict__ B, const float *__restrict__ sB,
                                                         uint64_t M, uint64_t N, uint64_t K, float *__restrict__ C,
                                                         uint32_t tiles_m, uint32_t tiles_n)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int GM_STAGE_BYTES = 2 * GM_KBS * GM_TILE * 64;  // A + B, 64 int8 bytes per row and K-block

    // ---- tile assignment: XCD-aware (block b runs on XCD b % 8): give each XCD a contiguous range of tiles,
    // walked in 8-wide column groups so neighbours share A rows / B columns in that XCD's L2
    const uint32_t nwg = tiles_m * tiles_n;
    uint32_t id = blockIdx.x;
    {
        const uint32_t q = nwg / 8, r = nwg % 8, xcd = id % 8, s = id / 8;
        id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + s;       // bijective for any nwg
    }
    const uint32_t GROUP = 8;
    const uint32_t per_group = GROUP * tiles_n;
    const uint32_t group = id / per_group;
    const uint32_t first_m = group * GROUP;
    const uint32_t gsize = (tiles_m - first_m) < GROUP ? (tiles_m - first_m) : GROUP;
    const uint32_t tm = first_m + (id % per_group) % gsize;
    const uint32_t tn = (id % per_group) / gsize;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;                   // 2 x 4 waves, each 64 rows x 32 columns
    const uint64_t m0 = (uint64_t)tm * GM_TILE, n0 = (uint64_t)tn * GM_TILE;
    const uint64_t kbn = K / 64;                               // K-blocks
    const uint64_t nstages = kbn / GM_KBS;                     // K is a multiple of 128

    // staging role: one (row, 16-byte piece) per operand per stage; a row holds 64 packed bytes per stage
    const int srow = tid / (2 * GM_KBS), spiece = tid % (2 * GM_KBS);
    u32x4 pa, pb;
    auto fetch = [&](uint64_t st) {
        const uint64_t off = st * (32 * GM_KBS) + 16 * spiece;
        pa = *reinterpret_cast<const u32x4 *>(A + (m0 + srow) * (K / 2) + off);
        pb = *reinterpret_cast<const u32x4 *>(B + (n0 + srow) * (K / 2) + off);
    };
    auto stash = [&](int buf) {
        char *base = smem + buf * GM_STAGE_BYTES;
        auto put = [&](char *tile, const u32x4 p) {
            const int kb = spiece >> 1, half = spiece & 1;
            u32x4 s0, s1;
            unpack32(p, s0, s1);
            char *r = tile + (kb * GM_TILE + srow) * 64;
            const int f = swz(srow, kb);
            *reinterpret_cast<u32x4 *>(r + (((2 * half) ^ f) << 4)) = s0;
            *reinterpret_cast<u32x4 *>(r + (((2 * half + 1) ^ f) << 4)) = s1;
        };
        put(base, pa);
        put(base + GM_KBS * GM_TILE * 64, pb);
    };

    float acc[4][2][4];
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int t = 0; t < 4; t++) acc[a][b][t] = 0.0f;

    const float *sArow = sA + ((m0 >> 6) + wr) * kbn;
    const float *sBrow = sB + ((n0 >> 6) + (wc >> 1)) * kbn;
    const int frow = lane & 15, fkg = lane >> 4;

    // MFMAs of a K-block, then its own fold; the MFMA->VALU dependency is covered by the other resident waves.  The
    // fragments of K-block kb+1 are requested from LDS right after the MFMAs of kb were issued, so their latency hides
    // behind the fold of kb.
    i32x4 S[4][2], fa[4], fb[2];
    auto frag = [&](const char *tile, int kb, int row) -> i32x4 {
        return *reinterpret_cast<const i32x4 *>(tile + (kb * GM_TILE + row) * 64 + ((fkg ^ swz(row, kb)) << 4));
    };
    auto load_frags = [&](const char *tA, const char *tB, int kb) {
#pragma unroll
        for (int a = 0; a < 4; a++) fa[a] = frag(tA, kb, wr * 64 + a * 16 + frow);
#pragma unroll
        for (int b = 0; b < 2; b++) fb[b] = frag(tB, kb, wc * 32 + b * 16 + frow);
    };
    auto mfma_all = [&]() {
#pragma unroll
        for (int a = 0; a < 4; a++)
#pragma unroll
            for (int b = 0; b < 2; b++) S[a][b] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa[a], fb[b], GM_BIAS4, 0, 0, 0);
    };
    auto fold_all = [&](float c) {
        const bool normal = __builtin_fabsf(c) >= 1.0e-30f || c == 0.0f;   // c / 256 must not lose bits: ONE uniform branch per K-block
        const float c8 = c * 0.00390625f;
        if (normal) {
#pragma unroll
            for (int a = 0; a < 4; a++)
#pragma unroll
                for (int b = 0; b < 2; b++) fold4(acc[a][b], S[a][b], c8);
        } else {
#pragma unroll
            for (int a = 0; a < 4; a++)
#pragma unroll
                for (int b = 0; b < 2; b++)
#pragma unroll
                    for (int t = 0; t < 4; t++) acc[a][b][t] = __builtin_fmaf(c, (float)((S[a][b][t] - GM_BIAS_BITS) >> 8), acc[a][b][t]);
        }
    };

    fetch(0);
    stash(0);
    __syncthreads();

    for (uint64_t st = 0; st < nstages; st++) {
        const int buf = (int)(st & 1);
        if (st + 1 < nstages) fetch(st + 1);
        const char *tA = smem + buf * GM_STAGE_BYTES;
        const char *tB = tA + GM_KBS * GM_TILE * 64;
        const uint64_t blk = st * GM_KBS;
        const float c0 = (sArow[blk] * CLV_RCP49) * sBrow[blk];
        const float c1 = (sArow[blk + 1] * CLV_RCP49) * sBrow[blk + 1];
        load_frags(tA, tB, 0);
        mfma_all();
        load_frags(tA, tB, 1);
        fold_all(c0);
        mfma_all();
        fold_all(c1);
        if (st + 1 < nstages) stash(buf ^ 1);
        __syncthreads();
    }

    // C/D layout of the 16x16 MFMA: column = lane & 15, row = 4 * (lane >> 4) + t
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const uint64_t i = m0 + wr * 64 + a * 16 + 4 * (lane >> 4) + t;
                const uint64_t j = n0 + wc * 32 + b * 16 + (lane & 15);
                __builtin_nontemporal_store(acc[a][b][t], &C[i * N + j]);
            }
}

int clm4_gemm_mfma(const int8_t *A, const float *sA, uint64_t M, uint64_t K, const int8_t *B, const float *sB, uint64_t N, float *C,
                   hipStream_t st)
{
    const uint32_t tiles_m = (uint32_t)(M / GM_TILE), tiles_n = (uint32_t)(N / GM_TILE);
    const size_t lds = 2 * (size_t)(2 * GM_KBS * GM_TILE * 64);
    CLV_HIP(hipFuncSetAttribute((const void *)k_m4_gemm_mfma, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_m4_gemm_mfma, dim3(tiles_m * tiles_n), dim3(512), lds, st, (const uint8_t *)A, sA, (const uint8_t *)B, sB, M, N, K, C,
                       tiles_m, tiles_n);
    CLV_LAUNCH_CHECK();
    return CLV_OK;
}
