// gemm4.hip -- CloverMatrix4 x CloverMatrix4^T -> fp32 on the gfx950 matrix cores.
//
// The reference has no GEMM; semantics are defined in oracle/clover4_oracle.h / DESIGN.md 6:
//   C[i][j] = fold_b fmaf(c_b, (float)S_b, C),  S_b = exact int32 sum of the 64 nibble products of K-block b,
//   c_b = f32(f32(sA[i>>6][b] * 1/49) * sB[j>>6][b]).
// One K-block (64 elements, one Clover scale block) is exactly one v_mfma_i32_16x16x64_i8.
//
// Mapping
//   workgroup 256 threads = 2x2 waves, tile 128x128; wave tile 64x64 = 4x4 MFMA tiles = ONE scale tile of A
//   and of B, so c_b is wave-uniform.
//   staging: global (packed nibbles, 64 B per row per stage = 2 K-blocks) -> registers -> unpack -> LDS int8,
//   double-buffered.  Unpacking needs no sign extension: (w & 0xF0F0F0F0) holds 16*q of the high nibbles and
//   ((w << 4) & 0xF0F0F0F0) 16*q of the low nibbles as int8, so the MFMA returns 256*S_b exactly; the 2^-8
//   is folded into c_b (exact power-of-two scaling; guarded against underflow).  The K order inside a
//   block is permuted (hi nibbles, then lo nibbles per dword) identically for A and B: the integer sum is
//   order-free.
//   LDS: [kblock][row][64 B] int8, 16-byte slots XOR-swizzled with f(row>>2) = {0,2,3,1} so that each
//   ds_read_b128 lane group (MI355X_MICROARCH.md LDS table) touches 16 distinct slots: conflict-free.
//   epilogue per K-block and accumulator element: v_cvt_f32_i32 + v_fma_f32 (VALU beside the MFMA pipe).
#include "common.h"

typedef int i32x4 __attribute__((ext_vector_type(4)));

#define GM_TILE 128
#define GM_KB_PER_STAGE 2
#define GM_STAGE_BYTES (2 * GM_KB_PER_STAGE * GM_TILE * 64)   // A + B, int8: 32 KiB

__device__ __forceinline__ int swz(int row) { return (0x1320 >> (4 * ((row >> 2) & 3))) & 3; }   // f = {0,2,3,1}

// 16 packed bytes (32 nibbles) -> two 16-byte int8 slots (each nibble as 16*q)
__device__ __forceinline__ void unpack32(const u32x4 p, u32x4 &s0, u32x4 &s1)
{
    const uint32_t M = 0xF0F0F0F0u;
    s0 = u32x4{p.x & M, (p.x << 4) & M, p.y & M, (p.y << 4) & M};
    s1 = u32x4{p.z & M, (p.z << 4) & M, p.w & M, (p.w << 4) & M};
}

__global__ __launch_bounds__(256, 2) void k_m4_gemm_mfma(const uint8_t *__restrict__ A, const float *__restrict__ sA,
                                                         const uint8_t *__restrict__ B, const float *__restrict__ sB,
                                                         uint64_t M, uint64_t N, uint64_t K, float *__restrict__ C,
                                                         uint32_t tiles_m, uint32_t tiles_n)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];

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
    const int wr = wave >> 1, wc = wave & 1;
    const uint64_t m0 = (uint64_t)tm * GM_TILE, n0 = (uint64_t)tn * GM_TILE;
    const uint64_t kbn = K / 64;                               // K-blocks
    const uint64_t nstages = kbn / GM_KB_PER_STAGE;            // K is a multiple of 128

    // staging role: 2 x (row, quarter) per operand per stage
    const int srow0 = tid >> 2, squarter = tid & 3;            // rows srow0 and srow0 + 64
    const uint8_t *Ag = A + (m0 + srow0) * (K / 2) + 16 * squarter;
    const uint8_t *Bg = B + (n0 + srow0) * (K / 2) + 16 * squarter;
    const uint64_t row64 = 64 * (K / 2);

    u32x4 pa0, pa1, pb0, pb1;
    auto fetch = [&](uint64_t st) {
        const uint64_t off = st * 64;                          // 64 packed bytes per row per stage
        pa0 = *reinterpret_cast<const u32x4 *>(Ag + off);
        pa1 = *reinterpret_cast<const u32x4 *>(Ag + off + row64);
        pb0 = *reinterpret_cast<const u32x4 *>(Bg + off);
        pb1 = *reinterpret_cast<const u32x4 *>(Bg + off + row64);
    };
    auto stash = [&](int buf) {
        char *base = smem + buf * GM_STAGE_BYTES;
        const int kb = squarter >> 1, half = squarter & 1;
        auto put = [&](char *tile, int row, const u32x4 p) {
            u32x4 s0, s1;
            unpack32(p, s0, s1);
            char *r = tile + (kb * GM_TILE + row) * 64;
            const int f = swz(row);
            *reinterpret_cast<u32x4 *>(r + (((2 * half) ^ f) << 4)) = s0;
            *reinterpret_cast<u32x4 *>(r + (((2 * half + 1) ^ f) << 4)) = s1;
        };
        put(base, srow0, pa0);
        put(base, srow0 + 64, pa1);
        put(base + GM_KB_PER_STAGE * GM_TILE * 64, srow0, pb0);
        put(base + GM_KB_PER_STAGE * GM_TILE * 64, srow0 + 64, pb1);
    };

    float acc[4][4][4];
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 4; b++)
#pragma unroll
            for (int t = 0; t < 4; t++) acc[a][b][t] = 0.0f;

    const float *sArow = sA + ((m0 >> 6) + wr) * kbn;
    const float *sBrow = sB + ((n0 >> 6) + wc) * kbn;
    const int frow = lane & 15, fkg = lane >> 4;

    fetch(0);
    stash(0);
    __syncthreads();

    for (uint64_t st = 0; st < nstages; st++) {
        const int buf = (int)(st & 1);
        if (st + 1 < nstages) fetch(st + 1);
        const char *tA = smem + buf * GM_STAGE_BYTES;
        const char *tB = tA + GM_KB_PER_STAGE * GM_TILE * 64;
#pragma unroll
        for (int kb = 0; kb < GM_KB_PER_STAGE; kb++) {
            const uint64_t blk = st * GM_KB_PER_STAGE + kb;
            const float c = (sArow[blk] * CLV_RCP49) * sBrow[blk];
            const float c8 = c * 0.00390625f;                               // 2^-8, exact unless it underflows
            const bool tiny = __builtin_fabsf(c) < 1.0e-30f;                // wave-uniform
            i32x4 fa[4], fb[4];
#pragma unroll
            for (int a = 0; a < 4; a++) {
                const int row = wr * 64 + a * 16 + frow;
                fa[a] = *reinterpret_cast<const i32x4 *>(tA + (kb * GM_TILE + row) * 64 + ((fkg ^ swz(row)) << 4));
            }
#pragma unroll
            for (int b = 0; b < 4; b++) {
                const int row = wc * 64 + b * 16 + frow;
                fb[b] = *reinterpret_cast<const i32x4 *>(tB + (kb * GM_TILE + row) * 64 + ((fkg ^ swz(row)) << 4));
            }
#pragma unroll
            for (int a = 0; a < 4; a++)
#pragma unroll
                for (int b = 0; b < 4; b++) {
                    const i32x4 s = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa[a], fb[b], i32x4{0, 0, 0, 0}, 0, 0, 0);
                    if (!tiny) {
#pragma unroll
                        for (int t = 0; t < 4; t++) acc[a][b][t] = __builtin_fmaf(c8, (float)s[t], acc[a][b][t]);
                    } else {
#pragma unroll
                        for (int t = 0; t < 4; t++) acc[a][b][t] = __builtin_fmaf(c, (float)(s[t] >> 8), acc[a][b][t]);
                    }
                }
        }
        if (st + 1 < nstages) stash(buf ^ 1);
        __syncthreads();
    }

    // C/D layout of the 16x16 MFMA: column = lane & 15, row = 4 * (lane >> 4) + t
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 4; b++)
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const uint64_t i = m0 + wr * 64 + a * 16 + 4 * (lane >> 4) + t;
                const uint64_t j = n0 + wc * 64 + b * 16 + (lane & 15);
                __builtin_nontemporal_store(acc[a][b][t], &C[i * N + j]);
            }
}

int clm4_gemm_mfma(const int8_t *A, const float *sA, uint64_t M, uint64_t K, const int8_t *B, const float *sB, uint64_t N, float *C,
                   hipStream_t st)
{
    const uint32_t tiles_m = (uint32_t)(M / GM_TILE), tiles_n = (uint32_t)(N / GM_TILE);
    const size_t lds = 2 * GM_STAGE_BYTES;
    static bool attr_set[64] = {false};
    int dev = 0;
    CLV_HIP(hipGetDevice(&dev));
    if (dev >= 0 && dev < 64 && !attr_set[dev]) {
        CLV_HIP(hipFuncSetAttribute((const void *)k_m4_gemm_mfma, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set[dev] = true;
    }
    hipLaunchKernelGGL(k_m4_gemm_mfma, dim3(tiles_m * tiles_n), dim3(256), lds, st, (const uint8_t *)A, sA, (const uint8_t *)B, sB, M, N, K, C,
                       tiles_m, tiles_n);
    CLV_LAUNCH_CHECK();
    return CLV_OK;
}
