// gemm6.hip -- CloverMatrix4 x CloverMatrix4^T -> fp32 through the block-scaled FP6 matrix instruction of gfx950.
//
// Same semantics as gemm4.hip:  C[i][j] = fold_b fmaf(c_b, (float)S_b, C)  with S_b the exact integer sum
// of the 64 nibble products of K-block b.  What changes is where (float)S_b comes from.  The int8 MFMA returns int32, and
// turning that into fp32 plus the fma costs two VALU instructions per element and K-block -- measured, that fold costs as
// much as all the MFMAs and does not overlap them (gemm4.hip: 0.31 ms MFMA alone, 0.63 ms with the fold, 8192^3).
// v_mfma_scale_f32_32x32x64_f8f6f4 contracts K = 64 -- exactly one Clover block -- returns fp32, runs at twice the int8
// rate with FP6 (E2M3) operands, and E2M3 holds every Clover value exactly: code = sign | 00 | magnitude is
// magnitude / 8 (sub-normal and first binade are one linear ramp).  Products are multiples of 1/64 below 1 and a sum of
// 64 of them is far inside fp32, so the matrix pipe returns S / 64 exactly; the instruction's block scales, 2^3 on either
// side, turn that into S itself, and ONE fma per element folds it in.
//
// Pass 1 (k_m4_to_fp6): nibbles -> FP6, 24 bytes per 32 elements (half a K-block = what one lane feeds the instruction).
//   Row-major, 48 bytes per row and K-block laid out [half 0: 16 B][half 1: 16 B][8 B][8 B], the two 8-byte tails
//   swapped in rows with bit 4 set (see the fragment reads).  Memory-bound, ~4 % of the GEMM time at 8192^3.  The element
//   order inside a half is whatever the conversion produces (the same for A and B): integer sums are order-free.
// Pass 2 (k_m4_gemm_fp6): 128x128 tile per 512-thread workgroup (2x4 waves, wave tile 64x32 = two 32x32 results inside
//   one scale tile of A and of B, so c_b is wave-uniform).  A stage is one pair of K-blocks, brought in by LDS-DMA
//   (global_load_lds_dwordx4: no staging registers, no ds_write) three stages deep.  The DMA writes lane-linear, so the
//   LDS image of an operand and K-block is the plain [row][48 B] array.  A fragment lane (row = lane & 31,
//   half = lane >> 5) reads 16 + 8 bytes: row stride 48 B = 3 x 16 with 3 odd, and the ds_read_b128 lane groups cover
//   every residue of row mod 16 once, so they are conflict-free; the ds_read_b64 half-waves see rows r and r + 16 on
//   the same banks, which the swapped tails move apart.  Both run at the full 256 B/clk (a ds_read2_b64 would not:
//   MI355X_MICROARCH.md LDS table).  One barrier per stage.
#include "common.h"

#include <stdlib.h>

typedef int i32x8 __attribute__((ext_vector_type(8)));

#define G6_TILE 128
#define G6_LDS_BYTES(WR) (3 * (2 * 64 * (WR) * 48 + 2 * 128 * 48))      // three stage buffers
#define G6_SCALE_8 0x82828282                       // E8M0 130 = 2^3 in every byte: (8 a)(8 b) turns magnitude / 8 back into integers

// ---- pass 1 -----------------------------------------------------------------------------------------------------
// 4 nibbles sitting in the low halves of 4 bytes -> 4 sign|00|magnitude codes -> 24 bits
__device__ __forceinline__ uint32_t fp6_codes24(uint32_t t)
{
    const uint32_t s = t & 0x08080808u;                       // sign bits of the two's-complement nibbles
    const uint32_t s1 = s >> 3;
    const uint32_t neg = (s << 1) - s1;                       // 0x0F in the negative bytes
    const uint32_t c = (((t ^ neg) + s1) | (s << 2));         // magnitude (16 - t or t), sign to bit 5
    return (c & 0x3Fu) | ((c >> 2) & 0xFC0u) | ((c >> 4) & 0x3F000u) | ((c >> 6) & 0xFC0000u);
}

__global__ __launch_bounds__(256) void k_m4_to_fp6(const u32x4 *__restrict__ q, uint8_t *__restrict__ w, uint64_t nhalves, uint64_t kbn)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nhalves; i += stride) {
        const u32x4 p = q[i];
        uint32_t x[8];
#pragma unroll
        for (int d = 0; d < 4; d++) {
            x[2 * d] = fp6_codes24((p[d] >> 4) & 0x0F0F0F0Fu);
            x[2 * d + 1] = fp6_codes24(p[d] & 0x0F0F0F0Fu);
        }
        const uint64_t blk = i >> 1;                      // row * kbn + K-block
        const uint32_t h = (uint32_t)i & 1u;
        const uint32_t swap = (uint32_t)((blk / kbn) >> 4) & 1u;
        uint8_t *o = w + blk * 48;
        *reinterpret_cast<u32x4 *>(o + 16 * h) = u32x4{x[0] | (x[1] << 24), (x[1] >> 8) | (x[2] << 16), (x[2] >> 16) | (x[3] << 8), x[4] | (x[5] << 24)};
        *reinterpret_cast<u32x2 *>(o + 32 + 8 * (h ^ swap)) = u32x2{(x[5] >> 8) | (x[6] << 16), (x[6] >> 16) | (x[7] << 8)};
    }
}

// ---- pass 2 -----------------------------------------------------------------------------------------------------
typedef __attribute__((address_space(1))) const void gptr_t;
typedef __attribute__((address_space(3))) void lptr_t;

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ i32x8 frag24(const char *p16, const char *p8)
{
    const u32x4 a = *reinterpret_cast<const u32x4 *>(p16);
    const u32x2 b = *reinterpret_cast<const u32x2 *>(p8);
    return i32x8{(int)a.x, (int)a.y, (int)a.z, (int)a.w, (int)b.x, (int)b.y, 0, 0};
}

template <int WR>       // wave rows: the tile is 64 WR x 128, 4 WR waves (WR = 2: two workgroups per CU, WR = 4: one)
__global__ __launch_bounds__(256 * WR, WR == 2 ? 4 : 1) void k_m4_gemm_fp6(const uint8_t *__restrict__ A6, const float *__restrict__ sA,
                                                        const uint8_t *__restrict__ B6, const float *__restrict__ sB, uint64_t M,
                                                        uint64_t N, uint64_t K, float *__restrict__ C, uint32_t tiles_m, uint32_t tiles_n)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];

    // tile assignment, XCD-aware as in gemm4.hip (block b runs on XCD b % 8)
    const uint32_t nwg = tiles_m * tiles_n;
    uint32_t id = blockIdx.x;
    {
        const uint32_t q = nwg / 8, r = nwg % 8, xcd = id % 8, s = id / 8;
        id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + s;
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
    const int wr = wave >> 2, wc = wave & 3;
    constexpr int SUBA = 64 * WR * 48, SUBB = 128 * 48;          // one operand, one K-block: [row][48 B]
    constexpr int BUF = 2 * SUBA + 2 * SUBB;                    // stage: [A j0][A j1][B j0][B j1]
    constexpr int NW = 4 * WR, NCH = BUF / 1024, NI = (NCH + NW - 1) / NW;
    const uint64_t m0 = (uint64_t)tm * (64 * WR), n0 = (uint64_t)tn * G6_TILE;
    const uint64_t kbn = K / 64;
    const uint64_t npairs = kbn / 2;
    const uint64_t rs = kbn * 48;                               // bytes per row of the FP6 images

    // DMA roles: the stage image is NCH chunks of 1 KiB (64 slots of 16 B); chunk q lies in sub-image A j (q < 6 WR,
    // j = q / (3 WR)) or B j and holds slots 64 c .. 64 c + 63 of its [row][3 x 16 B] array; wave w brings in chunks
    // w + NW i.  Source = uniform base + lane offset.  Rows past the end of A (WR = 4, M an odd multiple of 128) are read
    // from what follows in the workspace -- the image of B -- and never stored.
    const uint8_t *tileA = A6 + m0 * rs, *tileB = B6 + n0 * rs;
    uint32_t voff[NI];
#pragma unroll
    for (int i = 0; i < NI; i++) {
        const int q = wave + NW * i;
        const int c = q < 6 * WR ? q % (3 * WR) : (q - 6 * WR) % 6;
        const int s = 64 * c + lane;
        const int row = s / 3, piece = s - 3 * row;
        voff[i] = (uint32_t)row * (uint32_t)rs + 16u * piece;
    }
    const int nrole = NCH % NW == 0 || wave < NCH % NW ? NI : NI - 1;       // wave-uniform
    auto issue = [&](int buf, uint64_t p) {
#pragma unroll
        for (int i = 0; i < NI; i++) {
            const int q = wave + NW * i;
            if (i < nrole) {
                const bool isA = q < 6 * WR;
                const int j = isA ? q / (3 * WR) : (q - 6 * WR) / 6;
                const uint8_t *src = (isA ? tileA : tileB) + 96 * p + 48 * j;
                __builtin_amdgcn_global_load_lds((gptr_t *)(src + voff[i]), (lptr_t *)(smem + buf * BUF + 1024 * q), 16, 0, 0);
            }
        }
    };
    // all of this wave's DMAs but those of the newest stage have landed
    auto wait_prev = [&](bool newest_in_flight) {
        if (!newest_in_flight) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (nrole == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    };
    static_assert(NI == 3, "wait_prev covers 3 or 2 DMA instructions per wave and stage");

    // fragment lane: row = lane & 31 of the 32-row tile, half = lane >> 5 of the K-block
    const int frow = lane & 31, h = lane >> 5;
    const int tail = 32 + 8 * (h ^ ((lane >> 4) & 1));
    const int offA = (wr * 64 + frow) * 48 + 16 * h;
    const int offB = 2 * SUBA + (wc * 32 + frow) * 48 + 16 * h;
    // the 8-byte tails through unrelated registers: hipcc would otherwise pair them into ds_read2_b64 / ds_read2st64_b64,
    // which run at half the rate of two ds_read_b64 and have other bank rules
    int tA[2][2], tB[2];
#pragma unroll
    for (int j = 0; j < 2; j++) {
        tB[j] = 2 * SUBA + j * SUBB + (wc * 32 + frow) * 48 + tail;
        asm volatile("" : "+v"(tB[j]));
#pragma unroll
        for (int a = 0; a < 2; a++) {
            tA[j][a] = j * SUBA + (wr * 64 + a * 32 + frow) * 48 + tail;
            asm volatile("" : "+v"(tA[j][a]));
        }
    }

    f32x16 acc[2];
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int t = 0; t < 16; t++) acc[a][t] = 0.0f;

    const bool live = m0 + wr * 64 < M;                          // wave-uniform: M is a multiple of 128
    const float *sArow = sA + ((m0 >> 6) + (live ? wr : 0)) * kbn;
    const float *sBrow = sB + ((n0 >> 6) + (wc >> 1)) * kbn;
    const f32x16 zero16 = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};

    // Three buffers: stage p + 2 is requested while stage p is computed, so a DMA has two stage times to land.  Raw barrier:
    // __syncthreads() would add a vmcnt(0) and drain the prefetch (cdna_hip_programming.md, pipelining across barriers).
    issue(0, 0);
    if (npairs > 1) issue(1, 1);
    wait_prev(npairs > 1);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    int buf = 0;
    for (uint64_t p = 0; p < npairs; p++) {
        if (p + 2 < npairs) issue(buf >= 1 ? buf - 1 : 2, p + 2);            // (p + 2) % 3: last read in stage p - 1
        const char *base = smem + buf * BUF;
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const float c = (sArow[2 * p + j] * CLV_RCP49) * sBrow[2 * p + j];
            const i32x8 fb = frag24(base + offB + j * SUBB, base + tB[j]);
#pragma unroll
            for (int a = 0; a < 2; a++) {
                const i32x8 fa = frag24(base + offA + j * SUBA + a * 32 * 48, base + tA[j][a]);
                const f32x16 s = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fa, fb, zero16, 2, 2, 0, G6_SCALE_8, 0, G6_SCALE_8);
#pragma unroll
                for (int t = 0; t < 16; t++) acc[a][t] = __builtin_fmaf(c, s[t], acc[a][t]);
            }
        }
        // stage p + 1 must have landed before anyone reads it; stage p + 2 (if requested) may stay in flight
        // (the register operands only pin the wait behind the stage's arithmetic: hipcc otherwise hoists it to the top)
        asm volatile("" ::"v"(acc[0][15]), "v"(acc[1][15]) : "memory");
        wait_prev(p + 2 < npairs);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        buf = buf == 2 ? 0 : buf + 1;
    }

    if (!live) return;
    // C/D layout of the 32x32 tile: column = lane & 31, row = (t & 3) + 8 (t >> 2) + 4 (lane >> 5)
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int t = 0; t < 16; t++) {
            const uint64_t i = m0 + wr * 64 + a * 32 + (t & 3) + 8 * (t >> 2) + 4 * (lane >> 5);
            const uint64_t j = n0 + wc * 32 + (lane & 31);
            __builtin_nontemporal_store(acc[a][t], &C[i * N + j]);
        }
}

// workspace: the FP6 images of A and B, (M + N) * K * 3/4 bytes
int clm4_gemm_fp6(const int8_t *A, const float *sA, uint64_t M, uint64_t K, const int8_t *B, const float *sB, uint64_t N, float *C,
                  hipStream_t st)
{
    const uint64_t a_bytes = M * K / 4 * 3, b_bytes = N * K / 4 * 3;
    void *ws = nullptr;
    int rc = clv_internal_workspace(&ws, a_bytes + b_bytes);
    if (rc) return rc;
    uint8_t *A6 = reinterpret_cast<uint8_t *>(ws), *B6 = A6 + a_bytes;
    const int cus = clv_cu_count();
    auto widen = [&](const int8_t *q, uint8_t *w, uint64_t rows) {
        const uint64_t nh = rows * K / 32;
        uint64_t blocks = (nh + 255) / 256;
        if (blocks > (uint64_t)cus * 16) blocks = (uint64_t)cus * 16;
        hipLaunchKernelGGL(k_m4_to_fp6, dim3((unsigned)blocks), dim3(256), 0, st, (const u32x4 *)q, w, nh, K / 64);
    };
    widen(A, A6, M);
    widen(B, B6, N);
    CLV_LAUNCH_CHECK();
    // 256x128 tiles (one 16-wave workgroup per CU) stage 25 % fewer bytes per flop than two 128x128 workgroups; used once
    // there are enough of them to occupy the chip.  CLV_G6_ROWS=128|256 forces one (A/B runs).
    static const int force = [] { const char *e = getenv("CLV_G6_ROWS"); return e ? atoi(e) : 0; }();
    const uint32_t tiles_n = (uint32_t)(N / G6_TILE), big_m = (uint32_t)((M + 255) / 256);
    const bool big = force ? force == 256 : (uint64_t)big_m * tiles_n >= (uint64_t)cus;
    if (big) {
        CLV_HIP(hipFuncSetAttribute((const void *)k_m4_gemm_fp6<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)G6_LDS_BYTES(4)));
        hipLaunchKernelGGL(k_m4_gemm_fp6<4>, dim3(big_m * tiles_n), dim3(1024), G6_LDS_BYTES(4), st, A6, sA, B6, sB, M, N, K, C, big_m, tiles_n);
    } else {
        const uint32_t tiles_m = (uint32_t)(M / G6_TILE);
        CLV_HIP(hipFuncSetAttribute((const void *)k_m4_gemm_fp6<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)G6_LDS_BYTES(2)));
        hipLaunchKernelGGL(k_m4_gemm_fp6<2>, dim3(tiles_m * tiles_n), dim3(512), G6_LDS_BYTES(2), st, A6, sA, B6, sB, M, N, K, C, tiles_m, tiles_n);
    }
    CLV_LAUNCH_CHECK();
    return CLV_OK;
}
