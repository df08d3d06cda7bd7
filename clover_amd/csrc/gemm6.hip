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
//   48 bytes per row and K-block laid out [half 0: 16 B][half 1: 16 B][8 B][8 B], the two 8-byte tails swapped in rows
//   with bit 4 set (see the fragment reads); rows and K-blocks in the order pass 2 stages them (see k_m4_to_fp6).
//   Memory-bound, 6 % of the GEMM time at 8192^3.  The element order inside a half is whatever the conversion produces
//   (the same for A and B): integer sums are order-free.
// Pass 2 (k_m4_gemm_fp6): 128x128 tile per 256-thread workgroup (2x2 waves, wave tile 64x64 = four 32x32 results inside
//   one scale tile of A and of B, so c_b is wave-uniform), three workgroups per CU.  A stage is one pair of K-blocks,
//   brought in by LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write), double-buffered.  The DMA
//   writes lane-linear, so the LDS image of an operand and K-block is the plain [row][48 B] array.  A fragment lane
//   (row = lane & 31, half = lane >> 5) reads 16 + 8 bytes: row stride 48 B = 3 x 16 with 3 odd, and the ds_read_b128
//   lane groups cover every residue of row mod 16 once, so they are conflict-free; the ds_read_b64 half-waves see rows
//   r and r + 16 on the same banks, which the swapped tails move apart.  Both run at the full 256 B/clk (a
//   ds_read2_b64 would not: MI355X_MICROARCH.md LDS table).  One barrier per stage.
//   (An 8-wave workgroup with 64x32 wave tiles, two per CU, three buffers: 2-6 % slower from 4096^3 up -- half again the
//   fragment reads per MFMA -- and 10 % faster at 2048^3.)
#include "common.h"

#include <stdlib.h>
#include <string.h>

typedef int i32x8 __attribute__((ext_vector_type(8)));

#define G6_TILE 128
#define G6_LDS_BYTES (2 * 4 * G6_TILE * 48)         // two stage buffers of [A j0][A j1][B j0][B j1], 128 rows x 48 B each
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

// Output order = the order pass 2 stages it: [tile of 128 rows][pair of K-blocks][K-block][row][48 B], so that one stage of one
// operand is one contiguous run and every DMA instruction of pass 2 reads 1 KiB of consecutive bytes.
// A workgroup re-codes 64 rows x 8 K-blocks: loads walk along the rows (256 contiguous bytes per row), the codes go to LDS
// as [K-block][row][48 B], and each K-block's 64 x 48 = 3 KiB leave as one contiguous run.
#define F6_ROWS 64
#define F6_KB 8
// tile_rows: 128 (k_m4_gemm_fp6*) or 256 (k_m4_gemm_fp6_t256); rows_a / rows_b: the real row counts -- the grid covers the images
// padded to whole tiles, rows past the end are written as zeros (they are staged and multiplied, never stored).
__global__ __launch_bounds__(256) void k_m4_to_fp6(const u32x4 *__restrict__ qa, uint8_t *__restrict__ wa, uint32_t row_groups_a,
                                                   const u32x4 *__restrict__ qb, uint8_t *__restrict__ wb, uint64_t kbn, uint32_t tile_rows,
                                                   uint64_t rows_a, uint64_t rows_b)
{
    // both operands in one launch: blockIdx.y walks the 64-row groups of A, then those of B
    const bool second = blockIdx.y >= row_groups_a;
    const u32x4 *__restrict__ q = second ? qb : qa;
    uint8_t *__restrict__ w = second ? wb : wa;
    const uint32_t by = second ? blockIdx.y - row_groups_a : blockIdx.y;
    const uint64_t rows_real = second ? rows_b : rows_a;
    __shared__ __attribute__((aligned(16))) uint8_t img[F6_KB * F6_ROWS * 48];
    const uint64_t row0 = (uint64_t)by * F6_ROWS, kb0 = (uint64_t)blockIdx.x * F6_KB;
    const int tid = threadIdx.x;
#pragma unroll
    for (int it = 0; it < F6_ROWS * F6_KB * 2 / 256; it++) {
        const int e = tid + 256 * it;                         // (row, half) with the 16 halves of a row adjacent
        const int r = e >> 4, hh = e & 15, kb = hh >> 1, h = hh & 1;
        const uint64_t row = row0 + r;
        u32x4 p = {0u, 0u, 0u, 0u};
        if (kb0 + kb < kbn && row < rows_real) p = q[(row * kbn + kb0 + kb) * 2 + h];      // kbn is even, not a multiple of 8
        uint32_t x[8];
#pragma unroll
        for (int d = 0; d < 4; d++) {
            x[2 * d] = fp6_codes24((p[d] >> 4) & 0x0F0F0F0Fu);
            x[2 * d + 1] = fp6_codes24(p[d] & 0x0F0F0F0Fu);
        }
        const uint32_t swap = (uint32_t)(row >> 4) & 1u;
        uint8_t *o = img + (kb * F6_ROWS + r) * 48;
        *reinterpret_cast<u32x4 *>(o + 16 * h) = u32x4{x[0] | (x[1] << 24), (x[1] >> 8) | (x[2] << 16), (x[2] >> 16) | (x[3] << 8), x[4] | (x[5] << 24)};
        *reinterpret_cast<u32x2 *>(o + 32 + 8 * (h ^ swap)) = u32x2{(x[5] >> 8) | (x[6] << 16), (x[6] >> 16) | (x[7] << 8)};
    }
    __syncthreads();
    // K-block kb of rows row0 .. row0 + 63: sub-image (tile, pair, j), rows r0 .. r0 + 63 of it
    const uint64_t tile = row0 / tile_rows, r0 = row0 % tile_rows;
#pragma unroll
    for (int it = 0; it < F6_KB * F6_ROWS * 48 / 16 / 256; it++) {
        const int e = tid + 256 * it;                         // 16-byte piece of the LDS image
        const int kb = e / (F6_ROWS * 3), off = e - kb * (F6_ROWS * 3);
        const uint64_t sub = (tile * (kbn / 2) + (kb0 + kb) / 2) * 2 + ((kb0 + kb) & 1);
        if (kb0 + kb < kbn) *reinterpret_cast<u32x4 *>(w + (sub * tile_rows + r0) * 48 + 16 * (uint64_t)off) = *reinterpret_cast<const u32x4 *>(img + 16 * e);
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

__global__ __launch_bounds__(256, 3) void k_m4_gemm_fp6(const uint8_t *__restrict__ A6, const float *__restrict__ sA,
                                                            const uint8_t *__restrict__ B6, const float *__restrict__ sB, uint64_t /* M */,
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
    const int wr = wave >> 1, wc = wave & 1;
    constexpr int SUB = G6_TILE * 48;                           // one operand, one K-block: [row][48 B] = 6 KiB
    constexpr int BUF = 4 * SUB;                                // stage: [A j0][A j1][B j0][B j1] = 24 DMA chunks of 1 KiB
    const uint64_t m0 = (uint64_t)tm * G6_TILE, n0 = (uint64_t)tn * G6_TILE;
    const uint64_t kbn = K / 64;
    const uint64_t npairs = kbn / 2;

    // DMA roles: the stage image is 24 chunks of 1 KiB, 12 of A then 12 of B, and pass 1 laid both operands out in exactly
    // this order: a chunk is 1 KiB of consecutive bytes, lane l takes bytes 16 l.  (With row-major operands, 22 row
    // segments per instruction, the L1 address path was 80 % busy and the kernel 0.65 ms; like this 44 % and 0.55 ms.)
    // Wave w brings in chunks w + 4 i of A and of B, i < 3.
    const uint8_t *stageA = A6 + (uint64_t)tm * npairs * (2 * SUB) + 1024 * wave + 16 * lane;
    const uint8_t *stageB = B6 + (uint64_t)tn * npairs * (2 * SUB) + 1024 * wave + 16 * lane;
    auto issue = [&](int buf, uint64_t p) {
        char *l = smem + buf * BUF + 1024 * wave;
        const uint8_t *a = stageA + p * (2 * SUB), *b = stageB + p * (2 * SUB);
#pragma unroll
        for (int i = 0; i < 3; i++) {
            __builtin_amdgcn_global_load_lds((gptr_t *)(a + 4096 * i), (lptr_t *)(l + 4096 * i), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gptr_t *)(b + 4096 * i), (lptr_t *)(l + 2 * SUB + 4096 * i), 16, 0, 0);
        }
    };

    // fragment lane: row = lane & 31 of a 32-row tile, half = lane >> 5 of the K-block
    const int frow = lane & 31, h = lane >> 5;
    const int tail = 32 + 8 * (h ^ ((lane >> 4) & 1));
    const int offA = (wr * 64 + frow) * 48 + 16 * h;
    const int offB = 2 * SUB + (wc * 64 + frow) * 48 + 16 * h;
    // the 8-byte tails through unrelated registers: hipcc would otherwise pair them into ds_read2_b64 / ds_read2st64_b64,
    // which run at half the rate of two ds_read_b64 and have other bank rules
    int tA[2][2], tB[2][2];                                     // [K-block of the stage][32-row tile]
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
        for (int a = 0; a < 2; a++) {
            tA[j][a] = j * SUB + (wr * 64 + a * 32 + frow) * 48 + tail;
            tB[j][a] = (2 + j) * SUB + (wc * 64 + a * 32 + frow) * 48 + tail;
            asm volatile("" : "+v"(tA[j][a]), "+v"(tB[j][a]));
        }

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int t = 0; t < 16; t++) acc[a][b][t] = 0.0f;

    const float *sArow = sA + ((m0 >> 6) + wr) * kbn;
    const float *sBrow = sB + ((n0 >> 6) + wc) * kbn;
    const f32x16 zero16 = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};

    // Two buffers: stage p + 1 is requested while stage p is computed (three, two stages ahead with counted vmcnt, measured the
    // same and would cost the third workgroup per CU).  Raw barrier + explicit waits: the compiler does not know what an LDS-DMA
    // writes, and a __syncthreads() here adds nothing.
    issue(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    for (uint64_t p = 0; p < npairs; p++) {
        const int buf = (int)(p & 1);
        const char *base = smem + buf * BUF;
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const float c = (sArow[2 * p + j] * CLV_RCP49) * sBrow[2 * p + j];
            i32x8 fb[2];
#pragma unroll
            for (int b = 0; b < 2; b++) fb[b] = frag24(base + offB + j * SUB + b * 32 * 48, base + tB[j][b]);
#pragma unroll
            for (int a = 0; a < 2; a++) {
                const i32x8 fa = frag24(base + offA + j * SUB + a * 32 * 48, base + tA[j][a]);
#pragma unroll
                for (int b = 0; b < 2; b++) {
                    const f32x16 s = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fa, fb[b], zero16, 2, 2, 0, G6_SCALE_8, 0, G6_SCALE_8);
#pragma unroll
                    for (int t = 0; t < 16; t++) acc[a][b][t] = __builtin_fmaf(c, s[t], acc[a][b][t]);
                }
            }
            // The requests for stage p + 1 go out between the two K-blocks, pinned behind the first one's arithmetic.  At the top
            // of the stage every wave of the workgroup has just left the barrier and they would all block on the DMA queue
            // together: 0.565 -> 0.536 ms at 8192^3.  (In three parts after the stage's quarters: 0.76 ms, the pins get in the
            // way of hipcc's interleaving of MFMAs and folds.)
            if (j == 0 && p + 1 < npairs) {
                asm volatile("" ::"v"(acc[1][1][15]) : "memory");
                issue(buf ^ 1, p + 1);                                       // the other buffer was last read in stage p - 1
            }
        }
        // stage p + 1 must have landed before anyone reads it
        // (the register operands only pin the wait behind the stage's arithmetic: hipcc otherwise hoists it to the top)
        asm volatile("s_waitcnt vmcnt(0)" ::"v"(acc[0][0][15]), "v"(acc[1][1][15]) : "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }

    // C/D layout of the 32x32 tile: column = lane & 31, row = (t & 3) + 8 (t >> 2) + 4 (lane >> 5)
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int t = 0; t < 16; t++) {
                const uint64_t i = m0 + wr * 64 + a * 32 + (t & 3) + 8 * (t >> 2) + 4 * (lane >> 5);
                const uint64_t j = n0 + wc * 64 + b * 32 + (lane & 31);
                __builtin_nontemporal_store(acc[a][b][t], &C[i * N + j]);
            }
}

// ---- pass 2, hand-scheduled ---------------------------------------------------------------------------------------
// Same tile, same LDS image, same DMA and the same arithmetic as k_m4_gemm_fp6 above; what differs is WHO orders the instructions.
// hipcc issues a K-block's MFMAs first and its folds after, hoists fragment reads across K-blocks until it spills, and pairs the
// fold into v_pk_fma_f32, which serialises with the matrix pipe (profiles/r02_mfma_fold_probe*.txt).  Here the whole main loop and
// the store of C are ONE asm statement generated by tools/gen_gemm6_loop.py (schedule and register map are documented there):
// the fold of a result runs two MFMAs behind it, half scalar (beside the MFMA just issued) and half packed; fragments are
// single-buffered with re-loads placed right behind their last reader; one barrier per stage, placed so that the next stage's
// fragments are requested before the current stage's last two MFMAs.
#include "gemm6_loop.inc"

// stage0 / nstages: the range of stages (pairs of K-blocks) to contract -- all of them for the GEMM, a sub-range for clm4_gemm_i32
template <int V>
__global__ __launch_bounds__(256, 3) void k_m4_gemm_fp6_asm(const uint8_t *__restrict__ A6, const float *__restrict__ sA,
                                                                const uint8_t *__restrict__ B6, const float *__restrict__ sB, uint32_t stage0,
                                                                uint64_t N, uint64_t K, float *__restrict__ C, uint32_t tiles_m, uint32_t tiles_n,
                                                                uint32_t nstages)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
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
    const int wr = wave >> 1, wc = wave & 1;
    constexpr int SUB = G6_TILE * 48;
    const uint64_t m0 = (uint64_t)tm * G6_TILE, n0 = (uint64_t)tn * G6_TILE;
    const uint64_t kbn = K / 64;
    const uint64_t image_stages = kbn / 2;                       // stages per tile in the FP6 images
    const uint32_t npairs = nstages;

    // wave-uniform addresses (SGPRs inside the loop)
    const uint64_t ga = (uint64_t)(A6 + ((uint64_t)tm * image_stages + stage0) * (2 * SUB) + 1024 * wave);
    const uint64_t gb = (uint64_t)(B6 + ((uint64_t)tn * image_stages + stage0) * (2 * SUB) + 1024 * wave);
    const uint64_t sa = (uint64_t)(sA + ((m0 >> 6) + wr) * kbn + 2 * (uint64_t)stage0);
    const uint64_t sb = (uint64_t)(sB + ((n0 >> 6) + wc) * kbn + 2 * (uint64_t)stage0);
    const uint64_t cb = (uint64_t)(C + (m0 + wr * 64) * N + n0 + wc * 64);
    const uint32_t lds0 = (uint32_t)(uintptr_t)smem;                       // LDS byte address of the stage buffers
    const uint32_t dma = lds0 + 1024 * wave;
    const uint32_t cstride = (uint32_t)(N * sizeof(float));
    // per-lane: fragment addresses in buffer 0 (row = lane & 31 of a 32-row tile, half = lane >> 5 of the K-block; see k_m4_gemm_fp6)
    const int frow = lane & 31, h = lane >> 5;
    const int tail = 32 + 8 * (h ^ ((lane >> 4) & 1));
    uint32_t a16 = lds0 + (wr * 64 + frow) * 48 + 16 * h;
    uint32_t a8 = lds0 + (wr * 64 + frow) * 48 + tail;
    uint32_t b16 = lds0 + 2 * SUB + (wc * 64 + frow) * 48 + 16 * h;
    uint32_t b8 = lds0 + 2 * SUB + (wc * 64 + frow) * 48 + tail;
    const uint32_t voff = 16 * lane;
    const uint32_t coff = (uint32_t)((4 * (uint64_t)(lane >> 5) * N + (lane & 31)) * sizeof(float));

#define G6_RUN(STR)                                                                                                                     \
    asm volatile(STR : [a16] "+v"(a16), [a8] "+v"(a8), [b16] "+v"(b16), [b8] "+v"(b8)                                                   \
                 : [voff] "v"(voff), [coff] "v"(coff), [ga] "s"(ga), [gb] "s"(gb), [sa] "s"(sa), [sb] "s"(sb), [cb] "s"(cb), [np] "s"(npairs), \
                   [dma] "s"(dma), [cstride] "s"(cstride)                                                                                  \
                 : G6_LOOP_CLOBBERS)
    if constexpr (V == 0) G6_RUN(G6_LOOP_ASM);
    else if constexpr (V == 100) G6_RUN(G6_LOOP_ASM_I32);        // no scales: the MFMAs accumulate over the K-blocks, C leaves as int32
#ifdef G6_LOOP_EXPERIMENTS      // timing-only variants with parts of the loop left out (tools/gen_gemm6_loop.py ... experiments)
    else if constexpr (V == 1) G6_RUN(G6_LOOP_ASM_NODMA);
    else if constexpr (V == 2) G6_RUN(G6_LOOP_ASM_NOLDS);
    else if constexpr (V == 3) G6_RUN(G6_LOOP_ASM_NODMA_NOLDS);
    else if constexpr (V == 4) G6_RUN(G6_LOOP_ASM_NOBARRIER);
    else if constexpr (V == 5) G6_RUN(G6_LOOP_ASM_NOFOLD);
    else if constexpr (V == 6) G6_RUN(G6_LOOP_ASM_ARITH);
    else if constexpr (V == 7) G6_RUN(G6_LOOP_ASM_DMAFIXED);
    else if constexpr (V == 8) G6_RUN(G6_LOOP_ASM_DMABONLY);
#endif
#undef G6_RUN
}

// ---- pass 2, 256 x 256 tile, persistent ----------------------------------------------------------------------------
// 16 waves (4 x 4 wave tiles of 64 x 64), ONE workgroup per CU, 144 KiB of LDS (three 48 KiB stage buffers), 128 VGPRs per wave.
// Half the LDS-DMA requests and L2 bytes per MFMA of the 128 x 128 tile (see tools/gen_gemm6_loop256.py for the schedule).
// A workgroup walks tiles id = blockIdx.x, + gridDim.x, ...; the asm statement is one tile: prologue, main loop, asynchronous
// store of C.  Operand images are those of k_m4_to_fp6 with tile_rows = 256, padded with zero rows to whole tiles.
#ifdef CLV_GEMM_EXPERIMENTS          // the bench-only probe library (clover_amd/build.py): the product's loops + timing-only variants
#include "gemm6_loop256_exp.inc"
// marks the library as the probe build: clv_version() then reads "clover_hip_probe ..." (runtime.hip), which load_library() refuses by default
extern "C" const char *clvx_probe_tag(void) { return "clover_hip_probe 0.1 (gfx950; GEMM loop timing variants -- results WRONG by construction -- and the mvm / read-bandwidth experiment kernels)"; }
#else
#include "gemm6_loop256.inc"
#endif
#define G6T_TILE 256
#define G6T_LDS_BYTES (3 * 4 * G6T_TILE * 48)       // three stage buffers of 48 KiB

template <bool I32, int V = 0>
__global__ __launch_bounds__(1024) void k_m4_gemm_fp6_t256(const uint8_t *__restrict__ A6, const float *__restrict__ sA,
                                                             const uint8_t *__restrict__ B6, const float *__restrict__ sB, uint64_t M, uint64_t N,
                                                             uint64_t K, float *__restrict__ C, uint32_t tiles_m, uint32_t tiles_n, uint32_t stage0,
                                                             uint32_t nstages)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // 0..15
    const int wr = wave >> 2, wc = wave & 3;
    constexpr int SUB = G6T_TILE * 48;
    const uint64_t kbn = K / 64, image_stages = kbn / 2;
    const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
    const uint32_t cstride = (uint32_t)(N * sizeof(float));
    // the wave's three of the 48 KiB-pieces of a stage image [A j0 | A j1 | B j0 | B j1]: pieces wave, 16 + wave, 32 + wave
    const uint32_t l0 = 1024 * wave;
    const uint32_t l1 = wave < 8 ? 1024 * (16 + wave) : 2 * SUB + 1024 * (wave - 8);
    const uint32_t l2 = 2 * SUB + 1024 * (8 + wave);
    const int frow = lane & 31, h = lane >> 5;
    const int tail = 32 + 8 * (h ^ ((lane >> 4) & 1));
    uint32_t a16 = lds0 + (wr * 64 + frow) * 48 + 16 * h;
    uint32_t a8 = lds0 + (wr * 64 + frow) * 48 + tail;
    uint32_t b16 = lds0 + 2 * SUB + (wc * 64 + frow) * 48 + 16 * h;
    uint32_t b8 = lds0 + 2 * SUB + (wc * 64 + frow) * 48 + tail;
    const uint32_t voff = 16 * lane;
    // Tile order.  The grid is always 256 workgroups; they are dealt to the 8 XCDs round-robin, so blockIdx % 8 names the L2 a
    // workgroup sits behind and blockIdx / 8 its slot (0..31) there.  The 32 workgroups of an XCD take a 4 x 8 block of tiles
    // (tiles outside the matrix are skipped): 12 operand panels go through that L2 per block instead of 33 when the workgroups of
    // one XCD hold a whole tile row (measured at 8192^3 with such an order: 50 % L2 hits, 1.7 GB of misses for 50 MB of operands).
    // No division inside the loop and few live scalars: the asm statement owns all 128 VGPRs, so neither a hoisted reciprocal nor
    // an SGPR spill has a register to go to.
    const uint32_t blocks_m = (tiles_m + 3) >> 2, blocks_n = (tiles_n + 7) >> 3;
    const uint32_t slot = blockIdx.x >> 3, sm = slot & 3, sn = slot >> 2;
    uint32_t bi = __builtin_amdgcn_readfirstlane((blockIdx.x & 7) / blocks_n);
    uint32_t bj = __builtin_amdgcn_readfirstlane((blockIdx.x & 7) % blocks_n);
    for (; bi < blocks_m;) {
        const uint32_t tm = 4 * bi + sm, tn = 8 * bj + sn;
        bj += 8;
        while (bj >= blocks_n) {
            bj -= blocks_n;
            ++bi;
        }
        // all scalar and 32-bit: 64-bit compares exist only on the VALU, and their operands would sit in VGPRs across the asm
        if (tm >= tiles_m || tn >= tiles_n) continue;
        const uint8_t *ia = A6 + ((uint64_t)tm * image_stages + stage0) * (2 * SUB);
        const uint8_t *ib = B6 + ((uint64_t)tn * image_stages + stage0) * (2 * SUB);
        const uint64_t g0 = (uint64_t)(ia + 1024 * wave);
        const uint64_t g1 = (uint64_t)((wave < 8 ? ia + 16 * 1024 : ib - 8 * 1024) + 1024 * wave);
        const uint64_t g2 = (uint64_t)(ib + 1024 * (8 + wave));
        // scale rows of this wave's 64 rows / columns (clamped for waves outside the matrix: they compute, but never store)
        const uint32_t rows64_m = (uint32_t)(M >> 6), rows64_n = (uint32_t)(N >> 6);
        const uint32_t wa = 4 * tm + wr, wb = 4 * tn + wc;
        const uint32_t ra = wa < rows64_m ? wa : rows64_m - 1, rb = wb < rows64_n ? wb : rows64_n - 1;
        const uint64_t sa = (uint64_t)(sA + (uint64_t)ra * kbn + 2 * (uint64_t)stage0);
        const uint64_t sb = (uint64_t)(sB + (uint64_t)rb * kbn + 2 * (uint64_t)stage0);
        const uint64_t cb = (uint64_t)(C + (uint64_t)wa * 64 * N + (uint64_t)wb * 64);
        const uint32_t flag = __builtin_amdgcn_readfirstlane((wa < rows64_m && wb < rows64_n) ? 1u : 0u);
#define G6T_RUN(STR)                                                                                                                         \
    asm volatile(STR : [a16] "+v"(a16), [a8] "+v"(a8), [b16] "+v"(b16), [b8] "+v"(b8)                                                       \
                 : [voff] "v"(voff), [g0] "s"(g0), [g1] "s"(g1), [g2] "s"(g2), [sa] "s"(sa), [sb] "s"(sb), [cb] "s"(cb), [np] "s"(nstages), \
                   [lds] "s"(lds0), [cstride] "s"(cstride), [l0] "s"(l0), [l1] "s"(l1), [l2] "s"(l2), [flag] "s"(flag)                    \
                 : G6T_LOOP_CLOBBERS)
        if constexpr (V == 0) {
            if constexpr (I32) G6T_RUN(G6T_LOOP_ASM_I32);
            else G6T_RUN(G6T_LOOP_ASM);
        }
#ifdef G6T_LOOP_EXPERIMENTS
#define G6T_VARIANT(N)                                  \
    else if constexpr (V == N) {                        \
        if constexpr (I32) G6T_RUN(G6T_LOOP_ASM_I32_V##N); \
        else G6T_RUN(G6T_LOOP_ASM_V##N);                \
    }
        G6T_VARIANT(1) G6T_VARIANT(2) G6T_VARIANT(3) G6T_VARIANT(4) G6T_VARIANT(5) G6T_VARIANT(6) G6T_VARIANT(7) G6T_VARIANT(8) G6T_VARIANT(9) G6T_VARIANT(10) G6T_VARIANT(11) G6T_VARIANT(12) G6T_VARIANT(13)
#undef G6T_VARIANT
#endif
#undef G6T_RUN
    }
}

// ---- host side ----------------------------------------------------------------------------------------------------
// An operand's FP6 image (rows * K * 3/4 bytes, staging order).  clm4_gemm re-codes both operands per call into the stream's
// scratch; clm4_gemm_prepare does it once into a buffer of its own for an operand that is multiplied many times.
struct clm4_gemm_operand {
    uint8_t *image;
    uint64_t rows, K;
    uint32_t tile;          // tile rows of the image layout (128 / 256)
    int device;
};

static inline uint64_t pad_rows(uint64_t rows, uint32_t tile) { return (rows + tile - 1) / tile * tile; }
static inline uint64_t image_bytes(uint64_t rows, uint64_t K, uint32_t tile) { return pad_rows(rows, tile) * K / 4 * 3; }

// which workgroup tile: 256 x 256 (persistent, one workgroup per CU) once there are enough tiles to occupy the chip, else 128 x 128.
// CLV_GEMM_TILE=128|256 forces one (A/B runs, tests).
static uint32_t pick_tile(uint64_t M, uint64_t N)
{
    static const int forced = [] { const char *e = getenv("CLV_GEMM_TILE"); return e ? atoi(e) : 0; }();
    if (forced == 128 || forced == 256) return (uint32_t)forced;
    const uint64_t tiles = ((M + 255) / 256) * ((N + 255) / 256);
    return tiles >= (uint64_t)clv_cu_count() ? 256u : 128u;
}

static int recode(const int8_t *A, uint8_t *A6, uint64_t M, const int8_t *B, uint8_t *B6, uint64_t N, uint64_t K, uint32_t tile, hipStream_t st)
{
    // one launch for whichever operands still need it (a NULL source = already prepared); the grid covers the padded images
    const uint64_t ra = A ? pad_rows(M, tile) : 0, rb = B ? pad_rows(N, tile) : 0;
    if (!ra && !rb) return CLV_OK;
    hipLaunchKernelGGL(k_m4_to_fp6, dim3((unsigned)((K / 64 + F6_KB - 1) / F6_KB), (unsigned)((ra + rb) / F6_ROWS)), dim3(256), 0, st, (const u32x4 *)A, A6,
                       (uint32_t)(ra / F6_ROWS), (const u32x4 *)B, B6, K / 64, tile, M, N);
    CLV_LAUNCH_CHECK();
    return CLV_OK;
}

extern "C" int clm4_gemm_prepare(const int8_t *q, uint64_t rows, uint64_t K, clm4_gemm_operand **op, void *stream)
{
    CLV_REQUIRE(q && op, "clm4_gemm_prepare: null pointer");
    CLV_REQUIRE(rows && K && rows % 128 == 0 && K % 128 == 0, "clm4_gemm_prepare: rows=%llu K=%llu must be non-zero multiples of 128",
                (unsigned long long)rows, (unsigned long long)K);
    // the layout depends on the tile the product will use, which depends on the OTHER operand's size too: prepared operands use
    // the 256-row layout when they are large enough on their own (rows >= 2048), and clm4_gemm_prepared follows them
    const uint32_t tile = rows >= 2048 && pick_tile(rows, rows) == 256 ? 256u : 128u;
    clm4_gemm_operand *o = new clm4_gemm_operand{nullptr, rows, K, tile, 0};
    if (hipGetDevice(&o->device) != hipSuccess || hipMalloc((void **)&o->image, image_bytes(rows, K, tile)) != hipSuccess) {
        clv_set_error("clm4_gemm_prepare: %s", hipGetErrorString(hipGetLastError()));
        delete o;
        return CLV_ERR_HIP;
    }
    int rc = recode(q, o->image, rows, nullptr, nullptr, 0, K, tile, as_stream(stream));
    if (rc) { (void)hipFree(o->image); delete o; return rc; }
    *op = o;
    return CLV_OK;
}

extern "C" int clm4_gemm_release(clm4_gemm_operand *op)
{
    if (!op) return CLV_OK;
    if (op->image) CLV_HIP(hipFree(op->image));
    delete op;
    return CLV_OK;
}

// A / B: the nibbles (needed when the matching operand is NULL); opA / opB: prepared images or NULL
static int gemm_fp6_run(const clm4_gemm_operand *opA, const int8_t *A, const float *sA, uint64_t M, uint64_t K, const clm4_gemm_operand *opB,
                        const int8_t *B, const float *sB, uint64_t N, void *C, bool i32, uint64_t kb_begin, uint64_t kb_count, hipStream_t st)
{
    if (opA && opB && opA->tile != opB->tile) {
        // clm4_gemm_prepare picks an image's staging layout (128- or 256-row tiles) from that operand's own row count, so a large and a
        // small operand can disagree.  The smaller one is then re-coded for this call in the other's layout (it is the cheaper of the
        // two) -- which needs its nibbles: callers that may mix sizes pass A / B besides the images (the C++ containers always do).
        const bool drop_a = (M <= N && A) || !B;
        CLV_REQUIRE(A || B, "clm4_gemm_prepared: the operands were prepared for different tile shapes (%u, %u rows) and neither operand's "
                            "nibbles were passed to re-code one of them", opA->tile, opB->tile);
        if (drop_a) opA = nullptr; else opB = nullptr;
    }
    uint32_t tile = opA ? opA->tile : opB ? opB->tile : pick_tile(M, N);
    const uint64_t a_bytes = opA ? 0 : image_bytes(M, K, tile), b_bytes = opB ? 0 : image_bytes(N, K, tile);
    uint8_t *A6 = opA ? opA->image : nullptr, *B6 = opB ? opB->image : nullptr;
    if (a_bytes + b_bytes) {
        void *ws = nullptr;
        int rc = clv_internal_workspace(&ws, a_bytes + b_bytes, st);
        if (rc) return rc;
        if (!opA) A6 = reinterpret_cast<uint8_t *>(ws);
        if (!opB) B6 = reinterpret_cast<uint8_t *>(ws) + a_bytes;
        rc = recode(opA ? nullptr : A, A6, M, opB ? nullptr : B, B6, N, K, tile, st);
        if (rc) return rc;
    }
    if (tile == 256) {
        const uint32_t tm = (uint32_t)((M + 255) / 256), tn = (uint32_t)((N + 255) / 256);
        const uint32_t grid256 = 256;       // 8 XCDs x 32 slots: the kernel's tile order is built on it
        const uint32_t s0 = i32 ? (uint32_t)(kb_begin / 2) : 0u, ns = i32 ? (uint32_t)(kb_count / 2) : (uint32_t)(K / 128);
        int variant = 0;
#ifdef G6T_LOOP_EXPERIMENTS
        static const int env_variant = [] { const char *e = getenv("CLV_GEMM_LOOP"); return e && e[0] == 'v' ? atoi(e + 1) : 0; }();
        variant = env_variant;
#endif
#define G6T_LAUNCH(I, V)                                                                                                                   \
    do {                                                                                                                                   \
        static bool attr_set[64] = {};         /* once per device: the attribute call costs host time on every launch otherwise */         \
        int dev = 0;                                                                                                                       \
        CLV_HIP(hipGetDevice(&dev));                                                                                                       \
        if (dev < 0 || dev >= 64 || !attr_set[dev]) {                                                                                      \
            CLV_HIP(hipFuncSetAttribute((const void *)k_m4_gemm_fp6_t256<I, V>, hipFuncAttributeMaxDynamicSharedMemorySize, G6T_LDS_BYTES)); \
            if (dev >= 0 && dev < 64) attr_set[dev] = true;                                                                                \
        }                                                                                                                                  \
        hipLaunchKernelGGL((k_m4_gemm_fp6_t256<I, V>), dim3(grid256), dim3(1024), G6T_LDS_BYTES, st, A6, sA, B6, sB, M, N, K, (float *)C, tm, \
                           tn, s0, ns);                                                                                                    \
    } while (0)
#define G6T_LAUNCH_V(V) do { if (i32) G6T_LAUNCH(true, V); else G6T_LAUNCH(false, V); } while (0)
        switch (variant) {
#ifdef G6T_LOOP_EXPERIMENTS
        case 1: G6T_LAUNCH_V(1); break;
        case 2: G6T_LAUNCH_V(2); break;
        case 3: G6T_LAUNCH_V(3); break;
        case 4: G6T_LAUNCH_V(4); break;
        case 5: G6T_LAUNCH_V(5); break;
        case 6: G6T_LAUNCH_V(6); break;
        case 7: G6T_LAUNCH_V(7); break;
        case 8: G6T_LAUNCH_V(8); break;
        case 9: G6T_LAUNCH_V(9); break;
        case 10: G6T_LAUNCH_V(10); break;
        case 11: G6T_LAUNCH_V(11); break;
        case 12: G6T_LAUNCH_V(12); break;
        case 13: G6T_LAUNCH_V(13); break;
#endif
        default: G6T_LAUNCH_V(0); break;
        }
#undef G6T_LAUNCH_V
#undef G6T_LAUNCH
        CLV_LAUNCH_CHECK();
        return CLV_OK;
    }
    const uint32_t tiles_m = (uint32_t)(M / G6_TILE), tiles_n = (uint32_t)(N / G6_TILE);
    const dim3 grid(tiles_m * tiles_n), block(256);
    if (i32) {
        // K-blocks [kb_begin, kb_begin + kb_count), both even: whole stages of images that hold K / 128 stages per tile
        hipLaunchKernelGGL(k_m4_gemm_fp6_asm<100>, grid, block, G6_LDS_BYTES, st, A6, (const float *)nullptr, B6, (const float *)nullptr, (uint32_t)(kb_begin / 2),
                           N, K, (float *)C, tiles_m, tiles_n, (uint32_t)(kb_count / 2));
        CLV_LAUNCH_CHECK();
        return CLV_OK;
    }
    // CLV_GEMM_LOOP=hipcc: the compiler-scheduled main loop (A/B runs); default: the hand-scheduled one
    static const bool hipcc_loop = [] { const char *e = getenv("CLV_GEMM_LOOP"); return e && !strcmp(e, "hipcc"); }();
    if (hipcc_loop) {
        hipLaunchKernelGGL(k_m4_gemm_fp6, grid, block, G6_LDS_BYTES, st, A6, sA, B6, sB, M, N, K, (float *)C, tiles_m, tiles_n);
    } else {
#define G6_LAUNCH(V) hipLaunchKernelGGL(k_m4_gemm_fp6_asm<V>, grid, block, G6_LDS_BYTES, st, A6, sA, B6, sB, 0u, N, K, (float *)C, tiles_m, tiles_n, (uint32_t)(K / 128))
#ifdef G6_LOOP_EXPERIMENTS
        static const int variant = [] { const char *e = getenv("CLV_GEMM_LOOP"); return e && e[0] == 'v' ? atoi(e + 1) : 0; }();
        switch (variant) {
        case 1: G6_LAUNCH(1); break;
        case 2: G6_LAUNCH(2); break;
        case 3: G6_LAUNCH(3); break;
        case 4: G6_LAUNCH(4); break;
        case 5: G6_LAUNCH(5); break;
        case 6: G6_LAUNCH(6); break;
        case 7: G6_LAUNCH(7); break;
        case 8: G6_LAUNCH(8); break;
        default: G6_LAUNCH(0); break;
        }
#else
        G6_LAUNCH(0);
#endif
#undef G6_LAUNCH
    }
    CLV_LAUNCH_CHECK();
    return CLV_OK;
}

int clm4_gemm_fp6(const int8_t *A, const float *sA, uint64_t M, uint64_t K, const int8_t *B, const float *sB, uint64_t N, float *C,
                  hipStream_t st)
{
    return gemm_fp6_run(nullptr, A, sA, M, K, nullptr, B, sB, N, C, false, 0, 0, st);
}

extern "C" int clm4_gemm_prepared(const clm4_gemm_operand *opA, const int8_t *A, const float *sA, uint64_t M, uint64_t K,
                                  const clm4_gemm_operand *opB, const int8_t *B, const float *sB, uint64_t N, float *C, void *stream)
{
    CLV_REQUIRE((opA || A) && (opB || B) && sA && sB && C, "clm4_gemm_prepared: null pointer");
    CLV_REQUIRE(M && N && K && M % 128 == 0 && N % 128 == 0 && K % 128 == 0, "clm4_gemm_prepared: M=%llu N=%llu K=%llu must be non-zero multiples of 128",
                (unsigned long long)M, (unsigned long long)N, (unsigned long long)K);
    CLV_REQUIRE(!opA || (opA->rows == M && opA->K == K), "clm4_gemm_prepared: operand A was prepared as %llu x %llu", opA ? (unsigned long long)opA->rows : 0ull,
                opA ? (unsigned long long)opA->K : 0ull);
    CLV_REQUIRE(!opB || (opB->rows == N && opB->K == K), "clm4_gemm_prepared: operand B was prepared as %llu x %llu", opB ? (unsigned long long)opB->rows : 0ull,
                opB ? (unsigned long long)opB->K : 0ull);
    return gemm_fp6_run(opA, A, sA, M, K, opB, B, sB, N, C, false, 0, 0, as_stream(stream));
}

// exact integer sums of one K-block range, one thread per element (8 x v_dot8 per K-block): any range, used for odd ranges
__global__ __launch_bounds__(256) void k_m4_gemm_i32_simple(const uint8_t *__restrict__ A, uint64_t M, uint64_t K, const uint8_t *__restrict__ B, uint64_t N,
                                                            uint64_t kb0, uint64_t kbc, int32_t *__restrict__ S)
{
    const uint64_t j = (uint64_t)blockIdx.x * 16 + (threadIdx.x & 15);
    const uint64_t i = (uint64_t)blockIdx.y * 16 + (threadIdx.x >> 4);
    if (i >= M || j >= N) return;
    const u32x4 *a = reinterpret_cast<const u32x4 *>(A + i * (K / 2));
    const u32x4 *b = reinterpret_cast<const u32x4 *>(B + j * (K / 2));
    int acc = 0;
    for (uint64_t blk = kb0; blk < kb0 + kbc; blk++) acc += dot32(a[2 * blk], b[2 * blk]) + dot32(a[2 * blk + 1], b[2 * blk + 1]);
    S[i * N + j] = acc;
}

static int gemm_i32_checked(const char *who, const clm4_gemm_operand *opA, const int8_t *A, uint64_t M, uint64_t K, const clm4_gemm_operand *opB,
                            const int8_t *B, uint64_t N, uint64_t kb_begin, uint64_t kb_count, int32_t *S, void *stream)
{
    CLV_REQUIRE((opA || A) && (opB || B) && S, "%s: null pointer", who);
    CLV_REQUIRE(M && N && K && M % 128 == 0 && N % 128 == 0 && K % 128 == 0, "%s: M=%llu N=%llu K=%llu must be non-zero multiples of 128", who,
                (unsigned long long)M, (unsigned long long)N, (unsigned long long)K);
    CLV_REQUIRE(kb_count && kb_begin + kb_count <= K / 64, "%s: K-blocks [%llu, +%llu) of %llu", who, (unsigned long long)kb_begin,
                (unsigned long long)kb_count, (unsigned long long)(K / 64));
    // fp32 accumulation of integers is exact below 2^24: 49 * 64 per K-block
    CLV_REQUIRE(kb_count <= (1ull << 24) / (49 * 64), "%s: %llu K-blocks would leave the exact range of the matrix pipe", who, (unsigned long long)kb_count);
    CLV_REQUIRE(!opA || (opA->rows == M && opA->K == K), "%s: operand A was prepared as %llu x %llu", who, opA ? (unsigned long long)opA->rows : 0ull,
                opA ? (unsigned long long)opA->K : 0ull);
    CLV_REQUIRE(!opB || (opB->rows == N && opB->K == K), "%s: operand B was prepared as %llu x %llu", who, opB ? (unsigned long long)opB->rows : 0ull,
                opB ? (unsigned long long)opB->K : 0ull);
    hipStream_t st = as_stream(stream);
    static const bool simple = [] { const char *e = getenv("CLV_GEMM_KERNEL"); return e && !strcmp(e, "simple"); }();
    if (((kb_begin | kb_count) & 1) || simple) {
        // odd ranges run on the nibbles themselves
        CLV_REQUIRE(A && B, "%s: an odd K-block range needs the nibbles of both operands (A, B), not only their prepared images", who);
        hipLaunchKernelGGL(k_m4_gemm_i32_simple, dim3((unsigned)(N / 16), (unsigned)(M / 16)), dim3(256), 0, st, (const uint8_t *)A, M, K, (const uint8_t *)B, N,
                           kb_begin, kb_count, S);
        CLV_LAUNCH_CHECK();
        return CLV_OK;
    }
    return gemm_fp6_run(opA, A, nullptr, M, K, opB, B, nullptr, N, S, true, kb_begin, kb_count, st);
}

extern "C" int clm4_gemm_i32(const int8_t *A, uint64_t M, uint64_t K, const int8_t *B, uint64_t N, uint64_t kb_begin, uint64_t kb_count, int32_t *S,
                             void *stream)
{
    return gemm_i32_checked("clm4_gemm_i32", nullptr, A, M, K, nullptr, B, N, kb_begin, kb_count, S, stream);
}

extern "C" int clm4_gemm_i32_prepared(const clm4_gemm_operand *opA, const int8_t *A, uint64_t M, uint64_t K, const clm4_gemm_operand *opB, const int8_t *B,
                                      uint64_t N, uint64_t kb_begin, uint64_t kb_count, int32_t *S, void *stream)
{
    return gemm_i32_checked("clm4_gemm_i32_prepared", opA, A, M, K, opB, B, N, kb_begin, kb_count, S, stream);
}
