// transpose4.hip -- CloverMatrix4::transpose.
// One of the callers either side of the hot path.  With mvm these are the five steps of the reference's quantized IHT / GD
// iterations (test/performance/01_measure.h:923-946, 999-1021), so x, t1..t3 can stay in HBM across iterations.
#include "common.h"

// =================================================================================================
// f2  CloverMatrix4::transpose (CloverMatrix4.h:1549-1663): out(j,i) = in(i,j) nibble-wise, tile scales
//     transposed (the reference calls IPP for those, :1657-1658).
//     workgroup = 256 x 256 elements staged through LDS so that BOTH the reads and the writes are 128-byte
//     runs (a row of the tile is 128 B on either side); a thread transposes 8x8 nibble blocks in registers.
//     LDS rows are padded to 33 words: the 8x8-block reads (lanes = 8 words x 4 row groups) are conflict-free,
//     the writes 2-way (free for ds_write_b32).  Algorithmic bytes: 2 * (1/2 + 4/4096) per element.
// =================================================================================================
#define TR_T 256                      // tile edge in elements
#define TR_W (TR_T / 8)               // 32 words per tile row
#define TR_S (TR_W + 1)               // padded LDS row stride in words
#ifndef TR_BH
#define TR_BH 8                      // tiles per XCD block: BH x BW (4x8: 0.214 ms, 8x8: 0.20-0.21, 16x8: 0.21, 8x16: 0.22, 2x16: 0.23 at 32768^2)
#define TR_BW 8
#endif

__global__ __launch_bounds__(256) void k_m4_transpose(const uint32_t *__restrict__ q, const float *__restrict__ s, uint64_t rows,
                                                      uint64_t cols, uint32_t *__restrict__ qt, float *__restrict__ st,
                                                      uint32_t tiles_x)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t tr_lds[];
    uint32_t *tin = tr_lds;                       // [256][33]: the input tile, then (in place) the output tile
    uint32_t *tout = tr_lds;
    // Tile order: workgroups are dealt to the 8 XCDs round-robin; the workgroups that run together on one XCD take 4 x 8 blocks of
    // tiles, so that what goes through that L2 at one time is 1 KiB of every input row and 512 B of every output row, not 128 B.
    uint32_t bj = blockIdx.x % tiles_x;
    uint64_t bi = blockIdx.x / tiles_x;
    {
        const uint32_t ntiles = gridDim.x, tiles_y = ntiles / tiles_x;
        constexpr uint32_t BH = TR_BH, BW = TR_BW;
        if (ntiles % 8 == 0 && tiles_x % BW == 0 && tiles_y % BH == 0) {
            const uint32_t t = (blockIdx.x & 7) * (ntiles / 8) + (blockIdx.x >> 3);
            const uint32_t blk = t / (BH * BW), in = t % (BH * BW), bx = tiles_x / BW;
            bi = (uint64_t)(blk / bx) * BH + in / BW;
            bj = (blk % bx) * BW + in % BW;
        }
    }
    const int tid = threadIdx.x;
    const uint64_t wcols = cols / 8, wrows = rows / 8;
    const uint64_t r0 = bi * TR_T, c0w = (uint64_t)bj * TR_W;       // tile origin: row, word column

    // 1. global -> LDS: 8 lanes x 16 B per tile row
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const int idx = tid + 256 * k, r = idx >> 3, c = idx & 7;
        const u32x4 v = *reinterpret_cast<const u32x4 *>(q + (r0 + r) * wcols + c0w + 4 * c);      // (nt loads / stores: no difference, r4)
        uint32_t *d = tin + r * TR_S + 4 * c;
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
    __syncthreads();
    // 2. 8x8 nibble blocks: block (bg, w) = rows 8bg..8bg+7, word w.  lanes: w_lo = tid&7, bg_lo = (tid>>3)&3.  All four blocks of
    //    a thread are read into registers before anything is written back: ONE tile buffer (33 KiB, four workgroups per CU
    //    instead of two with separate in / out buffers)
    uint32_t wd[4][8];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int w = (tid & 7) + 8 * ((tid >> 5) & 3);
        const int bg = ((tid >> 3) & 3) + 4 * ((tid >> 7) + 2 * k);
#pragma unroll
        for (int r = 0; r < 8; r++) wd[k][r] = tin[(8 * bg + r) * TR_S + w];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int w = (tid & 7) + 8 * ((tid >> 5) & 3);
        const int bg = ((tid >> 3) & 3) + 4 * ((tid >> 7) + 2 * k);
#pragma unroll
        for (int e = 0; e < 8; e++) {
            uint32_t acc = 0;
#pragma unroll
            for (int r = 0; r < 8; r++) acc |= ((wd[k][r] >> nib_shift(e)) & 0xFu) << nib_shift(r);
            tout[(8 * w + e) * TR_S + bg] = acc;
        }
    }
    __syncthreads();
    // 3. LDS -> global: output tile row j (a column of the input tile) = 32 words
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const int idx = tid + 256 * k, r = idx >> 3, c = idx & 7;
        const uint32_t *d = tout + r * TR_S + 4 * c;
        const u32x4 v = {d[0], d[1], d[2], d[3]};
        *reinterpret_cast<u32x4 *>(qt + ((uint64_t)bj * TR_T + r) * wrows + bi * TR_W + 4 * c) = v;
    }
    // tile scales: this 256x256 tile covers a 4x4 patch of the 64x64 scale grid
    if (tid < 16) {
        const uint64_t ti = bi * 4 + (tid >> 2), tj = (uint64_t)bj * 4 + (tid & 3);
        st[tj * (rows / 64) + ti] = s[ti * (cols / 64) + tj];
    }
}

// rows or cols not divisible by 256 (they are multiples of 128): one 64x64 tile per 64-thread workgroup
__global__ __launch_bounds__(64) void k_m4_transpose_small(const uint32_t *__restrict__ q, const float *__restrict__ s, uint64_t rows,
                                                           uint64_t cols, uint32_t *__restrict__ qt, float *__restrict__ st,
                                                           uint32_t tiles_x)
{
    const uint32_t bj = blockIdx.x % tiles_x;
    const uint64_t bi = blockIdx.x / tiles_x;
    const int cb = threadIdx.x & 7, rb = threadIdx.x >> 3;
    const uint64_t wcols = cols / 8, wrows = rows / 8;       // words per row of in / out
    uint32_t w[8];
#pragma unroll
    for (int r = 0; r < 8; r++) w[r] = q[(bi * 64 + rb * 8 + r) * wcols + bj * 8 + cb];
    uint32_t o[8];
#pragma unroll
    for (int e = 0; e < 8; e++) {
        uint32_t acc = 0;
#pragma unroll
        for (int r = 0; r < 8; r++) acc |= ((w[r] >> nib_shift(e)) & 0xFu) << nib_shift(r);
        o[e] = acc;
    }
#pragma unroll
    for (int e = 0; e < 8; e++) qt[((uint64_t)bj * 64 + cb * 8 + e) * wrows + bi * 8 + rb] = o[e];
    if (threadIdx.x == 0) st[(uint64_t)bj * (rows / 64) + bi] = s[bi * tiles_x + bj];
}

extern "C" int clm4_transpose(const int8_t *q, const float *s, uint64_t rows, uint64_t cols, int8_t *qt, float *st, void *stream)
{
    CLV_REQUIRE(q && s && qt && st, "clm4_transpose: null pointer");
    CLV_REQUIRE(rows % 128 == 0 && cols % 128 == 0, "clm4_transpose: rows=%llu cols=%llu must be multiples of 128",
                (unsigned long long)rows, (unsigned long long)cols);
    CLV_REQUIRE(q != qt, "clm4_transpose: in-place transposition is not supported");
    if (!rows || !cols) return CLV_OK;
    if (rows % TR_T == 0 && cols % TR_T == 0) {
        const uint64_t tiles = (rows / TR_T) * (cols / TR_T);
        CLV_REQUIRE(tiles <= 0x7FFFFFFFull, "clm4_transpose: too many tiles");
        const size_t lds = TR_T * TR_S * sizeof(uint32_t);                     // 33 KiB
        hipLaunchKernelGGL(k_m4_transpose, dim3((unsigned)tiles), dim3(256), lds, as_stream(stream), (const uint32_t *)q, s, rows, cols,
                           (uint32_t *)qt, st, (uint32_t)(cols / TR_T));
    } else {
        const uint64_t tiles = (rows / 64) * (cols / 64);
        CLV_REQUIRE(tiles <= 0x7FFFFFFFull, "clm4_transpose: too many tiles");
        hipLaunchKernelGGL(k_m4_transpose_small, dim3((unsigned)tiles), dim3(64), 0, as_stream(stream), (const uint32_t *)q, s, rows, cols,
                           (uint32_t *)qt, st, (uint32_t)(cols / 64));
    }
    CLV_LAUNCH_CHECK();
    return CLV_OK;
}
