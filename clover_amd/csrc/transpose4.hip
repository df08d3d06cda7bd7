// transpose4.hip -- CloverMatrix4::transpose.
// One of the callers either side of the hot path.  With mvm these are the five steps of the reference's quantized IHT / GD
// iterations (test/performance/01_measure.h:923-946, 999-1021), so x, t1..t3 can stay in HBM across iterations.
#include "common.h"

// =================================================================================================
// f2  CloverMatrix4::transpose (CloverMatrix4.h:1549-1663): out(j,i) = in(i,j) nibble-wise, tile scales
//     transposed (the reference calls IPP for those, :1657-1658).
//     Round 6.  workgroup = 256 x 256 elements; both the reads and the writes are 128-byte runs (a row of the tile is 128 B on either
//     side).  A lane loads 8 rows x 16 B straight into registers (no LDS pass on the way in), transposes its four 8 x 8 nibble blocks
//     with a three-stage butterfly -- nibbles between row pairs (shift + v_bfi), bytes and halfwords with v_perm_b32: 32 VALU per block
//     = 0.5 per element, where rounds 2-5 assembled every output word from 8 shift-mask-or triples (3 per element) -- and only the OUTPUT
//     goes through LDS: ds_write_b32 into an XOR-swizzled [256][32] image (column ^ 4 * (row >> 5): the 32 lanes of a store group hit 32
//     banks), ds_read_b128 + 16-byte stores on the way out (no padding, so the 16-byte accesses stay aligned).  Edge tiles are masked:
//     rows / cols are multiples of 128, not of 256, and every shape runs on this kernel (the 64-thread kernel for ragged shapes is gone).
//     Algorithmic bytes: 2 * (1/2 + 4/4096) per element.
// =================================================================================================
#define TR_T 256                      // tile edge in elements
#define TR_W (TR_T / 8)               // 32 words per tile row
#ifndef TR_BH
#define TR_BH 8                      // tiles per XCD block: BH x BW (4x8: 0.214 ms, 8x8: 0.20-0.21, 16x8: 0.21, 8x16: 0.22, 2x16: 0.23 at 32768^2, r4)
#define TR_BW 8
#endif

// 8 x 8 nibble transpose: W[r] = the word of input row r (element e at nib_shift(e): EVEN elements in the high nibble of their byte),
// O[e] = the word of output row e.  In "position" space (position p = bits 4p .. 4p + 3, element e sits at position e ^ 1) this is the
// plain matrix transpose of the rows taken in the order r ^ 1, read out in the order e ^ 1: the two permutations are register renaming.
__device__ __forceinline__ void transpose8x8_nibbles(const uint32_t W[8], uint32_t O[8])
{
    uint32_t R[8];
#pragma unroll
    for (int r = 0; r < 8; r++) R[r] = W[r ^ 1];
#pragma unroll
    for (int i = 0; i < 8; i += 2) {                                       // nibbles: (row 2i, position 2j + 1) <-> (row 2i + 1, position 2j)
        const uint32_t a = R[i], b = R[i + 1];
        R[i] = (a & 0x0F0F0F0Fu) | ((b << 4) & 0xF0F0F0F0u);               // v_lshlrev + v_bfi
        R[i + 1] = (b & 0xF0F0F0F0u) | ((a >> 4) & 0x0F0F0F0Fu);
    }
#pragma unroll
    for (int r = 0; r < 8; r++)
        if ((r & 2) == 0) {                                                // bytes: rows r, r + 2
            const uint32_t a = R[r], b = R[r + 2];
            R[r] = __builtin_amdgcn_perm(b, a, 0x06020400u);               // [a.b0, b.b0, a.b2, b.b2]
            R[r + 2] = __builtin_amdgcn_perm(b, a, 0x07030501u);           // [a.b1, b.b1, a.b3, b.b3]
        }
#pragma unroll
    for (int r = 0; r < 4; r++) {                                          // halfwords: rows r, r + 4
        const uint32_t a = R[r], b = R[r + 4];
        R[r] = __builtin_amdgcn_perm(b, a, 0x05040100u);                   // [a.lo, b.lo]
        R[r + 4] = __builtin_amdgcn_perm(b, a, 0x07060302u);               // [a.hi, b.hi]
    }
#pragma unroll
    for (int e = 0; e < 8; e++) O[e] = R[e ^ 1];
}

template <bool EDGE, bool NT>
__device__ __forceinline__ void transpose_tile(const uint32_t *__restrict__ q, uint32_t *__restrict__ qt, uint64_t rows, uint64_t cols, uint64_t wcols,
                                               uint64_t wrows, uint64_t r0, uint64_t c0w, uint64_t bi, uint32_t bj, uint32_t *tl)
{
    const int tid = threadIdx.x;
    // 1. global -> registers: lane = (16-byte column group c, row group rgi): rows 8 rgi .. 8 rgi + 7 of the tile, words 4 c .. 4 c + 3.
    //    A load instruction of a wave reads 8 rows x 128 B.  Beyond the matrix (edge tiles): a clamped address, then zeros.
    const int c = tid & 7, rgi = tid >> 3;
    u32x4 v[8];
    {
        const bool col_in = !EDGE || c0w + 4 * c < wcols;
        const uint64_t colw = col_in ? c0w + 4 * c : 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const uint64_t row = r0 + 8 * rgi + i;
            const bool in = !EDGE || (col_in && row < rows);
            const u32x4 *lp = reinterpret_cast<const u32x4 *>(q + (in ? row : 0) * wcols + colw);
            const u32x4 ld = NT ? __builtin_nontemporal_load(lp) : *lp;
            v[i] = in ? ld : u32x4{0u, 0u, 0u, 0u};
        }
    }
    // 2. four 8 x 8 blocks per lane; output row j = 32 c + 8 cw + e of the tile, word rgi
#pragma unroll
    for (int cw = 0; cw < 4; cw++) {
        uint32_t W[8], O[8];
#pragma unroll
        for (int i = 0; i < 8; i++) W[i] = cw == 0 ? v[i].x : cw == 1 ? v[i].y : cw == 2 ? v[i].z : v[i].w;
#ifdef TR_VARIANT_NOMATH                  /* tools/build_variant.py: timing-only variants, results wrong by construction */
#pragma unroll
        for (int e = 0; e < 8; e++) O[e] = W[e];
#else
        transpose8x8_nibbles(W, O);
#endif
#pragma unroll
        for (int e = 0; e < 8; e++) tl[(32 * c + 8 * cw + e) * TR_W + (rgi ^ (4 * c))] = O[e];
    }
    __syncthreads();
    // 3. LDS -> global: output tile row j (a column of the input tile) = 32 words = 8 lanes x 16 B; all reads, then the stores
    u32x4 o[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const int idx = tid + 256 * k, j = idx >> 3, qd = idx & 7;
        o[k] = *reinterpret_cast<const u32x4 *>(tl + j * TR_W + ((4 * qd) ^ (4 * (j >> 5))));
#ifdef TR_VARIANT_COPY                    /* the access pattern alone: what was loaded goes straight to the output addresses */
        o[k] = v[k];
#endif
    }
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const int idx = tid + 256 * k, j = idx >> 3, qd = idx & 7;
        const uint64_t orow = (uint64_t)bj * TR_T + j, ocol = bi * TR_W + 4 * qd;
        if (!EDGE || (orow < cols && ocol < wrows)) {
            u32x4 *sp = reinterpret_cast<u32x4 *>(qt + orow * wrows + ocol);
            if (NT) __builtin_nontemporal_store(o[k], sp); else *sp = o[k];
        }
    }
}

template <bool NT>
__global__ __launch_bounds__(256) void k_m4_transpose(const uint32_t *__restrict__ q, const float *__restrict__ s, uint64_t rows,
                                                      uint64_t cols, uint32_t *__restrict__ qt, float *__restrict__ st,
                                                      uint32_t tiles_x)
{
    __shared__ __attribute__((aligned(16))) uint32_t tl[TR_T * TR_W];     // the OUTPUT tile, [256][32], column ^ 4 * (row >> 5)
    // Tile order: workgroups are dealt to the 8 XCDs round-robin; the workgroups that run together on one XCD take BH x BW blocks of
    // tiles, so that what goes through that L2 at one time is 1 KiB of every input row and of every output row, not 128 B.
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
    const uint64_t wcols = cols / 8, wrows = rows / 8;
    const uint64_t r0 = bi * TR_T, c0w = (uint64_t)bj * TR_W;       // tile origin: row, word column
    // interior tiles (all but the last tile row / column of a ragged shape) take the unguarded body: the guards of the edge body cost
    // the loads their overlap (a select per load keeps them in flight together, an exec-masked branch per load does not)
    if (r0 + TR_T <= rows && c0w + TR_W <= wcols) transpose_tile<false, NT>(q, qt, rows, cols, wcols, wrows, r0, c0w, bi, bj, tl);
    else transpose_tile<true, NT>(q, qt, rows, cols, wcols, wrows, r0, c0w, bi, bj, tl);
    const int tid = threadIdx.x;
    // tile scales: this 256x256 tile covers a 4x4 patch of the 64x64 scale grid
    if (tid < 16) {
        const uint64_t ti = bi * 4 + (tid >> 2), tj = (uint64_t)bj * 4 + (tid & 3);
        if (ti < rows / 64 && tj < cols / 64) st[tj * (rows / 64) + ti] = s[ti * (cols / 64) + tj];
    }
}

extern "C" int clm4_transpose(const int8_t *q, const float *s, uint64_t rows, uint64_t cols, int8_t *qt, float *st, void *stream)
{
    CLV_REQUIRE(q && s && qt && st, "clm4_transpose: null pointer");
    CLV_REQUIRE(rows % 128 == 0 && cols % 128 == 0, "clm4_transpose: rows=%llu cols=%llu must be multiples of 128",
                (unsigned long long)rows, (unsigned long long)cols);
    CLV_REQUIRE(q != qt, "clm4_transpose: in-place transposition is not supported");
    if (!rows || !cols) return CLV_OK;
    const uint64_t tiles_x = (cols + TR_T - 1) / TR_T, tiles = ((rows + TR_T - 1) / TR_T) * tiles_x;
    CLV_REQUIRE(tiles <= 0x7FFFFFFFull, "clm4_transpose: too many tiles");
    // streaming (nt) loads and stores once input + output cannot live in the 256 MiB Infinity Cache: 0.64 -> 0.68 of 8 TB/s at 32768^2 (the
    // same tile-shaped COPY, no transposition, reaches 0.65 / 0.70, a linear copy of this shape 0.735: tools/tile_copy_probe.hip);
    // below that size the default policy keeps the transposed matrix in cache for whoever multiplies by it next
    if (rows * cols > (256ull << 20))
        hipLaunchKernelGGL(k_m4_transpose<true>, dim3((unsigned)tiles), dim3(256), 0, as_stream(stream), (const uint32_t *)q, s, rows, cols, (uint32_t *)qt,
                           st, (uint32_t)tiles_x);
    else
        hipLaunchKernelGGL(k_m4_transpose<false>, dim3((unsigned)tiles), dim3(256), 0, as_stream(stream), (const uint32_t *)q, s, rows, cols, (uint32_t *)qt,
                           st, (uint32_t)tiles_x);
    CLV_LAUNCH_CHECK();
    return CLV_OK;
}
