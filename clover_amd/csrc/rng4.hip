// rng4.hip -- stochastic-rounding variants: the reference's sequential XORShift stream, consumed in parallel.
//
// The reference draws two 256-bit XORShift values per 64-element block, strictly in block order, from one
// per-object state (CloverVector4.h:690-734, simdxorshift128plus.h:97-109).  As written there, a draw
// never reads part1, so the effective state of each of the 4 generator lanes is ONE 64-bit word a, and a
// draw is   n = T(a),  out = n + a,  a <- n   with the GF(2)-linear map
//     T(a) = t ^ a ^ (t >> 18) ^ (a >> 5),  t = a ^ (a << 23).
// Linearity gives exact jump-ahead: the state before draw i is T^i(a0).  Each workgroup jumps to its own
// position with a table of T^(2^k) (64x64 bit matrices, built once on the host; rng_device.h), its lanes
// jump on to their 8-block segments with one more matrix, and from there T is stepped sequentially -- so the
// GPU produces the same nibbles as the reference's sequential quantize for the same keys, at memory speed,
// in ONE launch per operation: the workgroup that reads the state last also writes the advanced state back.
//
// State in device memory: rng_device.h.
#include "rng_device.h"

#include <stdlib.h>

#include <mutex>
#include <vector>

// ---- tables, one copy per device ------------------------------------------------------------------------
static std::mutex g_pow_mutex;
static uint64_t *g_pow_dev[64];

static void gf2_transpose(const uint64_t *cols, uint64_t *rows)
{
    for (int j = 0; j < 64; j++) {
        uint64_t r = 0;
        for (int i = 0; i < 64; i++) r |= ((cols[i] >> j) & 1ull) << i;
        rows[j] = r;
    }
}

int clv_rng_tables(RngTables *t)
{
    int dev = 0;
    CLV_HIP(hipGetDevice(&dev));
    CLV_REQUIRE(dev >= 0 && dev < 64, "device index %d out of range", dev);
    std::lock_guard<std::mutex> lock(g_pow_mutex);
    if (!g_pow_dev[dev]) {
        // column form first (column i = image of bit i): P[k] = T^(2^k), then M_e = T^(16 e) = (T^16)^e
        std::vector<uint64_t> P((size_t)(RNG_POW_LEVELS + 2 * RNG_SEG_MATS) * 64);
        for (int i = 0; i < 64; i++) P[i] = xs_T(1ull << i);
        for (int k = 1; k < RNG_POW_LEVELS; k++)
            for (int i = 0; i < 64; i++) P[64 * k + i] = gf2_matvec(&P[64 * (k - 1)], P[64 * (k - 1) + i]);
        uint64_t *M = &P[(size_t)RNG_POW_LEVELS * 64];
        for (int i = 0; i < 64; i++) M[i] = 1ull << i;
        for (int e = 1; e < RNG_SEG_MATS; e++)
            for (int i = 0; i < 64; i++) M[64 * e + i] = gf2_matvec(&P[64 * 4], M[64 * (e - 1) + i]);
        // ... and ML_e = T^(64 e) = (T^64)^e for the 32-block segments of the large-vector shape
        uint64_t *ML = &P[(size_t)(RNG_POW_LEVELS + RNG_SEG_MATS) * 64];
        for (int i = 0; i < 64; i++) ML[i] = 1ull << i;
        for (int e = 1; e < RNG_SEG_MATS; e++)
            for (int i = 0; i < 64; i++) ML[64 * e + i] = gf2_matvec(&P[64 * 6], ML[64 * (e - 1) + i]);
        // the device wants rows
        std::vector<uint64_t> R(P.size());
        for (int m = 0; m < RNG_POW_LEVELS + 2 * RNG_SEG_MATS; m++) gf2_transpose(&P[64 * m], &R[64 * m]);
        uint64_t *d = nullptr;
        CLV_HIP(hipMalloc(&d, R.size() * sizeof(uint64_t)));
        CLV_HIP(hipMemcpy(d, R.data(), R.size() * sizeof(uint64_t), hipMemcpyHostToDevice));
        g_pow_dev[dev] = d;
    }
    t->pow_rows = g_pow_dev[dev];
    t->seg_rows = g_pow_dev[dev] + (size_t)RNG_POW_LEVELS * 64;
    t->seg_rows_long = g_pow_dev[dev] + (size_t)(RNG_POW_LEVELS + RNG_SEG_MATS) * 64;
    return CLV_OK;
}

// the 4 nibbles of half an output dword with their noise: words W.x..W.w of one draw, byte `sh` (see k_m4_quantize_strip_st)
__device__ __forceinline__ uint32_t quant_pack4_st(const f32x4 v, float k, const u32x4 W, int sh)
{
#define ST_PRODUCT(x, w) __builtin_fmaf(x, k, sign_onto_nonneg(noise_of(w, sh), x))       /* quant1_st before the conversion */
    const uint32_t h = pack4_of_products(ST_PRODUCT(v.x, W.x), ST_PRODUCT(v.y, W.y), ST_PRODUCT(v.z, W.z), ST_PRODUCT(v.w, W.w));
#undef ST_PRODUCT
    return k < __builtin_inff() ? h : 0u;
}

// ---- vector quantize, stochastic (CloverVector4.h:605-807 with the rnd_* branch) --------------------
// wave = S segments of 8 consecutive blocks; lane (seg = l>>2, k = l&3) generates its segment's draws into LDS
// (S = 16: four blocks per round, two rounds); then lane = 8 elements quantises 8 blocks per sub-step, all loads of a
// round in flight together.  S is picked by size (st_segments): small vectors want many short waves, because one
// wave walks its blocks serially; large ones want the jump-ahead amortised over 128 blocks.
// noise group g = element/8 (draw g>>2, byte g&3), AVX lane j = element%8 -> W[j].
template <int S>
__global__ __launch_bounds__(256) void k_v4_quantize_st(const f32x4 *__restrict__ x, uint32_t *__restrict__ q, float *__restrict__ s,
                                                        uint64_t nblocks, uint64_t *state, uint64_t seq, RngTables T)
{
    typedef StShape<S> Sh;
    __shared__ __attribute__((aligned(16))) uint64_t raw_all[4][Sh::NBR * 2 * 4];   // per wave: NBR blocks x 2 draws x 4 lanes
    __shared__ uint64_t base[4];
    const int wave = threadIdx.x >> 6;
    const int lane = threadIdx.x & 63;
    SegRows<Sh::NSEG> segs;
    segs.load(Sh::seg_table(T), wave * Sh::NSEG);
    rng_workgroup_begin(state, seq, T.pow_rows, blockIdx.x, Sh::SHIFT, 2 * nblocks, base);
    uint64_t *raw = raw_all[wave];
    const uint64_t blk0 = ((uint64_t)blockIdx.x * 4 + wave) * (Sh::SEGLEN * Sh::NSEG);
    const int seg = lane >> 2, k = lane & 3;
    // start of this lane's 8-block segment = T^(16 * e) applied to the workgroup's base state
    uint64_t a = segs.starts(base);
    const int rho = lane & 7;

    if constexpr (Sh::NSEG == 16) {
        // Large vectors: lane = one float4, as in the deterministic kernel.  A round is 16 pieces of 4 consecutive blocks (one per
        // segment): load j reads piece j = one contiguous KiB; a block is a DPP row of 16 lanes (maximum by row rotations); a lane
        // quantises half an output dword -- elements 4c..4c+3 of its block, c = lane & 15: noise group g = c >> 1 (draw g >> 2, byte
        // g & 3), words W[4 (c & 1) .. +3] -- and lane pairs swap halves so that even lanes store piece j, odd lanes piece j + 1.
        const int c = lane & 15, g = c >> 1, odd = lane & 1, sub = lane >> 1;
        for (int r = 0; r < Sh::ROUNDS; r++) {
            f32x4 v[16];
#pragma unroll
            for (int j = 0; j < 16; j++) {                   // the round's loads first: in flight while the generator lanes step
                const uint64_t blk = blk0 + (uint64_t)j * Sh::SEGLEN + Sh::BPR * r + (lane >> 4);
                v[j] = __builtin_nontemporal_load(&x[blk < nblocks ? blk * 16 + c : 0]);
            }
            if (r) __syncthreads();                          // the previous round's noise has been consumed
            a = gen_blocks(a, Sh::BPR, raw + (size_t)(Sh::BPR * seg) * 8, k);
            __syncthreads();
            const u32x4 *noise = reinterpret_cast<const u32x4 *>(raw) + (size_t)((lane >> 4) * 2 + (g >> 2)) * 2 + (c & 1);   // + 16 per piece
            uint32_t half[16];
#pragma unroll
            for (int j = 0; j < 16; j++) {
                const uint64_t blk = blk0 + (uint64_t)j * Sh::SEGLEN + Sh::BPR * r + (lane >> 4);
                float m = fmaxf(fmaxf(__builtin_fabsf(v[j].x), __builtin_fabsf(v[j].y)), fmaxf(__builtin_fabsf(v[j].z), __builtin_fabsf(v[j].w)));
                m = fix_zero_max(row16_max(m));
                half[j] = quant_pack4_st(v[j], 7.0f / m, noise[(size_t)Sh::BPR * 4 * j], g & 3);
                if (c == 0 && blk < nblocks) s[blk] = m;
            }
#pragma unroll
            for (int j = 0; j < 16; j += 2) {
                const uint32_t give = odd ? half[j] : half[j + 1];
                const uint32_t recv = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)give, 0xB1, 0xF, 0xF, false);    // quad_perm [1,0,3,2]
                const uint32_t word = odd ? (recv | (half[j + 1] << 16)) : (half[j] | (recv << 16));
                const uint64_t pb = blk0 + (uint64_t)(j + odd) * Sh::SEGLEN + Sh::BPR * r;                          // first block of the piece
                if (pb + (sub >> 3) < nblocks) __builtin_nontemporal_store(word, &q[pb * 8 + sub]);
            }
        }
        return;
    }
    for (int r = 0; r < Sh::ROUNDS; r++) {
        if (lane < 4 * Sh::NSEG) a = gen_blocks(a, Sh::BPR, raw + (size_t)(Sh::BPR * seg) * 8, k);
        __syncthreads();
        f32x4 lo[Sh::STEPS], hi[Sh::STEPS];
#pragma unroll
        for (int u = 0; u < Sh::STEPS; u++) {
            const uint64_t blk = Sh::block(blk0, r, 8 * u + (lane >> 3));
            const uint64_t i = blk < nblocks ? blk * 8 + rho : 0;           // output dword index (clamped: block 0 always exists)
            lo[u] = __builtin_nontemporal_load(&x[2 * i]);
            hi[u] = __builtin_nontemporal_load(&x[2 * i + 1]);
        }
#pragma unroll
        for (int u = 0; u < Sh::STEPS; u++) {
            const int bl = 8 * u + (lane >> 3);
            const uint64_t blk = Sh::block(blk0, r, bl);
            const float v[8] = {lo[u].x, lo[u].y, lo[u].z, lo[u].w, hi[u].x, hi[u].y, hi[u].z, hi[u].w};
            float m = 0.0f;
#pragma unroll
            for (int e = 0; e < 8; e++) m = fmaxf(m, __builtin_fabsf(v[e]));
            m = fmaxf(m, __shfl_xor(m, 1));
            m = fmaxf(m, __shfl_xor(m, 2));
            m = fmaxf(m, __shfl_xor(m, 4));
            m = fix_zero_max(m);
            const float kq = 7.0f / m;
            const u32x4 *Wp = reinterpret_cast<const u32x4 *>(raw + (size_t)(bl * 2 + (rho >> 2)) * 4);
            const u32x4 W0 = Wp[0], W1 = Wp[1];
            const uint32_t W[8] = {W0.x, W0.y, W0.z, W0.w, W1.x, W1.y, W1.z, W1.w};
            float nz[8];
#pragma unroll
            for (int e = 0; e < 8; e++) nz[e] = noise_of(W[e], rho & 3);
            const uint32_t packed = quant_pack8(v, kq, nz);
            if (blk < nblocks) {
                q[blk * 8 + rho] = packed;
                if (rho == 0) s[blk] = m;
            }
        }
        if (r + 1 < Sh::ROUNDS) __syncthreads();
    }
}

// ---- matrix quantize, stochastic (CloverMatrix4.h:512-766) ----------------------------------------------
// stream order: tile t = bj * (rows/64) + bi (column-block outer), then the tile's 64 rows, two draws each.
__global__ __launch_bounds__(256) void k_m4_quantize_st(const float *__restrict__ A, uint64_t cols, uint32_t *__restrict__ q,
                                                        float *__restrict__ s, uint32_t tiles_x, uint64_t tiles_y, uint64_t *state,
                                                        uint64_t seq, RngTables T)
{
    __shared__ __attribute__((aligned(16))) uint64_t raw[64 * 2 * 4];
    __shared__ float sh[4];
    __shared__ uint64_t base[4];
    const uint32_t bj = blockIdx.x % tiles_x;
    const uint64_t bi = blockIdx.x / tiles_x;
    const uint64_t t = (uint64_t)bj * tiles_y + bi;                 // position of this tile in the stream
    const int tid = threadIdx.x;
    const int o = tid & 7;
    const int r0 = tid >> 3;

    SegRows<8> segs;
    if (tid < 64) segs.load(T.seg_rows, 0);
    rng_workgroup_begin(state, seq, T.pow_rows, t, 7, (uint64_t)tiles_x * tiles_y * 128, base);   // tile = 64 rows = 2^7 draws
    if (tid < 64) {                                                  // wave 0: 8 segments x 4 generator lanes
        const uint64_t a = segs.starts(base);
        if (tid < 32) gen_blocks(a, 8, raw + (size_t)(8 * (tid >> 2)) * 8, tid & 3);
    }

    float v[2][8];
    float m = 0.0f;
#pragma unroll
    for (int p = 0; p < 2; p++) {
        const uint64_t row = bi * 64 + r0 + 32 * p;
        const f32x4 *src = reinterpret_cast<const f32x4 *>(A + row * cols + bj * 64 + o * 8);
        const f32x4 lo = __builtin_nontemporal_load(&src[0]);
        const f32x4 hi = __builtin_nontemporal_load(&src[1]);
        v[p][0] = lo.x; v[p][1] = lo.y; v[p][2] = lo.z; v[p][3] = lo.w;
        v[p][4] = hi.x; v[p][5] = hi.y; v[p][6] = hi.z; v[p][7] = hi.w;
#pragma unroll
        for (int e = 0; e < 8; e++) m = fmaxf(m, __builtin_fabsf(v[p][e]));
    }
    m = wave_max(m);
    if ((tid & 63) == 0) sh[tid >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
    m = fix_zero_max(m);
    const float kq = 7.0f / m;
    if (tid == 0) s[bi * tiles_x + bj] = m;
#pragma unroll
    for (int p = 0; p < 2; p++) {
        const int rl = r0 + 32 * p;                                  // tile row = stream block within the tile
        const u32x4 *Wp = reinterpret_cast<const u32x4 *>(raw + (size_t)(rl * 2 + (o >> 2)) * 4);
        const u32x4 W0 = Wp[0], W1 = Wp[1];
        const uint32_t W[8] = {W0.x, W0.y, W0.z, W0.w, W1.x, W1.y, W1.z, W1.w};
        float nz[8];
#pragma unroll
        for (int e = 0; e < 8; e++) nz[e] = noise_of(W[e], o & 3);
        const uint64_t row = bi * 64 + rl;
        q[(row * cols + bj * 64) / 8 + o] = quant_pack8(v[p], kq, nz);
    }
}

// strip form of the same (see k_m4_quantize_strip, matrix4.hip): workgroup = 64 rows x 4 tiles side by side.  Wave w
// generates the draws of tile bj = 4 sj + w -- stream position t = bj * tiles_y + bi, all four generator lanes -- and
// every lane then reads the 16 bytes of noise words its four elements of a row need.
//
// NV = tiles a workgroup takes one BELOW the other (consecutive bi of one bj are consecutive in the stream).  The generator set-up
// of a wave -- 4 jump-aheads T^(128 t) (~220 VALU) and one GF(2) product per generator lane for its segment start (9 VALU each) --
// was ~60 % of the kernel's instructions with one tile per wave (32 lanes generate 16 draws each).  With NV = 2 the 64 lanes of a
// wave generate 16 segments of 8 rows, the jump-ahead is paid once per two tiles and the generation itself runs at full width.
template <int NV>
__global__ __launch_bounds__(256) void k_m4_quantize_strip_st(const float *__restrict__ A, uint64_t cols, uint32_t *__restrict__ q,
                                                              float *__restrict__ s, uint32_t strips_x, uint32_t tiles_x, uint64_t tiles_y,
                                                              uint64_t *state, uint64_t seq, RngTables T)
{
    __shared__ __attribute__((aligned(16))) uint64_t raw[4][NV * 64 * 2 * 4];  // per tile column: NV x 64 rows x 2 draws x 4 lanes
    __shared__ float sh[NV][4][4];
    const uint32_t sj = blockIdx.x % strips_x;
    const uint64_t bi0 = (uint64_t)(blockIdx.x / strips_x) * NV;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint64_t col = (uint64_t)sj * 256 + 4 * lane;
    const bool live = col < cols;

    // the loads of the first tile row first: the generator work below overlaps their latency
    f32x4 v[16];
#pragma unroll
    for (int r = 0; r < 16; r++)
        v[r] = live ? __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(A + (bi0 * 64 + wave * 16 + r) * cols + col)) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};

    SegRows<8 * NV> segs;
    segs.load(T.seg_rows, 0);
    seq = rng_effective_seq(state, seq);
    const int slot = rng_read_slot(state, seq);
    uint64_t a0[4], b[4];
#pragma unroll
    for (int k = 0; k < 4; k++) a0[k] = state[slot * RNG_SLOT_WORDS + 4 + k];
    const uint64_t bj = (uint64_t)sj * 4 + wave;                      // this wave's tile column (may lie beyond the matrix: then unused)
    const uint64_t t = bj * tiles_y + bi0;
#pragma unroll
    for (int k = 0; k < 4; k++) b[k] = wave_pow_apply(T.pow_rows, a0[k], bj < tiles_x ? t : 0, 7);      // tile = 64 rows = 2^7 draws
    const uint64_t st = segs.starts_from(b);
    if (lane < 32 * NV) gen_blocks(st, 8, raw[wave] + (size_t)(8 * (lane >> 2)) * 8, lane & 3);
    const uint64_t a0w = wave == 0 ? a0[0] : wave == 1 ? a0[1] : wave == 2 ? a0[2] : a0[3];
    rng_commit(state, seq, slot, T.pow_rows, a0w, (uint64_t)tiles_x * tiles_y * 128);

    const int tl = lane >> 4;
    // elements 4c..4c+3 of a tile row (c = lane & 15): noise group g = c >> 1 (draw g >> 2, byte g & 3), words W[4 (c & 1) .. +3]
    const int c = lane & 15, g = c >> 1;
    const int odd = lane & 1;
#pragma unroll
    for (int i = 0; i < NV; i++) {
        const uint64_t bi = bi0 + i;
        if (bi >= tiles_y) break;                                      // odd tile count: the last workgroup row has one tile
        const uint64_t row0 = bi * 64 + wave * 16;
        if (i > 0) {
#pragma unroll
            for (int r = 0; r < 16; r++)
                v[r] = live ? __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(A + (row0 + r) * cols + col)) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        }
        float m = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; r++)
            m = fmaxf(m, fmaxf(fmaxf(__builtin_fabsf(v[r].x), __builtin_fabsf(v[r].y)), fmaxf(__builtin_fabsf(v[r].z), __builtin_fabsf(v[r].w))));
        m = row16_max(m);
        if ((lane & 15) == 0) sh[i][wave][lane >> 4] = m;
        __syncthreads();                                               // tile maxima (and, the first time, every tile's draws) are in LDS
        m = fix_zero_max(fmaxf(fmaxf(sh[i][0][tl], sh[i][1][tl]), fmaxf(sh[i][2][tl], sh[i][3][tl])));
        const float k = 7.0f / m;
        if (wave == 0 && (lane & 15) == 0 && live) s[bi * tiles_x + sj * 4 + tl] = m;
        const u32x4 *noise = reinterpret_cast<const u32x4 *>(raw[tl] + (size_t)i * 64 * 8) + (g >> 2) * 2 + (c & 1);     // + 4 per tile row
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            const int tr = wave * 16 + r;                              // tile row = stream block within the tile
            const uint32_t h0 = quant_pack4_st(v[r], k, noise[4 * tr], g & 3), h1 = quant_pack4_st(v[r + 1], k, noise[4 * (tr + 1)], g & 3);
            const uint32_t give = odd ? h0 : h1;
            const uint32_t recv = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)give, 0xB1, 0xF, 0xF, false);    // quad_perm [1,0,3,2]
            const uint32_t word = odd ? (recv | (h1 << 16)) : (h0 | (recv << 16));
            if (live) __builtin_nontemporal_store(word, &q[((row0 + r + odd) * cols + (col & ~7ull)) / 8]);
        }
    }
}

// ---- host side ---------------------------------------------------------------------------------------------
// segments (of 8 blocks) per wave for the vector kernels.  The choice never changes results, only speed;
// clv_rng_set_segments (clover_hip.h: a tuning knob, and how the tests reach every kernel shape at small sizes) or
// CLV_ST_SEGMENTS=1|4|16|64 force one shape.
static int g_st_forced = [] { const char *e = getenv("CLV_ST_SEGMENTS"); return e ? atoi(e) : 0; }();

extern "C" int clv_rng_set_segments(int s)
{
    CLV_REQUIRE(s == 0 || s == 1 || s == 4 || s == 16 || s == 64, "clv_rng_set_segments: %d is not one of 0, 1, 4, 16, 64", s);
    g_st_forced = s;
    return CLV_OK;
}

// long_ok: the 16 x 32-block shape pays only where the kernel is VALU-bound (scaleAndAdd: 0.57 -> 0.50 ms at n = 2^30); the
// quantize kernels stream 8x the bytes per element and lose 15 % with it (more rounds, each with its barrier)
int clv_st_segments(uint64_t nblocks, bool long_ok)
{
    static const int env_forced = [] { const char *e = getenv("CLV_ST_SEGMENTS"); return e ? atoi(e) : 0; }();      // A/B runs
    if (env_forced == 1 || env_forced == 4 || env_forced == 16 || env_forced == 64) return env_forced;
    if (g_st_forced == 1 || g_st_forced == 4 || g_st_forced == 16 || g_st_forced == 64) return g_st_forced;
    return nblocks <= 8192 ? 1 : nblocks <= (1u << 18) ? 4 : (nblocks <= (1u << 21) || !long_ok) ? 16 : 64;
}

int clv4_quantize_stochastic(const float *x, uint64_t n_pad, int8_t *q, float *s, uint64_t *rng, hipStream_t st)
{
    RngTables T;
    int rc = clv_rng_tables(&T);
    if (rc) return rc;
    const uint64_t nb = n_pad / 64;
    const uint64_t seq = clv_rng_seq_for(rng, st);
#define QST_LAUNCH(S)                                                                                                          \
    hipLaunchKernelGGL(k_v4_quantize_st<S>, dim3((unsigned)((nb + 32 * S - 1) / (32 * S))), dim3(256), 0, st, (const f32x4 *)x, \
                       (uint32_t *)q, s, nb, rng, seq, T)
    switch (clv_st_segments(nb, false)) {
    case 1: QST_LAUNCH(1); break;
    case 4: QST_LAUNCH(4); break;
    case 16: QST_LAUNCH(16); break;
    default: QST_LAUNCH(64); break;
    }
#undef QST_LAUNCH
    CLV_LAUNCH_CHECK();
    return CLV_OK;
}

int clm4_quantize_stochastic(const float *A, uint64_t rows, uint64_t cols, int8_t *q, float *s, uint64_t *rng, hipStream_t st)
{
    RngTables T;
    int rc = clv_rng_tables(&T);
    if (rc) return rc;
    const uint64_t tiles = (rows / 64) * (cols / 64);
    static const bool tile_kernel = getenv("CLV_M4Q_TILE") != nullptr;       // A/B switch: the older one-tile-per-workgroup kernel
    if (!tile_kernel) {
        const uint32_t strips_x = (uint32_t)((cols + 255) / 256);
        static const bool one_tile = getenv("CLV_M4Q_NV1") != nullptr;       // A/B switch: one tile row per workgroup
        if (one_tile || rows / 64 < 2)
            hipLaunchKernelGGL(k_m4_quantize_strip_st<1>, dim3((unsigned)((rows / 64) * strips_x)), dim3(256), 0, st, A, cols, (uint32_t *)q, s,
                               strips_x, (uint32_t)(cols / 64), rows / 64, rng, clv_rng_seq_for(rng, st), T);
        else
            hipLaunchKernelGGL(k_m4_quantize_strip_st<2>, dim3((unsigned)(((rows / 64 + 1) / 2) * strips_x)), dim3(256), 0, st, A, cols, (uint32_t *)q,
                               s, strips_x, (uint32_t)(cols / 64), rows / 64, rng, clv_rng_seq_for(rng, st), T);
        CLV_LAUNCH_CHECK();
        return CLV_OK;
    }
    hipLaunchKernelGGL(k_m4_quantize_st, dim3((unsigned)tiles), dim3(256), 0, st, A, cols, (uint32_t *)q, s, (uint32_t)(cols / 64),
                       rows / 64, rng, clv_rng_seq_for(rng, st), T);
    CLV_LAUNCH_CHECK();
    return CLV_OK;
}
