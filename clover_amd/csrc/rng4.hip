// rng4.hip -- stochastic-rounding variants: the reference's sequential XORShift stream, consumed in parallel.
//
// The reference draws two 256-bit XORShift values per 64-element block, strictly in block order, from one
// per-object state (CloverVector4.h:690-734, simdxorshift128plus.h:97-109).  As written there, a draw
// never reads part1, so the effective state of each of the 4 generator lanes is ONE 64-bit word a, and a
// draw is   n = T(a),  out = n + a,  a <- n   with the GF(2)-linear map
//     T(a) = t ^ a ^ (t >> 18) ^ (a >> 5),  t = a ^ (a << 23).
// Linearity gives exact jump-ahead: the state before draw i is T^i(a0).  A prefix kernel computes the
// start state of every 8-block segment from a table of T^(2^k) (64x64 bit matrices, built once on the
// host), then the streaming kernels step T sequentially inside their segment -- so the GPU produces the
// same nibbles as the reference's sequential quantize for the same keys, at memory speed.
//
// State in device memory: uint64 st[8] = part1 lanes 0..3 (s0), part2 lanes 0..3 (s1).
#include "common.h"

#include <mutex>
#include <vector>

#define RNG_POW_LEVELS 56
#define RNG_SEG_MATS 64          // T^(16*e), e = 0..63: start of 8-block segment e relative to a workgroup base

__host__ __device__ __forceinline__ uint64_t xs_T(uint64_t a)
{
    const uint64_t t = a ^ (a << 23);
    return t ^ a ^ (t >> 18) ^ (a >> 5);
}

// r = M * v over GF(2); M is 64 columns (column i = image of bit i)
__host__ __device__ __forceinline__ uint64_t gf2_matvec(const uint64_t *M, uint64_t v)
{
    uint64_t r = 0;
#pragma unroll 8
    for (int i = 0; i < 64; i++) r ^= (0 - ((v >> i) & 1ull)) & M[i];
    return r;
}

// v <- T^e (v) using the power table P[k] = T^(2^k), starting at level k0 (i.e. T^(e * 2^k0))
__device__ __forceinline__ uint64_t gf2_pow_apply(const uint64_t *P, uint64_t v, uint64_t e, int k0)
{
    for (int k = k0; e != 0 && k < RNG_POW_LEVELS; k++, e >>= 1)
        if (e & 1ull) v = gf2_matvec(P + 64 * k, v);
    return v;
}

// ---- power table, one copy per device ----------------------------------------------------------------
static std::mutex g_pow_mutex;
static uint64_t *g_pow_dev[64];

static int rng_pow_table(const uint64_t **table)
{
    int dev = 0;
    CLV_HIP(hipGetDevice(&dev));
    CLV_REQUIRE(dev >= 0 && dev < 64, "device index %d out of range", dev);
    std::lock_guard<std::mutex> lock(g_pow_mutex);
    if (!g_pow_dev[dev]) {
        std::vector<uint64_t> P((size_t)(RNG_POW_LEVELS + RNG_SEG_MATS) * 64);
        for (int i = 0; i < 64; i++) P[i] = xs_T(1ull << i);
        for (int k = 1; k < RNG_POW_LEVELS; k++)
            for (int i = 0; i < 64; i++) P[64 * k + i] = gf2_matvec(&P[64 * (k - 1)], P[64 * (k - 1) + i]);
        // segment matrices behind the power levels: M_0 = I, M_(e+1) = T^16 * M_e
        uint64_t *M = &P[(size_t)RNG_POW_LEVELS * 64];
        for (int i = 0; i < 64; i++) M[i] = 1ull << i;
        for (int e = 1; e < RNG_SEG_MATS; e++)
            for (int i = 0; i < 64; i++) M[64 * e + i] = gf2_matvec(&P[64 * 4], M[64 * (e - 1) + i]);
        uint64_t *d = nullptr;
        CLV_HIP(hipMalloc(&d, P.size() * sizeof(uint64_t)));
        CLV_HIP(hipMemcpy(d, P.data(), P.size() * sizeof(uint64_t), hipMemcpyHostToDevice));
        g_pow_dev[dev] = d;
    }
    *table = g_pow_dev[dev];
    return CLV_OK;
}

// segment matrices live right behind the power levels; valid once rng_pow_table() ran on this device
static const uint64_t *rng_segmat()
{
    int dev = 0;
    (void)hipGetDevice(&dev);
    return g_pow_dev[dev] + (size_t)RNG_POW_LEVELS * 64;
}

// ---- prefix kernel ---------------------------------------------------------------------------------------
// starts[idx*4 + k] = T^(idx << shift)(a0[k])  for idx < count; the last 8 threads produce the state after
// `total` draws: fin[0..3] = T^(total-1) (part1), fin[4..7] = T^total (part2).
__global__ __launch_bounds__(256) void k_rng_prefix(const uint64_t *__restrict__ state, const uint64_t *__restrict__ P,
                                                    uint64_t count, int shift, uint64_t total, uint64_t *__restrict__ starts,
                                                    uint64_t *__restrict__ fin)
{
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < count * 4) {
        const int k = (int)(t & 3);
        starts[t] = gf2_pow_apply(P, state[4 + k], t >> 2, shift);
    } else if (t < count * 4 + 8) {
        const int u = (int)(t - count * 4);
        const int k = u & 3;
        if (total == 0) fin[u] = state[u];
        else fin[u] = gf2_pow_apply(P, state[4 + k], (u < 4) ? total - 1 : total, 0);
    }
}

__global__ void k_rng_commit(uint64_t *__restrict__ state, const uint64_t *__restrict__ fin)
{
    if (threadIdx.x < 8) state[threadIdx.x] = fin[threadIdx.x];
}

// noise lane: byte `sh` of W, as the reference builds it (mask, shift left, int->float, * 2^-31)
__device__ __forceinline__ float noise_of(uint32_t W, int sh)
{
    return (float)(int)((W & 0x7F7F7F7Fu) << (8 * sh)) * (1.0f / 2147483648.0f);
}

// generate the two draws of `nblk` consecutive blocks for generator lane k and store the raw 64-bit outputs
// at raw[(blk*2 + draw)*4 + k]  (so W[2k], W[2k+1] of a draw are the two dwords of entry k)
__device__ __forceinline__ uint64_t gen_blocks(uint64_t a, int nblk, uint64_t *raw, int k)
{
#pragma unroll
    for (int i = 0; i < nblk * 2; i++) {
        const uint64_t n = xs_T(a);
        raw[i * 4 + k] = n + a;
        a = n;
    }
    return a;
}

// ---- vector quantize, stochastic (CloverVector4.h:605-807 with the rnd_* branch) --------------------
// wave = 128 consecutive blocks = 16 segments of 8; lane (seg = l>>2, k = l&3) generates its segment's
// draws 4 blocks at a time into LDS; then lane = 8 elements quantises 8 blocks per sub-step.
// noise group g = element/8 (draw g>>2, byte g&3), AVX lane j = element%8 -> W[j].
#define SQ_WAVE_BLOCKS 128

__global__ __launch_bounds__(256) void k_v4_quantize_st(const f32x4 *__restrict__ x, uint32_t *__restrict__ q, float *__restrict__ s,
                                                        uint64_t nblocks, const uint64_t *__restrict__ starts,
                                                        const uint64_t *__restrict__ segmat)
{
    __shared__ __attribute__((aligned(16))) uint64_t raw_all[4][64 * 2 * 4];   // per wave: 64 blocks x 2 draws x 4 lanes
    const int wave = threadIdx.x >> 6;
    const int lane = threadIdx.x & 63;
    uint64_t *raw = raw_all[wave];
    const uint64_t w = (uint64_t)blockIdx.x * 4 + wave;            // global wave index
    const uint64_t blk0 = w * SQ_WAVE_BLOCKS;
    const int seg = lane >> 2, k = lane & 3;
    // start of this lane's 8-block segment = T^(16 * e) applied to the workgroup's base state (one matvec)
    uint64_t a = gf2_matvec(segmat + 64 * (wave * 16 + seg), starts[(uint64_t)blockIdx.x * 4 + k]);

    for (int r = 0; r < 2; r++) {
        // 4 blocks of this lane's segment: local block id = 4*seg + i
        a = gen_blocks(a, 4, raw + (size_t)(4 * seg) * 8, k);
        __syncthreads();
#pragma unroll 2
        for (int u = 0; u < 8; u++) {
            const int bl = 8 * u + (lane >> 3);                     // local block 0..63
            const int rho = lane & 7;
            const uint64_t blk = blk0 + (uint64_t)(bl >> 2) * 8 + 4 * r + (bl & 3);
            if (blk < nblocks) {
                const uint64_t i = blk * 8 + rho;                   // output dword index
                const f32x4 lo = __builtin_nontemporal_load(&x[2 * i]);
                const f32x4 hi = __builtin_nontemporal_load(&x[2 * i + 1]);
                const float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
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
                q[i] = quant_pack8(v, kq, nz);
                if (rho == 0) s[blk] = m;
            }
        }
        __syncthreads();
    }
}

// ---- matrix quantize, stochastic (CloverMatrix4.h:512-766) ----------------------------------------------
// stream order: tile t = bj * (rows/64) + bi (column-block outer), then the tile's 64 rows, two draws each.
__global__ __launch_bounds__(256) void k_m4_quantize_st(const float *__restrict__ A, uint64_t cols, uint32_t *__restrict__ q,
                                                        float *__restrict__ s, uint32_t tiles_x, uint64_t tiles_y,
                                                        const uint64_t *__restrict__ starts, const uint64_t *__restrict__ segmat)
{
    __shared__ __attribute__((aligned(16))) uint64_t raw[64 * 2 * 4];
    __shared__ float sh[4];
    const uint32_t bj = blockIdx.x % tiles_x;
    const uint64_t bi = blockIdx.x / tiles_x;
    const uint64_t t = (uint64_t)bj * tiles_y + bi;                 // position of this tile in the stream
    const int tid = threadIdx.x;
    const int o = tid & 7;
    const int r0 = tid >> 3;

    if (tid < 32) {                                                  // 8 segments x 4 generator lanes
        const int seg = tid >> 2, k = tid & 3;
        gen_blocks(gf2_matvec(segmat + 64 * seg, starts[t * 4 + k]), 8, raw + (size_t)(8 * seg) * 8, k);
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

// ---- mvm epilogue, stochastic (CloverMatrix4.h:919-1080) ------------------------------------------------
// one wave per 64-row group; lane = output row l.  The dots sit pre-transposed in the reference's
// block_values, so noise group g, AVX lane j lands on row 8j+g: row l uses group l&7, W[l>>3].
__global__ __launch_bounds__(256) void k_m4_requantize_st(const float *__restrict__ d, uint64_t ngroups, uint32_t *__restrict__ r,
                                                          float *__restrict__ sr, const uint64_t *__restrict__ starts)
{
    __shared__ __attribute__((aligned(16))) uint64_t raw_all[4][2 * 4];
    const int wave = threadIdx.x >> 6;
    const int lane = threadIdx.x & 63;
    const uint64_t g = (uint64_t)blockIdx.x * 4 + wave;
    uint64_t *raw = raw_all[wave];
    if (g < ngroups && lane < 4) gen_blocks(starts[g * 4 + lane], 1, raw, lane);
    __syncthreads();
    if (g >= ngroups) return;
    const float dv = d[g * 64 + lane];
    const int grp = lane & 7, j = lane >> 3;
    const uint32_t *W = reinterpret_cast<const uint32_t *>(raw + (size_t)(grp >> 2) * 4);
    const float noise = noise_of(W[j], grp & 3);

    float m = wave_max(__builtin_fabsf(dv));
    m = fix_zero_max(m);
    const float kq = 7.0f / m;
    const int qv = quant1(dv, kq, noise);
    uint32_t w = ((uint32_t)qv & 0xFu) << nib_shift(lane & 7);
    w |= __shfl_xor(w, 1);
    w |= __shfl_xor(w, 2);
    w |= __shfl_xor(w, 4);
    if ((lane & 7) == 0) r[g * 8 + (lane >> 3)] = w;
    if (lane == 0) sr[g] = m;
}

// ---- host side ---------------------------------------------------------------------------------------------
// runs the prefix for `count` start states spaced 2^shift draws apart, `total` draws consumed overall
static int rng_prefix(uint64_t *state, uint64_t count, int shift, uint64_t total, uint64_t **starts, uint64_t **fin, hipStream_t st)
{
    const uint64_t *P = nullptr;
    int rc = rng_pow_table(&P);
    if (rc) return rc;
    void *ws = nullptr;
    // the row-dot scratch of clm4_mvm may occupy the front of the internal workspace: keep rng data behind it
    rc = clv_internal_workspace(&ws, (count * 4 + 8) * sizeof(uint64_t));
    if (rc) return rc;
    *starts = (uint64_t *)ws;
    *fin = *starts + count * 4;
    const uint64_t threads = count * 4 + 8;
    hipLaunchKernelGGL(k_rng_prefix, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, state, P, count, shift, total, *starts, *fin);
    CLV_LAUNCH_CHECK();
    return CLV_OK;
}

const uint64_t *clv_rng_segmat() { return rng_segmat(); }

// exported to the other translation units (next4.hip)
int clv_rng_prefix(uint64_t *state, uint64_t count, int shift, uint64_t total, uint64_t **starts, uint64_t **fin, hipStream_t st)
{
    return rng_prefix(state, count, shift, total, starts, fin, st);
}

int clv_rng_commit(uint64_t *state, const uint64_t *fin, hipStream_t st)
{
    hipLaunchKernelGGL(k_rng_commit, dim3(1), dim3(64), 0, st, state, fin);
    CLV_LAUNCH_CHECK();
    return CLV_OK;
}

int clv4_quantize_stochastic(const float *x, uint64_t n_pad, int8_t *q, float *s, uint64_t *rng, hipStream_t st)
{
    const uint64_t nb = n_pad / 64;
    const uint64_t wgs = (nb + 4 * SQ_WAVE_BLOCKS - 1) / (4 * SQ_WAVE_BLOCKS);   // one base state per workgroup = 512 blocks = 2^10 draws
    uint64_t *starts, *fin;
    int rc = rng_prefix(rng, wgs, 10, 2 * nb, &starts, &fin, st);
    if (rc) return rc;
    hipLaunchKernelGGL(k_v4_quantize_st, dim3((unsigned)wgs), dim3(256), 0, st, (const f32x4 *)x, (uint32_t *)q, s, nb, starts,
                       rng_segmat());
    CLV_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_rng_commit, dim3(1), dim3(64), 0, st, rng, fin);
    CLV_LAUNCH_CHECK();
    return CLV_OK;
}

int clm4_quantize_stochastic(const float *A, uint64_t rows, uint64_t cols, int8_t *q, float *s, uint64_t *rng, hipStream_t st)
{
    const uint64_t tiles = (rows / 64) * (cols / 64);
    uint64_t *starts, *fin;
    int rc = rng_prefix(rng, tiles, 7, tiles * 128, &starts, &fin, st);      // one base per tile = 64 rows = 2^7 draws
    if (rc) return rc;
    hipLaunchKernelGGL(k_m4_quantize_st, dim3((unsigned)tiles), dim3(256), 0, st, A, cols, (uint32_t *)q, s, (uint32_t)(cols / 64),
                       rows / 64, starts, rng_segmat());
    CLV_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_rng_commit, dim3(1), dim3(64), 0, st, rng, fin);
    CLV_LAUNCH_CHECK();
    return CLV_OK;
}

// d lives at the front of the internal workspace (clm4_mvm put it there): the prefix data goes behind it
int clm4_requantize_stochastic(const float *d, uint64_t rows, int8_t *r, float *sr, uint64_t *rng, hipStream_t st)
{
    const uint64_t ng = rows / 64;
    const uint64_t *P = nullptr;
    int rc = rng_pow_table(&P);
    if (rc) return rc;
    void *ws = nullptr;
    const uint64_t d_bytes = (rows * sizeof(float) + 255) & ~255ull;
    rc = clv_internal_workspace(&ws, d_bytes + (ng * 4 + 8) * sizeof(uint64_t));
    if (rc) return rc;
    if ((const void *)d != ws) { clv_set_error("clm4_requantize_stochastic: workspace moved"); return CLV_ERR_INVALID; }
    uint64_t *starts = (uint64_t *)((char *)ws + d_bytes);
    uint64_t *fin = starts + ng * 4;
    const uint64_t threads = ng * 4 + 8;
    hipLaunchKernelGGL(k_rng_prefix, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, rng, P, ng, 1, 2 * ng, starts, fin);
    CLV_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_m4_requantize_st, dim3((unsigned)((ng + 3) / 4)), dim3(256), 0, st, d, ng, (uint32_t *)r, sr, starts);
    CLV_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_rng_commit, dim3(1), dim3(64), 0, st, rng, fin);
    CLV_LAUNCH_CHECK();
    return CLV_OK;
}
