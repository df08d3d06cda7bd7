// vector4.hip -- CloverVector4 hot path on gfx950: quantize, restore, dot (exact order / fast), word sums.
//
// Data layout in HBM = the reference's (CloverVector4.h:68-103): n_pad/2 value bytes (element 2i in the
// high nibble of byte i), one fp32 scale per 64 elements.  The streaming kernels map a lane to one float4 of the fp32
// side, so that every wave-wide load or store instruction covers one contiguous KiB; a 64-element block is then one DPP
// row of 16 lanes (block maximum by row rotations) and nothing goes through LDS.  Work is cut into fixed chunks, one per
// wave (wave_chunks), not into one long span per resident wave.
#include "common.h"

#include <stdlib.h>

// ------------------------------------------------------------------------------------------------
// quantize (rounding disabled): CloverVector4.h:605-807 with rnd_* == 0
//   algorithmic bytes: 4.5625 per element.
// ------------------------------------------------------------------------------------------------
// tail path (a chunk that is not a whole main-loop step): lane = 8 elements = one output dword from two float4, 8 lanes per block
__device__ __forceinline__ void quantize_word(const f32x4 a, const f32x4 b, uint64_t i, uint32_t *__restrict__ q, float *__restrict__ s)
{
    const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    float m = 0.0f;
#pragma unroll
    for (int e = 0; e < 8; e++) m = fmaxf(m, __builtin_fabsf(v[e]));
    m = fmaxf(m, __shfl_xor(m, 1));
    m = fmaxf(m, __shfl_xor(m, 2));
    m = fmaxf(m, __shfl_xor(m, 4));
    m = fix_zero_max(m);
    const float k = 7.0f / m;                 // IEEE-correct fp32 division (CloverVector4.h:668)
    __builtin_nontemporal_store(quant_pack8(v, k, nullptr), &q[i]);
    if ((i & 7) == 0) s[i >> 3] = m;
}

// Wave = one contiguous chunk (wave_chunks).  Main loop: lane -> float4, so each of the eight
// 16-byte load instructions of a step reads one contiguous KiB; a 64-element block is then exactly one DPP row of 16
// lanes (maximum by row rotations), a lane quantises half an output dword and lane pairs swap halves so that even
// lanes store the dwords of load j and odd lanes those of load j+1.  8 KiB per wave in flight.
#define VQ_UNROLL 4
__global__ __launch_bounds__(256) void k_v4_quantize(const f32x4 *__restrict__ x, uint32_t *__restrict__ q,
                                                     float *__restrict__ s, uint64_t nwords, uint64_t words_per_wave)
{
    const uint64_t wave = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const uint64_t w0 = wave * words_per_wave;
    const uint64_t w1 = (w0 + words_per_wave) < nwords ? (w0 + words_per_wave) : nwords;     // multiples of 64
    uint64_t w = w0;
    const int odd = lane & 1, sub = lane >> 1;
    for (; w + 64 * VQ_UNROLL <= w1; w += 64 * VQ_UNROLL) {
        f32x4 v[2 * VQ_UNROLL];
#pragma unroll
        for (int j = 0; j < 2 * VQ_UNROLL; j++) v[j] = __builtin_nontemporal_load(&x[2 * w + 64 * j + lane]);
        uint32_t half[2 * VQ_UNROLL];
#pragma unroll
        for (int j = 0; j < 2 * VQ_UNROLL; j++) {
            float m = fmaxf(fmaxf(__builtin_fabsf(v[j].x), __builtin_fabsf(v[j].y)), fmaxf(__builtin_fabsf(v[j].z), __builtin_fabsf(v[j].w)));
            m = fix_zero_max(row16_max(m));
            half[j] = quant_pack4(v[j], 7.0f / m);            // IEEE-correct fp32 division (CloverVector4.h:668)
            if ((lane & 15) == 0) s[(w >> 3) + 4 * j + (lane >> 4)] = m;
        }
#pragma unroll
        for (int j = 0; j < 2 * VQ_UNROLL; j += 2) {
            const uint32_t give = odd ? half[j] : half[j + 1];
            const uint32_t recv = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)give, 0xB1, 0xF, 0xF, false);    // quad_perm [1,0,3,2]
            const uint32_t word = odd ? (recv | (half[j + 1] << 16)) : (half[j] | (recv << 16));
            __builtin_nontemporal_store(word, &q[w + 32 * (j + odd) + sub]);
        }
    }
    for (; w < w1; w += 64) {
        const uint64_t i = w + lane;                 // nwords is a multiple of 16: whole 8-lane blocks are in or out
        if (i < w1) quantize_word(__builtin_nontemporal_load(&x[2 * i]), __builtin_nontemporal_load(&x[2 * i + 1]), i, q, s);
    }
}

// ------------------------------------------------------------------------------------------------
// restore: CloverVector4.h:1027-1093.  x = (scale / 7.0f) * q  (division first, then one multiply)
// ------------------------------------------------------------------------------------------------
// lane -> output float4: every store instruction of a wave writes one contiguous KiB (two lanes share an input dword,
// each converts the half it needs); 4 such stores per step
template <bool NT>
__global__ __launch_bounds__(256) void k_v4_restore(const uint32_t *__restrict__ q, const float *__restrict__ s,
                                                    f32x4 *__restrict__ x, uint64_t nwords, uint64_t words_per_wave)
{
    const uint64_t wave = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int half = lane & 1, sub = lane >> 1;
    const uint64_t w0 = wave * words_per_wave;
    const uint64_t w1 = (w0 + words_per_wave) < nwords ? (w0 + words_per_wave) : nwords;      // spans are multiples of 64 words
    for (uint64_t w = w0; w < w1; w += 128) {
        uint32_t wd[4];
        float sc[4];
#pragma unroll
        for (int h = 0; h < 4; h++) {
            const uint64_t i = w + 32 * h + sub;
            const uint64_t ic = i < w1 ? i : w0;
            wd[h] = q[ic];
            sc[h] = s[ic >> 3];
        }
#pragma unroll
        for (int h = 0; h < 4; h++) {
            const uint64_t i = w + 32 * h + sub;
            const uint32_t hw = wd[h] >> (16 * half);           // the 4 nibbles of this half: elements 4*half .. 4*half+3
            const float k = div7(sc[h]);
            f32x4 v;
            v.x = (float)(((int)(hw << 24)) >> 28) * k;         // element 0 of the half: high nibble of byte 0
            v.y = (float)(((int)(hw << 28)) >> 28) * k;
            v.z = (float)(((int)(hw << 16)) >> 28) * k;
            v.w = (float)(((int)(hw << 20)) >> 28) * k;
            if (i < w1) {
                if (NT) __builtin_nontemporal_store(v, &x[2 * i + half]);      // output larger than the Infinity Cache: stream it past the caches
                else x[2 * i + half] = v;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// exact integer word sums: I[w] = sum of the 8 nibble products of word w
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_v4_word_isums(const uint32_t *__restrict__ qu, const uint32_t *__restrict__ qv,
                                                       int32_t *__restrict__ I, uint64_t nwords)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nwords; i += stride)
        I[i] = sdot8(qu[i], qv[i], 0);
}

// ------------------------------------------------------------------------------------------------
// dot, reference order (CloverVector4.h:1095-1192): 16 sequential fp32 fma chains -- chain l = 8 a + w
// (accumulator a = block parity, AVX lane w = word of the block) -- then the fixed add tree of
// CloverBase.h:149-157.  The chains are sequential by definition (n/128 dependent fmas each), so the floor is
// one dependent v_fma_f32 (~2.3 ns) per block pair on ONE wave, and the job is to keep everything else out of
// that wave's instruction issue:
//   k_v4_dot_prep2  (all CUs) : both fma operands in chain order -- per block of 16 block pairs ("steps") and chain lane l:
//                               80 bytes = f of the 16 steps (f = (float)sdot8 of the chain's word) + ONE float4 of
//                               c = f32(f32(su/49)*sv) in quad layout; steps past the end hold zeros (fma(0, 0, acc) = acc);
//   k_v4_dot_chain2 (one wave): five 16-byte loads per lane bring 16 steps of both operands straight into registers (c is
//                               shared by the 8 lanes of an accumulator: each lane keeps a quarter of it and the fma reads
//                               its factor through DPP quad_perm, a broadcast inside the instruction); DOTX_D blocks stay in
//                               flight in a register ring, refilled in place behind the fmas that read them, with counted
//                               vmcnt waits.  No LDS, no barrier, no helper waves: round 1 staged the operands through LDS
//                               and the lone chain wave spent ~13 ns per ds_read_b128 (2 per 4 steps) -- 0.78 ms at
//                               n = 2^24 against a floor of 0.30.  The loop is one asm statement generated by
//                               tools/gen_dot_chain.py (layout and register map are documented there).
// ------------------------------------------------------------------------------------------------
#include "dot_chain.inc"

__global__ __launch_bounds__(256) void k_v4_dot_prep2(const uint32_t *__restrict__ qu, const float *__restrict__ su,
                                                      const uint32_t *__restrict__ qv, const float *__restrict__ sv,
                                                      uint64_t npairs, uint64_t nblocks_padded, f32x4 *__restrict__ X)
{
    // thread = (block Q of 16 steps, chain lane l, piece 0..4): pieces 0..3 = f of steps 4 piece .. 4 piece + 3, piece 4 = the c quad
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < nblocks_padded * 80; t += stride) {
        const uint64_t Q = t / 80;
        const int r = (int)(t - Q * 80), l = r / 5, piece = r - 5 * l, a = l >> 3, w = l & 7;
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint64_t p = piece < 4 ? 16 * Q + 4 * piece + i : 16 * Q + 4 * i + (l & 3);
            v[i] = 0.0f;
            if (p < npairs) {
                const uint64_t blk = 2 * p + a;
                v[i] = piece < 4 ? (float)sdot8(qu[8 * blk + w], qv[8 * blk + w], 0) : (su[blk] * CLV_RCP49) * sv[blk];
            }
        }
        X[t] = f32x4{v[0], v[1], v[2], v[3]};              // (Q * 16 + l) * 5 + piece = t: 80 bytes per lane, 1280 per block
    }
}

__global__ __launch_bounds__(64) void k_v4_dot_chain2(const f32x4 *__restrict__ X, uint32_t iterations, float *__restrict__ out)
{
    const int lane = threadIdx.x, j = lane & 15;
    const uint32_t voff = 80u * j;                    // lanes 16..63 shadow lanes 0..15 (same addresses, same arithmetic)
    const uint64_t x = (uint64_t)X;
    float acc;
    asm volatile(DOTX_LOOP_ASM : [acc] "=v"(acc) : [x] "s"(x), [voff] "v"(voff), [n] "s"(iterations) : DOTX_LOOP_CLOBBERS);
    // lanes 0..15 hold chain l: accumulator a = l>>3, AVX lane w = l&7
    const float v = acc + __shfl(acc, (j + 8) & 15);   // acc[0][w] + acc[1][w]          (:1190)
    const float xx = __shfl(v, (j + 4) & 15) + v;      // x[w] = v[w+4] + v[w], w = 0..3  (CloverBase.h:153)
    const float x2 = __shfl(xx, (j + 2) & 15);
    const float y = xx + x2;                           // y0 = x0 + x2 (lane 0), y1 = x1 + x3 (lane 1)
    const float y1 = __shfl(y, 1);
    if (lane == 0) *out = y + y1;
}

// ------------------------------------------------------------------------------------------------
// dot, fast order: exact integer block sums, per-block scale, fp32 partials reduced by a fixed tree.
//   lane = 16 B of each operand (half a block); lane pairs combine their integer sums first, so each
//   block contributes exactly c[b] * I[b] like the reference; only the fp32 summation ORDER differs.
//   algorithmic bytes: 1.125 per element.
// ------------------------------------------------------------------------------------------------
#include "dot_common.h"      // DOT_FAST_THREADS, block_sum_256, dot_hand_over_and_collect

__global__ __launch_bounds__(DOT_FAST_THREADS) void k_v4_dot_partial(const u32x4 *__restrict__ qu, const float *__restrict__ su,
                                                                     const u32x4 *__restrict__ qv, const float *__restrict__ sv,
                                                                     uint64_t nvec, float *__restrict__ partial)
{
    __shared__ float sh[4];
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    float acc = 0.0f;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += stride) {
        const u32x4 a = __builtin_nontemporal_load(&qu[i]);
        const u32x4 b = __builtin_nontemporal_load(&qv[i]);
        int I = dot32(a, b);
        I += __shfl_xor(I, 1);
        if ((i & 1) == 0) {
            const uint64_t blk = i >> 1;
            const float c = (su[blk] * CLV_RCP49) * sv[blk];
            acc = __builtin_fmaf(c, (float)I, acc);
        }
    }
    const float t = block_sum_256(acc, sh);
    if (threadIdx.x == 0) partial[blockIdx.x] = t;
}

__global__ __launch_bounds__(DOT_FAST_THREADS) void k_v4_dot_final(const float *__restrict__ partial, int count, float *__restrict__ out)
{
    __shared__ float sh[4];
    float acc = 0.0f;
    for (int i = threadIdx.x; i < count; i += DOT_FAST_THREADS) acc += partial[i];
    const float t = block_sum_256(acc, sh);
    if (threadIdx.x == 0) *out = t;
}

// The same dot in ONE launch (round 5).  At configs[1]'s size (n = 2^24: 18.9 MB, cache-resident) the two-kernel form is two fixed
// costs: k_v4_dot_partial 5.5 us + k_v4_dot_final 4.6 us (profiles/r04_dot_n2p24_kernel_stats.txt) for 2.4 us worth of bytes.
// Here every workgroup hands its partial over through an 8-byte slot {valid = 1, partial} -- one agent-scope store, partial and flag
// arrive together -- and the LAST workgroup of the grid, once its own share is done, collects the slots and runs k_v4_dot_final's
// tree: thread t adds slots t, t + 256, ... in that order, then block_sum_256 -- the same bits as the two-kernel form, whatever
// order the workgroups finish in.  No atomic read-modify-write anywhere: same-address device-scope atomics cost 42-62 ns EACH on this
// chip (DESIGN_HISTORY 5), a ticket per workgroup would be 50 us.  The collector clears every slot it has read, so the slots are zero
// again when the kernel ends (clv_internal_sync_slots' contract) and a captured graph replays correctly.  Forward progress: the grid
// never exceeds what is resident at once (dot_fast_grid: 4 workgroups of 256 per CU), so the workgroups the collector waits for are
// running or about to be scheduled whatever else shares the device.
// Loads: all U steps of a thread are requested before the first is used (two dependent round trips at n = 2^24 became one).
template <int U>
__global__ __launch_bounds__(DOT_FAST_THREADS) void k_v4_dot_fast1(const u32x4 *__restrict__ qu, const float *__restrict__ su,
                                                                   const u32x4 *__restrict__ qv, const float *__restrict__ sv,
                                                                   uint64_t nvec, unsigned long long *slots, float *__restrict__ out)
{
    __shared__ float sh[4];
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    float acc = 0.0f;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += U * stride) {
        u32x4 a[U], b[U];
        float cu[U], cv[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint64_t j = i + u * stride, jc = j < nvec ? j : i;         // nvec is even and i, stride have one parity: lane pairs stay together
            a[u] = __builtin_nontemporal_load(&qu[jc]);
            b[u] = __builtin_nontemporal_load(&qv[jc]);
            cu[u] = su[jc >> 1];
            cv[u] = sv[jc >> 1];
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            int I = dot32(a[u], b[u]);
            I += __shfl_xor(I, 1);
            const uint64_t j = i + u * stride;
            // branch-free (a branch lets hipcc sink the scale loads into it, behind the nibble loads' round trip): odd lanes and lanes past
            // the end multiply by 0 -- fma(0, I, acc) == acc exactly -- so every lane's sum is k_v4_dot_partial's, in the same order
            const float c = ((j & 1) == 0 && j < nvec) ? (cu[u] * CLV_RCP49) * cv[u] : 0.0f;
            acc = __builtin_fmaf(c, (float)I, acc);
        }
    }
    dot_hand_over_and_collect(block_sum_256(acc, sh), slots, out, sh);
}

// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
static inline int stream_grid(uint64_t items, int threads, int per_cu)
{
    const uint64_t want = (items + threads - 1) / threads;
    const uint64_t cap = (uint64_t)clv_cu_count() * per_cu;
    return (int)(want < cap ? (want ? want : 1) : cap);
}

int clv4_quantize_stochastic(const float *x, uint64_t n_pad, int8_t *q, float *s, uint64_t *rng_state_dev, hipStream_t st);

// One fixed chunk per wave, a grid of them: the workgroups resident at any moment cover one moving window of the vector.
// (Round 1 started with one long contiguous span per resident wave -- 8192 independent sequential streams -- which was
// 12-25 % slower at n = 2^30: quantize 0.97 -> 0.82 ms, restore 1.12 -> 0.84 ms, same kernels.)
static inline void wave_chunks(uint64_t nwords, uint64_t chunk_words, uint64_t *words_per_wave, uint64_t *waves)
{
    *words_per_wave = chunk_words;
    *waves = (nwords + chunk_words - 1) / chunk_words;
}

extern "C" int clv4_quantize(const float *x, uint64_t n_pad, int8_t *q, float *s, uint64_t *rng_state_dev, void *stream)
{
    CLV_REQUIRE(x && q && s, "clv4_quantize: null pointer");
    CLV_REQUIRE(n_pad % 128 == 0, "clv4_quantize: n_pad=%llu is not a multiple of 128", (unsigned long long)n_pad);
    if (!n_pad) return CLV_OK;
    if (rng_state_dev) return clv4_quantize_stochastic(x, n_pad, q, s, rng_state_dev, as_stream(stream));
    const uint64_t nwords = n_pad / 8;                                   // multiple of 16
    uint64_t words_per_wave, waves;
    wave_chunks(nwords, 64 * VQ_UNROLL, &words_per_wave, &waves);       // one main-loop step (8 KiB in) per wave
    hipLaunchKernelGGL(k_v4_quantize, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, as_stream(stream),
                       (const f32x4 *)x, (uint32_t *)q, s, nwords, words_per_wave);
    CLV_LAUNCH_CHECK();
    return CLV_OK;
}

extern "C" int clv4_restore(const int8_t *q, const float *s, uint64_t n_pad, float *x, void *stream)
{
    CLV_REQUIRE(x && q && s, "clv4_restore: null pointer");
    CLV_REQUIRE(n_pad % 128 == 0, "clv4_restore: n_pad=%llu is not a multiple of 128", (unsigned long long)n_pad);
    if (!n_pad) return CLV_OK;
    const uint64_t nwords = n_pad / 8;
    uint64_t words_per_wave, waves;
    wave_chunks(nwords, 128, &words_per_wave, &waves);                  // one step (4 KiB out) per wave
    if (n_pad * sizeof(float) > (256ull << 20))
        hipLaunchKernelGGL(k_v4_restore<true>, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, as_stream(stream),
                           (const uint32_t *)q, s, (f32x4 *)x, nwords, words_per_wave);
    else
        hipLaunchKernelGGL(k_v4_restore<false>, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, as_stream(stream),
                           (const uint32_t *)q, s, (f32x4 *)x, nwords, words_per_wave);
    CLV_LAUNCH_CHECK();
    return CLV_OK;
}

extern "C" int clv4_word_isums(const int8_t *qu, const int8_t *qv, uint64_t n_pad, int32_t *isums, void *stream)
{
    CLV_REQUIRE(qu && qv && isums, "clv4_word_isums: null pointer");
    CLV_REQUIRE(n_pad % 128 == 0, "clv4_word_isums: n_pad=%llu is not a multiple of 128", (unsigned long long)n_pad);
    if (!n_pad) return CLV_OK;
    const uint64_t nwords = n_pad / 8;
    hipLaunchKernelGGL(k_v4_word_isums, dim3(stream_grid(nwords, 256, 8)), dim3(256), 0, as_stream(stream),
                       (const uint32_t *)qu, (const uint32_t *)qv, isums, nwords);
    CLV_LAUNCH_CHECK();
    return CLV_OK;
}

static inline int dot_fast_grid(uint64_t n_pad)
{
    return stream_grid(n_pad / 32, DOT_FAST_THREADS, 4);
}

static inline uint64_t dot_exact_blocks(uint64_t n_pad) { return (n_pad / 128 + 15) / 16; }        // 16 block pairs each
// rounded up to whole loop iterations of the chain kernel (DOTX_D blocks each)
static inline uint64_t dot_exact_blocks_padded(uint64_t n_pad)
{
    return (dot_exact_blocks(n_pad) + DOTX_D - 1) / DOTX_D * DOTX_D;
}

// The chain kernel for other callers in the library (CloverVector8::dot, mixed8.hip): X holds `steps` fma steps in k_v4_dot_prep2's layout
// (per block of 16 steps and chain lane l: 80 bytes = 16 f + one float4 of c in quad layout; zeros past the end)
uint64_t clv_internal_dot_chain_blocks_padded(uint64_t steps)
{
    const uint64_t blocks = (steps + 15) / 16;
    return (blocks + DOTX_D - 1) / DOTX_D * DOTX_D;
}
uint64_t clv_internal_dot_chain_bytes(uint64_t steps) { return (clv_internal_dot_chain_blocks_padded(steps) + DOTX_D) * 1280 + 256; }
int clv_internal_dot_chain(const void *X, uint64_t blocks_padded, float *out_dev, hipStream_t st)
{
    CLV_REQUIRE(blocks_padded / DOTX_D <= 0xFFFFFFFFull, "dot chain: vector too long");
    hipLaunchKernelGGL(k_v4_dot_chain2, dim3(1), dim3(64), 0, st, (const f32x4 *)X, (uint32_t)(blocks_padded / DOTX_D), out_dev);
    CLV_LAUNCH_CHECK();
    return CLV_OK;
}

extern "C" uint64_t clv4_dot_workspace_bytes(uint64_t n_pad)
{
    const uint64_t fast = (uint64_t)clv_cu_count() * 4 * sizeof(float) + 256;
    // + DOTX_D blocks: the last iteration's refills read one iteration past the end (never consumed, but the memory must exist)
    const uint64_t exact = (dot_exact_blocks_padded(n_pad) + DOTX_D) * 1280 + 256;
    return fast > exact ? fast : exact;
}

extern "C" int clv4_dot(const int8_t *qu, const float *su, const int8_t *qv, const float *sv, uint64_t n_pad,
                        int mode, float *out_dev, void *workspace, void *stream)
{
    CLV_REQUIRE(qu && su && qv && sv && out_dev, "clv4_dot: null pointer");
    CLV_REQUIRE(n_pad % 128 == 0, "clv4_dot: n_pad=%llu is not a multiple of 128", (unsigned long long)n_pad);
    CLV_REQUIRE(mode == CLV_DOT_EXACT || mode == CLV_DOT_FAST, "clv4_dot: unknown mode %d", mode);
    hipStream_t st = as_stream(stream);
    if (!n_pad) { CLV_HIP(hipMemsetAsync(out_dev, 0, sizeof(float), st)); return CLV_OK; }
    const int grid = dot_fast_grid(n_pad);
    static const bool two_launches = getenv("CLV_DOT_FAST_TWO_LAUNCHES") != nullptr;      // A/B switch: the round-1..4 form (same bits)
    const bool needs_scratch = mode == CLV_DOT_EXACT || two_launches || grid > DOT_FAST_THREADS * DOT_MAX_SLOTS_PER_THREAD;
    if (!workspace && needs_scratch) {
        int rc = clv_internal_workspace(&workspace, clv4_dot_workspace_bytes(n_pad), as_stream(stream));
        if (rc) return rc;
    }
    if (mode == CLV_DOT_EXACT) {
        const uint64_t npairs = n_pad / 128, gpad = dot_exact_blocks_padded(n_pad);
        CLV_REQUIRE(gpad / DOTX_D <= 0xFFFFFFFFull, "clv4_dot: vector too long");
        f32x4 *X = (f32x4 *)workspace;
        hipLaunchKernelGGL(k_v4_dot_prep2, dim3(stream_grid(gpad * 80, 256, 8)), dim3(256), 0, st, (const uint32_t *)qu, su,
                           (const uint32_t *)qv, sv, npairs, gpad, X);
        CLV_LAUNCH_CHECK();
        hipLaunchKernelGGL(k_v4_dot_chain2, dim3(1), dim3(64), 0, st, (const f32x4 *)X, (uint32_t)(gpad / DOTX_D), out_dev);
        CLV_LAUNCH_CHECK();
        return CLV_OK;
    }
    if (two_launches || grid > DOT_FAST_THREADS * DOT_MAX_SLOTS_PER_THREAD) {
        hipLaunchKernelGGL(k_v4_dot_partial, dim3(grid), dim3(DOT_FAST_THREADS), 0, st, (const u32x4 *)qu, su, (const u32x4 *)qv, sv,
                           n_pad / 32, (float *)workspace);
        CLV_LAUNCH_CHECK();
        hipLaunchKernelGGL(k_v4_dot_final, dim3(1), dim3(DOT_FAST_THREADS), 0, st, (const float *)workspace, grid, out_dev);
        CLV_LAUNCH_CHECK();
        return CLV_OK;
    }
    void *slots = nullptr;
    int rc = clv_internal_sync_slots(&slots, (uint64_t)grid * 8, st);
    if (rc) return rc;
    const uint64_t nvec = n_pad / 32, per_thread = (nvec + (uint64_t)grid * DOT_FAST_THREADS - 1) / ((uint64_t)grid * DOT_FAST_THREADS);
    static const int force_u = [] { const char *e = getenv("CLV_DOT_FAST_U"); return e ? atoi(e) : 0; }();      // A/B switch
    // loads in flight per thread, measured on one box (profiles/r05_dot_fast_ab.jsonl, us at n = 2^24 / 2^26 / 2^29 / 2^30): U = 1: 4.19 / 11.6 /
    // 88.7 / 174.0, U = 2: 4.02 / 12.2 / 92.5 / 189.3, U = 4: 5.05 / 12.3 / 120.0 / 217.1 (two launches: 6.38 / 14.7 / 91.0 / 181.7) -- deeper
    // only pays where a thread has two steps in all (one round trip instead of two); long vectors want the plain loop
    const int u = force_u ? force_u : per_thread <= 2 ? 2 : 1;
#define DOT1_LAUNCH(U)                                                                                                                      \
    hipLaunchKernelGGL(k_v4_dot_fast1<U>, dim3(grid), dim3(DOT_FAST_THREADS), 0, st, (const u32x4 *)qu, su, (const u32x4 *)qv, sv, nvec, \
                       (unsigned long long *)slots, out_dev)
    if (u >= 4) DOT1_LAUNCH(4);
    else if (u == 2) DOT1_LAUNCH(2);
    else DOT1_LAUNCH(1);
#undef DOT1_LAUNCH
    CLV_LAUNCH_CHECK();
    return CLV_OK;
}
