// threshold4.hip -- CloverVector4 / CloverVector8 ::threshold(K): the radix select (FAST: one-workgroup and large-vector
// forms) and the reference's min-heap walk (REFERENCE).
// One of the callers either side of the hot path.  With mvm these are the five steps of the reference's quantized IHT / GD
// iterations (test/performance/01_measure.h:923-946, 999-1021), so x, t1..t3 can stay in HBM across iterations.
#include "common.h"
#include "thresh_device.h"

#include <type_traits>

// =================================================================================================
// f3  CloverVector4::threshold(K) (CloverVector4.h:1913-1975): keep the K largest |value| among the first n
//     elements, zero the other nibbles, scales untouched.  The reference walks a K-entry min-heap
//     sequentially (O(n log K)); here: exact radix select on the fp32 bit pattern of
//     |value| = |f32(s/7) * q| (12 + 12 + 8 bits, three histogram passes over 0.56 B/element), then
//     one pass that keeps everything above the K-th value and the lowest-index ties.
//     The kept multiset of magnitudes is identical to the reference's; WHICH of several equal magnitudes
//     survive is heap-order dependent there and lowest-index-first here.
// =================================================================================================
struct ThreshState {
    uint32_t prefix;      // selected high bits so far
    uint32_t remaining;   // how many elements still to take inside the selected bin
    uint32_t tau;         // final: bit pattern of the K-th largest magnitude
    uint32_t ties_keep;   // final: how many elements == tau survive
};

__device__ __forceinline__ uint32_t mag_key(uint32_t w, int e, float s7)
{
    return __float_as_uint(__builtin_fabsf(s7 * (float)unpack1(w, e)));
}

// The large-n kernels below serve both element widths.  BITS = 4: CloverVector4, 8 elements per word, |value| = |f32(s/7) * q|
// (CloverVector4::get, :206-209).  BITS = 8: CloverVector8 (same algorithm, CloverVector8.h:1680-1740), 4 elements per word,
// |value| = |f32((float)q * s) / 127| (CloverVector8::get, :137-140).
template <int BITS>
struct ThreshElems {
    static constexpr int EPW = 32 / BITS;                                   // elements per 32-bit word
    static constexpr int WPB = 64 / EPW;                                    // words per 64-element block
    __device__ static __forceinline__ uint32_t mask(int e) { return BITS == 4 ? 0xFu << nib_shift(e) : 0xFFu << (8 * e); }
    __device__ static __forceinline__ uint32_t key(uint32_t w, int e, float sc)
    {
        if (BITS == 4) return mag_key(w, e, div7(sc));
        const float q = (float)((int)(w << (24 - 8 * e)) >> 24);
        return __float_as_uint(__builtin_fabsf(div127(q * sc)));
    }
};

// level 0: bins = key >> 20 (4096); level 1: key>>20 == prefix, bins = (key >> 8) & 0xFFF; level 2:
// key>>8 == prefix, bins = key & 0xFF
template <int LEVEL, int BITS>
__global__ __launch_bounds__(256) void k_thresh_hist(const uint32_t *__restrict__ q, const float *__restrict__ s, uint64_t n,
                                                     const ThreshState *__restrict__ ts, uint32_t *__restrict__ hist)
{
    typedef ThreshElems<BITS> E;
    __shared__ uint32_t lh[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) lh[i] = 0;
    __syncthreads();
    const uint32_t prefix = LEVEL ? ts->prefix : 0;
    const uint64_t nwords = (n + E::EPW - 1) / E::EPW;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nwords; i += stride) {
        const uint32_t w = q[i];
        const float sc = s[i / E::WPB];
#pragma unroll
        for (int e = 0; e < E::EPW; e++) {
            if (i * E::EPW + e >= n) break;
            const uint32_t key = E::key(w, e, sc);
            if (LEVEL == 0) atomicAdd(&lh[key >> 20], 1u);
            else if (LEVEL == 1) { if ((key >> 20) == prefix) atomicAdd(&lh[(key >> 8) & 0xFFF], 1u); }
            else { if ((key >> 8) == prefix) atomicAdd(&lh[key & 0xFF], 1u); }
        }
    }
    __syncthreads();
    const int nb = LEVEL == 2 ? 256 : 4096;
    for (int i = threadIdx.x; i < nb; i += 256) if (lh[i]) atomicAdd(&hist[i], lh[i]);
}

// one WG: find, from the top, the bin in which the cumulative count reaches `remaining`; leaves the histogram zeroed
template <int LEVEL>
__global__ __launch_bounds__(256) void k_thresh_select(uint32_t *__restrict__ hist, ThreshState *__restrict__ ts, uint32_t k)
{
    __shared__ uint32_t wsum[4];
    constexpr int nb = LEVEL == 2 ? 256 : 4096;
    constexpr int per = nb / 256;
    const int t = threadIdx.x;
    const int top = (255 - t) * per + per - 1;                   // thread t owns the t-th run of `per` bins from the top
    uint32_t bins[per], sum = 0;
#pragma unroll
    for (int i = 0; i < per; i++) { bins[i] = hist[top - i]; sum += bins[i]; }
    uint32_t v = wave_scan_incl(sum);
    if ((t & 63) == 63) wsum[t >> 6] = v;
    __syncthreads();
    for (int w = 0; w < (t >> 6); w++) v += wsum[w];
    const uint32_t need = LEVEL == 0 ? k : ts->remaining;
    if (v >= need && v - sum < need) {
        uint32_t above = v - sum;
        int i = 0;
#pragma unroll
        for (int j = 0; j < per - 1; j++)
            if (i == j && above + bins[j] < need) { above += bins[j]; i = j + 1; }
        const uint32_t bin = (uint32_t)(top - i);
        const uint32_t prev = LEVEL == 0 ? 0 : ts->prefix;
        const uint32_t prefix = LEVEL == 0 ? bin : (LEVEL == 1 ? (prev << 12) | bin : (prev << 8) | bin);
        ts->prefix = prefix;
        ts->remaining = need - above;
        if (LEVEL == 2) { ts->tau = prefix; ts->ties_keep = need - above; }
    }
#pragma unroll
    for (int i = 0; i < per; i++) hist[top - i] = 0;             // ready for the next level
}

// chunked (not grid-stride) so that index order = (block, thread, element)
#define TH_WORDS_PER_BLOCK 2048      // 8 words per thread

template <int BITS>
__global__ __launch_bounds__(256) void k_thresh_count_ties(const uint32_t *__restrict__ q, const float *__restrict__ s, uint64_t n,
                                                           const ThreshState *__restrict__ ts, uint32_t *__restrict__ block_ties)
{
    typedef ThreshElems<BITS> E;
    __shared__ uint32_t cnt;
    if (threadIdx.x == 0) cnt = 0;
    __syncthreads();
    const uint32_t tau = ts->tau;
    const uint64_t w0 = (uint64_t)blockIdx.x * TH_WORDS_PER_BLOCK + threadIdx.x * 8;
    uint32_t c = 0;
    for (int k = 0; k < 8; k++) {
        const uint64_t i = w0 + k;
        if (i * E::EPW >= n) break;
        const uint32_t w = q[i];
        const float sc = s[i / E::WPB];
#pragma unroll
        for (int e = 0; e < E::EPW; e++) if (i * E::EPW + e < n && E::key(w, e, sc) == tau) c++;
    }
    if (c) atomicAdd(&cnt, c);
    __syncthreads();
    if (threadIdx.x == 0) block_ties[blockIdx.x] = cnt;
}

__global__ __launch_bounds__(256) void k_thresh_scan(uint32_t *__restrict__ block_ties, uint32_t nblocks)
{
    // exclusive scan by one WG (nblocks is n / 16384: small)
    __shared__ uint32_t wsum[4];
    const int t = threadIdx.x;
    uint32_t carry = 0;
    for (uint32_t base = 0; base < nblocks; base += 256) {
        const uint32_t i = base + t;
        const uint32_t v = i < nblocks ? block_ties[i] : 0;
        uint32_t incl = wave_scan_incl(v);
        __syncthreads();                                         // wsum of the previous round has been read
        if ((t & 63) == 63) wsum[t >> 6] = incl;
        __syncthreads();
        for (int w = 0; w < (t >> 6); w++) incl += wsum[w];
        if (i < nblocks) block_ties[i] = carry + incl - v;
        carry += wsum[0] + wsum[1] + wsum[2] + wsum[3];
    }
}

template <int BITS>
__global__ __launch_bounds__(256) void k_thresh_apply(uint32_t *__restrict__ q, const float *__restrict__ s, uint64_t n,
                                                      const ThreshState *__restrict__ ts, const uint32_t *__restrict__ block_ties)
{
    typedef ThreshElems<BITS> E;
    __shared__ uint32_t tcnt[256];
    const uint32_t tau = ts->tau, keep = ts->ties_keep;
    const uint64_t w0 = (uint64_t)blockIdx.x * TH_WORDS_PER_BLOCK + threadIdx.x * 8;
    uint32_t words[8];
    uint32_t tie_mask[8];
    uint32_t c = 0;
    for (int k = 0; k < 8; k++) {
        const uint64_t i = w0 + k;
        words[k] = 0;
        tie_mask[k] = 0;
        if (i * E::EPW >= n) continue;
        const uint32_t w = q[i];
        const float sc = s[i / E::WPB];
        uint32_t outw = 0;
#pragma unroll
        for (int e = 0; e < E::EPW; e++) {
            if (i * E::EPW + e >= n) continue;
            const uint32_t key = E::key(w, e, sc);
            if (key > tau) outw |= w & E::mask(e);
            else if (key == tau) { tie_mask[k] |= 1u << e; c++; }
        }
        words[k] = outw;
    }
    tcnt[threadIdx.x] = c;
    __syncthreads();
    // exclusive prefix of tie counts inside the block (Hillis-Steele over 256 threads)
    uint32_t incl = c;
    for (int o = 1; o < 256; o <<= 1) {
        const uint32_t add = threadIdx.x >= (unsigned)o ? tcnt[threadIdx.x - o] : 0;
        __syncthreads();
        incl += add;
        tcnt[threadIdx.x] = incl;
        __syncthreads();
    }
    uint32_t rank = block_ties[blockIdx.x] + incl - c;
    for (int k = 0; k < 8; k++) {
        const uint64_t i = w0 + k;
        if (i * E::EPW >= n) break;
        uint32_t outw = words[k];
        const uint32_t w = q[i];
#pragma unroll
        for (int e = 0; e < E::EPW; e++)
            if (tie_mask[k] & (1u << e)) {
                if (rank < keep) outw |= w & E::mask(e);
                rank++;
            }
        // elements at or beyond n (the padding) are left as they are, like the reference's loop to `length`
        const uint64_t first = i * E::EPW;
        if (first + E::EPW > n) {
            for (int e = 0; e < E::EPW; e++) if (first + e >= n) outw |= w & E::mask(e);
        }
        q[i] = outw;
    }
}

// ---- single-workgroup path (n_pad <= 131072: every IHT size the reference benchmarks): one launch, the vector,
// the histograms and the scans all live in LDS.  One workgroup on one CU is latency-bound code, so this kernel is
// written to execute few instructions and few barriers:
//  * A block holds at most 9 distinct magnitudes (|nibble| in 0..8 times one scale), so the selection runs over
//    9 * n/64 weighted CANDIDATES (block, |nibble|), weight = how many elements of the block carry that magnitude,
//    instead of over the n elements.
//  * Radix select with four 8-bit levels: a 256-bin histogram is scanned by four waves in one step, where the
//    4096-bin levels used before cost 16 serial, bank-conflicting LDS reads per thread and level.
//  * Words are handled whole (SWAR over the 8 nibbles): magnitudes, per-block cut-offs and tie masks are bit
//    operations on the 32-bit word; there is no per-element float work after the selection.
#define TS_THREADS 1024
#define TS_MAXW 16

// Bit-sliced magnitude counts of a FULL block (64 nibbles in 8 words).  A 4 x 4 bit transpose inside every nibble column turns four words
// into the four bit planes of their 32 nibbles (plane j, bit 4e + k = bit j of nibble e of word k); |v| is then taken on the planes
// (two's complement: a1 = v1 ^ (sign & v0), a2 = v2 ^ (sign & (v1 | v0)), a3 = sign & ~(v2 | v1 | v0), i.e. only for -8) and every
// magnitude 1..7 is one three-input boolean + one popcount per half block: ~1.5 VALU per element against ~5 for the nibble-by-nibble
// walk (round 3: the count pass ran at 2.7 TB/s, VALU-bound).  Element order inside the block does not matter for counts.
__device__ __forceinline__ void th4_planes(uint32_t x0, uint32_t x1, uint32_t x2, uint32_t x3, uint32_t P[4])
{
    uint32_t t;
    t = ((x0 >> 1) ^ x1) & 0x55555555u; x1 ^= t; x0 ^= t << 1;
    t = ((x2 >> 1) ^ x3) & 0x55555555u; x3 ^= t; x2 ^= t << 1;
    t = ((x0 >> 2) ^ x2) & 0x33333333u; x2 ^= t; x0 ^= t << 2;
    t = ((x1 >> 2) ^ x3) & 0x33333333u; x3 ^= t; x1 ^= t << 2;
    P[0] = x0; P[1] = x1; P[2] = x2; P[3] = x3;
}

__device__ __forceinline__ unsigned long long th4_count_full_block(const uint32_t w[8])
{
    uint32_t A[2][4];
#pragma unroll
    for (int h = 0; h < 2; h++) {
        uint32_t v[4];
        th4_planes(w[4 * h], w[4 * h + 1], w[4 * h + 2], w[4 * h + 3], v);
        const uint32_t low = v[1] | v[0];
        A[h][0] = v[0];
        A[h][1] = v[1] ^ (v[3] & v[0]);
        A[h][2] = v[2] ^ (v[3] & low);
        A[h][3] = v[3] & ~(v[2] | low);
    }
    uint32_t c[9], sum = 0;
#pragma unroll
    for (int m = 1; m <= 7; m++) {
        uint32_t n = 0;
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const uint32_t b2 = (m & 4) ? A[h][2] : ~A[h][2], b1 = (m & 2) ? A[h][1] : ~A[h][1], b0 = (m & 1) ? A[h][0] : ~A[h][0];
            n += __popc(b2 & b1 & b0);
        }
        c[m] = n;
        sum += n;
    }
    c[8] = __popc(A[0][3]) + __popc(A[1][3]);
    c[0] = 64u - sum - c[8];
    // fields of 7 bits at bit 7 m: m = 0..3 in the low word, m = 4 straddles bit 32
    const uint32_t lo = c[0] | (c[1] << 7) | (c[2] << 14) | (c[3] << 21) | (c[4] << 28);
    const uint32_t hi = (c[4] >> 4) | (c[5] << 3) | (c[6] << 10) | (c[7] << 17) | (c[8] << 24);
    return ((unsigned long long)hi << 32) | lo;
}

// Round 4: the same selection with a third of the synchronisation (measured with cycle stamps at N = 8192, round-3 kernel: 16.5 k cycles
// = load 1.6 k, counts 1.8 k, four levels 7.2 k, cut-offs 0.8 k, ties + scan 4.0 k, apply 0.6 k; about 20 barriers):
//  * W = words per thread is a template parameter (1 at N = 8192): the thread's words live in registers, nothing is staged in LDS, no
//    16-slot loop with dead slots;
//  * a thread computes the (key, weight) of its <= 9 W / 8 + 1 candidates ONCE, in registers, before the levels;
//  * a level is: the candidates' LDS atomics, ONE barrier, and then EVERY wave scans the 256 bins itself (4 bins per lane, one DPP scan,
//    one ballot) -- no second and third barrier to publish the selected bin;
//  * the per-block cut-offs are computed by each thread for its own words' blocks (9 products): no table, no barrier;
//  * the tie ranks need one barrier (16 wave totals), not two.
// Same keys, same selection, same lowest-index tie rule: results are bit-identical (all threshold and IHT tests).
#ifndef TS4_COPIES
#define TS4_COPIES 4           // 1 / 2 / 4 / 8 copies: n = 131072 33.9 / 30.4 / 28.7 / 29.1 us, n = 32768 10.6 / 9.8 / 9.5 / 10.3, n = 8192 6.1 / 6.4 / 6.0 / 6.5 (r6, same box)
#endif
#define TS4_CS (TS4_COPIES == 1 ? 256 : 260)
template <int W>
__global__ __launch_bounds__(TS_THREADS) void k_thresh_small(uint32_t *__restrict__ q, const float *__restrict__ s, uint32_t n, uint32_t k)
{
    constexpr int MAXB = TS_THREADS * TS_MAXW / 8;                       // 2048 blocks at most
    constexpr int NC = (9 * W + 7) / 8 + 1;                              // candidates per thread at most
    // TS4_COPIES copies of a level's bins, TS4_CS words apart (a bin's copies on different banks), chosen by the lane: the candidates of
    // a vector crowd into a few bins, and LDS atomics on one address serialise
    __shared__ __attribute__((aligned(16))) uint32_t hist[4 * TS4_COPIES * TS4_CS];
    __shared__ unsigned long long cnt[MAXB];                             // per block: 9 fields of 7 bits, field m = #(|nibble| == m)
    __shared__ float s7[MAXB];
    __shared__ uint32_t wtot[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t nwords = (n + 7) / 8, nblocks = (n + 63) / 64;
    const uint32_t w0 = tid * W;
    uint32_t w[W];
#pragma unroll
    for (int j = 0; j < W; j++) w[j] = q[w0 + j < nwords ? w0 + j : 0];
    for (uint32_t i = tid; i < nblocks; i += TS_THREADS) { s7[i] = div7(s[i]); if (W < 8) cnt[i] = 0ull; }
    for (int i = tid; i < 4 * TS4_COPIES * TS4_CS; i += TS_THREADS) hist[i] = 0;
    __syncthreads();                                                     // tables zero before anybody adds to them; s7 visible (also for k = 0)

    uint32_t tau = 0x7F800000u, keep = 0;
    if (k != 0) {
        // magnitude tables.  W >= 8: a thread holds whole blocks (plain stores, bit-sliced counts for full blocks); else LDS atomics
        if constexpr (W >= 8) {
#pragma unroll
            for (int g = 0; g < W / 8; g++) {
                const uint32_t b = (w0 >> 3) + g;
                if (b < nblocks) {
                    unsigned long long acc = 0;
                    if (64u * b + 64u <= n) {
                        acc = th4_count_full_block(&w[8 * g]);
                    } else {
#pragma unroll
                        for (int j = 0; j < 8; j++) {
                            const uint32_t first = 64u * b + 8u * j;
                            acc += th4_count_word(w[8 * g + j], first >= n ? 0u : (n - first < 8 ? n - first : 8u));
                        }
                    }
                    cnt[b] = acc;
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < W; j++) {
                const uint32_t i = w0 + j;
                if (i < nwords) atomicAdd(&cnt[i >> 3], th4_count_word(w[j], n - 8 * i < 8 ? n - 8 * i : 8u));
            }
        }
        __syncthreads();
        // this thread's candidates (block, magnitude): key and weight, once
        uint32_t ckey[NC], cwgt[NC];
#pragma unroll
        for (int c = 0; c < NC; c++) {
            const uint32_t ci = tid + TS_THREADS * c;
            const bool in = ci < 9 * nblocks;
            const uint32_t b = in ? ci / 9u : 0u, m = ci - 9u * b;
            cwgt[c] = in ? (uint32_t)(cnt[b] >> (7 * m)) & 0x7Fu : 0u;
            ckey[c] = cand_key(s7[b], (int)m);
        }
        uint32_t prefix = 0, need = k;
#pragma unroll
        for (int level = 0; level < 4; level++) {
            const int shift = 24 - 8 * level;
            uint32_t *h = hist + TS4_COPIES * TS4_CS * level, *hc = h + TS4_CS * (lane & (TS4_COPIES - 1));
#pragma unroll
            for (int c = 0; c < NC; c++)
                if (cwgt[c] && (level == 0 || (ckey[c] >> (shift + 8)) == prefix)) atomicAdd(&hc[(ckey[c] >> shift) & 0xFFu], cwgt[c]);
            __syncthreads();
            // every wave selects for itself: lane l owns bins 255 - 4 l ... 252 - 4 l (from the top), added up over the copies
            uint32_t t0 = 0, t1 = 0, t2 = 0, t3 = 0;
#pragma unroll
            for (int cpy = 0; cpy < TS4_COPIES; cpy++) {
                const u32x4 h4 = *reinterpret_cast<const u32x4 *>(h + TS4_CS * cpy + 252 - 4 * lane);
                t0 += h4.w; t1 += h4.z; t2 += h4.y; t3 += h4.x;
            }
            const uint32_t sum = t0 + t1 + t2 + t3;
            const uint32_t incl = wave_scan_incl(sum);
            const unsigned long long hit = __ballot(incl >= need && incl - sum < need);
            const int L = __builtin_ctzll(hit);                          // exactly one lane: the level's total weight is >= need
            uint32_t above = __shfl(incl - sum, L);
            const uint32_t T0 = __shfl(t0, L), T1 = __shfl(t1, L), T2 = __shfl(t2, L);
            uint32_t pick = 0;
            if (above + T0 < need) { above += T0; pick = 1;
                if (above + T1 < need) { above += T1; pick = 2;
                    if (above + T2 < need) { above += T2; pick = 3; } } }
            prefix = (prefix << 8) | (255u - 4u * (uint32_t)L - pick);
            need -= above;
        }
        tau = prefix;
        keep = need;
    }
    // survivors above tau, ties, the rank of this thread's first tie (ties in index order: the first `keep` survive)
    uint32_t keepbits[W], tiebits[W], c = 0;
    uint32_t lo_t = 0, hi_t = 0, cur_b = 0xFFFFFFFFu;
#pragma unroll
    for (int j = 0; j < W; j++) {
        const uint32_t i = w0 + j;
        keepbits[j] = tiebits[j] = 0;
        if (i < nwords) {
            if ((i >> 3) != cur_b) {                                     // per block: magnitudes >= hi_t are above tau, [lo_t, hi_t) equal it
                cur_b = i >> 3;
                const float sc = s7[cur_b];
                lo_t = hi_t = 0;
#pragma unroll
                for (int m = 0; m <= 8; m++) {
                    const uint32_t key = cand_key(sc, m);
                    lo_t += key < tau;
                    hi_t += key <= tau;
                }
            }
            const uint32_t ab = abs_nibbles(swap_nibbles(w[j]));
            const uint32_t valid = first_nibbles(n - 8 * i < 8 ? n - 8 * i : 8);
            const uint32_t above = ge_nibbles(ab, hi_t);
            keepbits[j] = above | (0x88888888u & ~valid);                 // padding is left alone
            tiebits[j] = ge_nibbles(ab, lo_t) & ~above & valid;
            c += __popc(tiebits[j]);
        }
    }
    const uint32_t v = wave_scan_incl(c);
    if (lane == 63) wtot[wave] = v;
    __syncthreads();
    const uint32_t tot = lane < 16 ? wtot[lane] : 0;
    const uint32_t inc = wave_scan_incl(tot);
    uint32_t rank = v - c + __shfl(inc - tot, wave);
#pragma unroll
    for (int j = 0; j < W; j++) {
        const uint32_t i = w0 + j;
        if (i < nwords) {
            uint32_t t = tiebits[j], kb = keepbits[j];
            const uint32_t nt = __popc(t), room = keep > rank ? keep - rank : 0;
            if (room >= nt) kb |= t;
            else for (uint32_t r = 0; r < room; r++) { kb |= t & (0u - t); t &= t - 1; }
            rank += nt;
            const uint32_t full = (kb >> 3) * 0xFu;                       // bit 3 -> whole nibble, then back to the stored nibble order
            q[i] = w[j] & swap_nibbles(full);
        }
    }
}

// ---- single-workgroup path for CloverVector8 (n_pad <= 32768): element keys live in registers (8 words = 32 elements per
// thread), radix select in four 8-bit levels straight over the elements (a block has up to 128 distinct magnitudes, so the
// candidate trick of the 4-bit kernel does not pay), same DPP scans, same lowest-index tie rule.
#define TS8_MAXW 8
template <int TS8_W>          // words per thread, compile-time so that the unrolled loops carry no dead slots
__global__ __launch_bounds__(TS_THREADS) void k_thresh8_small(uint32_t *__restrict__ q, const float *__restrict__ s, uint32_t n, uint32_t k)
{
    typedef ThreshElems<8> E;
    constexpr int COPIES = 8;                      // private histograms by lane & 7: the keys of a vector crowd into a few bins
    constexpr int CS = 260;                        // words between two copies: the same bin of two copies must not share an LDS bank (r6)
    __shared__ __attribute__((aligned(16))) uint32_t hist[4 * COPIES * CS];
    __shared__ __attribute__((aligned(16))) uint32_t hsum[2 * 256];    // per-level sums over the copies (two buffers: a fast wave may already
    __shared__ uint32_t wsum[16];                                       // write the next level's while a slow one still reads this level's)
    const int tid = threadIdx.x, lane = tid & 63;
    const uint32_t nwords = (n + 3) / 4;
    constexpr uint32_t W = TS8_W;                                     // contiguous words per thread: index order = thread order
    const uint32_t w0 = tid * W;
    uint32_t words[TS8_W], keys[TS8_W][4];
    uint32_t valid = 0;                                                // bit 4j+e: element e of word j exists (index < n)
#pragma unroll
    for (uint32_t j = 0; j < TS8_W; j++) {
        const uint32_t i = w0 + j;
        const bool in = j < W && i < nwords;
        words[j] = in ? q[i] : 0u;
        const float sc = in ? s[i / E::WPB] : 0.0f;
#pragma unroll
        for (int e = 0; e < 4; e++) {
            keys[j][e] = E::key(words[j], e, sc);
            if (in && i * 4 + e < n) valid |= 1u << (4 * j + e);
        }
    }
#pragma unroll
    for (int i = 0; i < (4 * COPIES * CS + TS_THREADS - 1) / TS_THREADS; i++)
        if (tid + TS_THREADS * i < 4 * COPIES * CS) hist[tid + TS_THREADS * i] = 0;
    __syncthreads();

    uint32_t tau = 0x7F800000u, keep = 0;
    if (k != 0) {
        uint32_t prefix = 0, need = k;
        for (int level = 0; level < 4; level++) {
            const int shift = 24 - 8 * level;
            uint32_t *h = hist + COPIES * CS * level;
            uint32_t *hp = h + CS * (tid & (COPIES - 1));
#pragma unroll
            for (uint32_t j = 0; j < TS8_W; j++)
#pragma unroll
                for (int e = 0; e < 4; e++)
                    if ((valid >> (4 * j + e)) & 1u) {
                        const uint32_t key = keys[j][e];
                        if (level == 0 || (key >> (shift + 8)) == prefix) atomicAdd(&hp[(key >> shift) & 0xFFu], 1u);
                    }
            __syncthreads();
            // the private copies are added up once (256 threads, one bin each), then every wave selects for itself from the sums: two
            // barriers per level instead of three.  (Letting every wave add up the eight copies itself saves one more barrier but reads
            // 128 KiB of LDS per level: measured slower, 8.0 against 7.5 us at N = 8192.)
            if (tid < 256) {
                uint32_t mine = 0;
#pragma unroll
                for (int cpy = 0; cpy < COPIES; cpy++) mine += h[CS * cpy + tid];
                hsum[256 * (level & 1) + tid] = mine;
            }
            __syncthreads();
            const u32x4 h4 = *reinterpret_cast<const u32x4 *>(hsum + 256 * (level & 1) + 252 - 4 * lane);       // lane l: bins 255 - 4 l ... 252 - 4 l
            const uint32_t t0 = h4.w, t1 = h4.z, t2 = h4.y, t3 = h4.x;
            const uint32_t sum = t0 + t1 + t2 + t3;
            const uint32_t incl = wave_scan_incl(sum);
            const unsigned long long hit = __ballot(incl >= need && incl - sum < need);
            const int L = __builtin_ctzll(hit);                          // exactly one lane: the level's total is >= need
            uint32_t above = __shfl(incl - sum, L);
            const uint32_t T0 = __shfl(t0, L), T1 = __shfl(t1, L), T2 = __shfl(t2, L);
            uint32_t pick = 0;
            if (above + T0 < need) { above += T0; pick = 1;
                if (above + T1 < need) { above += T1; pick = 2;
                    if (above + T2 < need) { above += T2; pick = 3; } } }
            prefix = (prefix << 8) | (255u - 4u * (uint32_t)L - pick);
            need -= above;
        }
        tau = prefix;
        keep = need;
    }
    uint32_t c = 0;
#pragma unroll
    for (uint32_t j = 0; j < TS8_W; j++)
#pragma unroll
        for (int e = 0; e < 4; e++) c += ((valid >> (4 * j + e)) & 1u) && keys[j][e] == tau;
    // ties in index order: the first `keep` of them survive (one barrier: the 16 wave totals)
    const uint32_t vinc = wave_scan_incl(c);
    if (lane == 63) wsum[tid >> 6] = vinc;
    __syncthreads();
    const uint32_t tot = lane < 16 ? wsum[lane] : 0;
    const uint32_t inc = wave_scan_incl(tot);
    uint32_t rank = vinc - c + __shfl(inc - tot, tid >> 6);
#pragma unroll
    for (uint32_t j = 0; j < TS8_W; j++) {
        const uint32_t i = w0 + j;
        if (j < W && i < nwords) {
            uint32_t outw = 0;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const uint32_t byte = words[j] & E::mask(e);
                if (!((valid >> (4 * j + e)) & 1u)) { outw |= byte; continue; }          // padding is left alone
                if (keys[j][e] > tau) outw |= byte;
                else if (keys[j][e] == tau) { if (rank < keep) outw |= byte; rank++; }
            }
            q[i] = outw;
        }
    }
}

// ---- large CloverVector4 vectors: the same weighted-candidate idea as k_thresh_small, across kernels ----------------------------
// One pass turns every 64-element block into its 9 magnitude counts (8 bytes); the three radix levels then run over these
// tables (12 bytes per block instead of 36 bytes of elements, 9 weighted histogram updates per block instead of 64), the tie
// counts come from the tables as well, and only the final pass touches the elements again: 2 element passes instead of 5.
// Radix levels: 12 + 12 + 8 bits (LEVEL 0 = most significant).  The fixed cost per workgroup of a level (16 KiB of LDS bins to
// clear and to flush with global atomics, which serialise per address at ~50 ns) is what such a kernel costs, not the 12 bytes per
// block it reads.  (A variant that let the last workgroup to finish do the selection was 6x slower: its one-ticket-per-workgroup
// atomic serialises the same way.)
// elements of block b equal to tau, from its table
__device__ __forceinline__ uint32_t th4_block_ties(unsigned long long c, float s7, uint32_t tau)
{
    uint32_t t = 0;
#pragma unroll
    for (int m = 0; m <= 8; m++) t += cand_key(s7, m) == tau ? (uint32_t)(c >> (7 * m)) & 0x7Fu : 0u;
    return t;
}

// ---- round 2: the same algorithm in 6 launches instead of 10 -------------------------------------------------------------------
// The three one-workgroup select kernels and the one-workgroup scan are gone: every workgroup of a kernel that needs the outcome
// of an earlier level recomputes it from that level's finished histogram (16 KiB out of L2, one block scan), and the apply kernel
// adds up the tie counts in front of its chunk itself (group totals + prefixes inside a group).  Each level has its own
// histogram, zeroed by the first kernel.  Same keys, same selection, same tie rule: results are unchanged bit for bit.
struct Th4Sel {
    uint32_t prefix, remaining;
    uint32_t size;       // elements in the selected bin (what the next level's histogram must add up to)
};

// one workgroup of 256 or more threads (the first 256 do the work, everybody takes the barriers): from the top of `hist`, the bin in
// which the cumulative count reaches `need`
// (two halves, so that a caller can put other loads between the histogram's loads and the first use of the bins)
template <int LEVEL>
struct Th4Bins {
    static constexpr int nb = LEVEL == 2 ? 256 : 4096;
    static constexpr int per = nb / 256;
    uint32_t bins[per];
    int top;
    bool on;
    __device__ __forceinline__ void load(const uint32_t *__restrict__ hist)
    {
        const int t = threadIdx.x;
        on = t < 256;
        top = (255 - (on ? t : 0)) * per + per - 1;              // thread t owns the t-th run of `per` bins from the top
#pragma unroll
        for (int i = 0; i < per; i++) bins[i] = hist[top - i];  // unconditional (the idle threads read thread 0's bins): the loads go out together
    }
    __device__ __forceinline__ Th4Sel select(uint32_t need, uint32_t prev, uint32_t *sel /* LDS[4] */, uint32_t *wsum /* LDS[4] */)
    {
        const int t = threadIdx.x;
        uint32_t sum = 0;
#pragma unroll
        for (int i = 0; i < per; i++) { if (!on) bins[i] = 0u; sum += bins[i]; }
        uint32_t v = wave_scan_incl(sum);
        __syncthreads();                                         // sel / wsum of an earlier call have been read
        if (on && (t & 63) == 63) wsum[t >> 6] = v;
        __syncthreads();
        if (on) {
            for (int w = 0; w < (t >> 6); w++) v += wsum[w];
            if (v >= need && v - sum < need) {
                uint32_t above = v - sum, size = bins[0];
                int i = 0;
#pragma unroll
                for (int j = 0; j < per - 1; j++)
                    if (i == j && above + bins[j] < need) { above += bins[j]; i = j + 1; size = bins[j + 1]; }
                const uint32_t bin = (uint32_t)(top - i);
                sel[0] = LEVEL == 0 ? bin : (LEVEL == 1 ? (prev << 12) | bin : (prev << 8) | bin);
                sel[1] = need - above;
                sel[2] = size;
            }
        }
        __syncthreads();
        return Th4Sel{sel[0], sel[1], sel[2]};
    }
};
template <int LEVEL>
__device__ __forceinline__ Th4Sel th4_wg_select(const uint32_t *__restrict__ hist, uint32_t need, uint32_t prev, uint32_t *sel /* LDS[4] */,
                                                uint32_t *wsum /* LDS[4] */)
{
    Th4Bins<LEVEL> b;
    b.load(hist);
    return b.select(need, prev, sel, wsum);
}

// the selections of levels 0 .. UPTO-1 (what level UPTO's histogram, or the tie count, needs)
template <int UPTO>
__device__ __forceinline__ Th4Sel th4_selected(const uint32_t *__restrict__ hists, uint32_t k, uint32_t *sel, uint32_t *wsum)
{
    Th4Sel r{0, k, 0};
    if (UPTO >= 1) r = th4_wg_select<0>(hists, k, 0, sel, wsum);
    if (UPTO >= 2) r = th4_wg_select<1>(hists + 4096, r.remaining, r.prefix, sel, wsum);
    if (UPTO >= 3) r = th4_wg_select<2>(hists + 8192, r.remaining, r.prefix, sel, wsum);
    return r;
}

// every block's 9 magnitude counts, and a clean slate for all three histograms
__global__ __launch_bounds__(256) void k_th4_count6(const u32x4 *__restrict__ q, uint64_t n, unsigned long long *__restrict__ cnt, uint64_t nblocks,
                                                    uint32_t *hists)
{
    if (blockIdx.x == 0)
        for (int i = threadIdx.x; i < 3 * 4096; i += 256) hists[i] = 0;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; b < nblocks; b += stride) {
        const u32x4 lo = q[2 * b], hi = q[2 * b + 1];
        const uint32_t w[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
        if (b * 64 + 64 <= n) {                                  // every block but possibly the last
            cnt[b] = th4_count_full_block(w);
            continue;
        }
        unsigned long long acc = 0;                              // the block that n cuts: element by element, the first n - 64 b of them
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const uint32_t ab = abs_nibbles(swap_nibbles(w[j]));
            const uint64_t first = b * 64 + 8 * j;
            const uint32_t valid = first >= n ? 0u : (n - first < 8 ? (uint32_t)(n - first) : 8u);
#pragma unroll
            for (uint32_t e = 0; e < 8; e++)
                if (e < valid) acc += 1ull << (7u * ((ab >> (4 * e)) & 0xFu));
        }
        cnt[b] = acc;
    }
}

// one block's nine weighted candidates into the bins of radix level LEVEL (lh: LDS), those under `prefix` only (LEVEL > 0)
template <int LEVEL>
__device__ __forceinline__ void th4_add_block(uint32_t *lh, unsigned long long c, float sc, uint32_t prefix, uint32_t m0)
{
    const float s7 = div7(sc);
    uint32_t m = m0;
#pragma unroll
    for (int it = 0; it <= 8; it++) {
        const uint32_t wgt = (uint32_t)(c >> (7 * m)) & 0x7Fu;
        if (wgt) {
            const uint32_t key = __float_as_uint(__builtin_fabsf(s7 * (float)m));        // cand_key(s7, m)
            if (LEVEL == 0) atomicAdd(&lh[key >> 20], wgt);
            else if (LEVEL == 1) { if ((key >> 20) == prefix) atomicAdd(&lh[(key >> 8) & 0xFFF], wgt); }
            else if (LEVEL == 2) { if ((key >> 8) == prefix) atomicAdd(&lh[key & 0xFF], wgt); }
            else if (LEVEL == 3) { if ((key >> 20) == prefix) atomicAdd(&lh[(key >> 12) & 0xFF], wgt); }      // the 12 + 8 + 12 split of
            else { if ((key >> 12) == prefix) atomicAdd(&lh[key & 0xFFF], wgt); }                              // k_th4_select_persist
        }
        m = m == 8 ? 0 : m + 1;
    }
}

#ifndef TH4_HU
#define TH4_HU 4             // blocks per thread and step of a histogram level (8 with half the workgroups: 30-37 us per level at n = 2^28, this: 20-25)
#endif
#ifndef TH4_HWG_PER_CU_X2
#define TH4_HWG_PER_CU_X2 4  // histogram workgroups (of 1024 threads) per TWO CUs
#endif
// radix level LEVEL over the candidate tables, with the earlier levels' selections recomputed per workgroup
template <int LEVEL>
__global__ __launch_bounds__(1024) void k_th4_hist6(const unsigned long long *__restrict__ cnt, const float *__restrict__ s, uint64_t nblocks,
                                                    uint32_t *__restrict__ hists, uint32_t k)
{
    __shared__ uint32_t lh[4096];
    __shared__ uint32_t sel[4], wsum[4];
    constexpr int NB = LEVEL == 2 ? 256 : 4096;
    for (int i = threadIdx.x; i < NB; i += blockDim.x) lh[i] = 0;
    const uint32_t prefix = th4_selected<LEVEL>(hists, k, sel, wsum).prefix;       // ends with a barrier: lh is clear for everybody
    if (LEVEL == 0) __syncthreads();
    // The 64 lanes of a wave hold 64 neighbouring blocks, whose scales -- hence whose keys for one magnitude m -- mostly fall into the
    // same bin: with every lane on the same m, one LDS atomic instruction is up to 64 updates of ONE address, which the LDS serialises.
    // So the lanes walk the nine magnitudes in rotated order, lane l starting at m = l mod 9: one instruction then spreads over nine
    // bins.  Sums are order-free: same histogram.
    const uint32_t m0 = (threadIdx.x & 63) % 9u;
    auto add_block = [&](unsigned long long c, float sc) { th4_add_block<LEVEL>(lh, c, sc, prefix, m0); };
    // TH4_HU blocks per thread and step, all loads first: a thread walks only a handful of steps, so the level is as long as its
    // chain of load latencies (a block whose table is all zero adds nothing: the tail needs no branch)
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; b < nblocks; b += TH4_HU * stride) {
        unsigned long long c[TH4_HU];
        float sc[TH4_HU];
#pragma unroll
        for (int u = 0; u < TH4_HU; u++) {
            const uint64_t bu = b + u * stride;
            const bool in = bu < nblocks;
            const uint64_t bc = in ? bu : b;                                  // clamped address, unconditional load
            c[u] = cnt[bc];
            sc[u] = s[bc];
            if (!in) c[u] = 0ull;
        }
#pragma unroll
        for (int u = 0; u < TH4_HU; u++) add_block(c[u], sc[u]);
    }
    __syncthreads();
    uint32_t *hist = hists + 4096 * LEVEL;
    for (int i = threadIdx.x; i < NB; i += blockDim.x) if (lh[i]) atomicAdd(&hist[i], lh[i]);
}

#define TH4_GROUPS 512u      // workgroups of the tie kernel (and entries the apply kernel adds up at most); every one re-reads the
                             // three histograms (33 KiB) to recompute tau: 2048 groups were slower (68 MB of L2 reads) than 512
#define TH4_MAX_CPG 512u     // chunks per group at most: n < 2^32 -> fewer than 2^18 chunks of 256 blocks, over 512 groups

// tie counts.  Workgroup g takes the chunks (of 256 blocks = one workgroup of the apply kernel) [g * cpg, (g + 1) * cpg): it
// leaves the EXCLUSIVE prefix of every chunk inside its group in chunk_ties and the group's total in group_ties; workgroup 0
// also publishes tau and the number of ties to keep.  k == 0: nothing survives (tau beyond any magnitude).
__global__ __launch_bounds__(256) void k_th4_ties6(const unsigned long long *__restrict__ cnt, const float *__restrict__ s, uint64_t nblocks,
                                                   const uint32_t *__restrict__ hists, uint32_t k, ThreshState *__restrict__ ts,
                                                   uint32_t *__restrict__ chunk_ties, uint32_t *__restrict__ group_ties, uint32_t nchunks, uint32_t cpg)
{
    __shared__ uint32_t sel[4], wsum[4];
    Th4Sel r{0x7F800000u, 0, 0};
    if (k != 0) r = th4_selected<3>(hists, k, sel, wsum);
    const uint32_t tau = r.prefix, keep = r.remaining;
    if (blockIdx.x == 0 && threadIdx.x == 0) *ts = ThreshState{tau, keep, tau, keep};
    // chunk totals first -- a wave per chunk, four lane-steps of 64 blocks and one wave reduction, no barrier inside the loop -- then ONE
    // block scan over the group's (at most 512) totals.  (Round 3 took a workgroup scan and two barriers per chunk: 24 us at n = 2^28.)
    __shared__ uint32_t tot[TH4_MAX_CPG];
    const uint32_t c0 = blockIdx.x * cpg, c1 = c0 + cpg < nchunks ? c0 + cpg : nchunks;
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (uint32_t c = c0 + wave; c < c1; c += 8) {                        // two chunks per step: all sixteen loads first
        unsigned long long cc[2][4];
        float sc[2][4];
#pragma unroll
        for (int u = 0; u < 2; u++)
#pragma unroll
            for (int q4 = 0; q4 < 4; q4++) {                              // (an all-zero table has no ties: the tails need no branch)
                const uint64_t b = (uint64_t)(c + 4 * u) * 256 + 64 * q4 + lane;
                const bool in = c + 4 * u < c1 && b < nblocks;
                const uint64_t bc = in ? b : 0;                               // clamped address, unconditional load: all sixteen go out
                cc[u][q4] = cnt[bc];                                          // back to back (a predicated load becomes a branch + wait)
                sc[u][q4] = s[bc];
                if (!in) cc[u][q4] = 0ull;
            }
#pragma unroll
        for (int u = 0; u < 2; u++) {
            uint32_t t = 0;
#pragma unroll
            for (int q4 = 0; q4 < 4; q4++) t += th4_block_ties(cc[u][q4], div7(sc[u][q4]), tau);
            t = wave_scan_incl(t);
            if (lane == 63 && c + 4 * u < c1) tot[c + 4 * u - c0] = t;
        }
    }
    __syncthreads();
    const uint32_t i0 = 2 * threadIdx.x, n_here = c1 > c0 ? c1 - c0 : 0;
    const uint32_t a = i0 < n_here ? tot[i0] : 0u, b2 = i0 + 1 < n_here ? tot[i0 + 1] : 0u;
    uint32_t incl = wave_scan_incl(a + b2);
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    uint32_t before = incl - (a + b2);
    for (uint32_t w = 0; w < wave; w++) before += wsum[w];
    if (i0 < n_here) chunk_ties[c0 + i0] = before;
    if (i0 + 1 < n_here) chunk_ties[c0 + i0 + 1] = before + a;
    if (threadIdx.x == 0) group_ties[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// thread = one 64-element block: survivors above tau and the first `keep` ties in index order, whole words at a time; the rank of
// the chunk's first tie is computed here: ties of all groups in front + the chunk's prefix inside its group
__global__ __launch_bounds__(256) void k_th4_apply6(u32x4 *__restrict__ q, const float *__restrict__ s, uint64_t n, uint64_t nblocks,
                                                    const unsigned long long *__restrict__ cnt, const ThreshState *__restrict__ ts,
                                                    const uint32_t *__restrict__ chunk_ties, const uint32_t *__restrict__ group_ties, uint32_t cpg)
{
    __shared__ uint32_t wsum[4], gsum[4];
    const uint32_t tau = ts->tau, keep = ts->ties_keep;
    const uint32_t group = blockIdx.x / cpg;
    uint32_t before = 0;
    for (uint32_t g = threadIdx.x; g < group; g += 256) before += group_ties[g];
    before = wave_scan_incl(before);
    if ((threadIdx.x & 63) == 63) gsum[threadIdx.x >> 6] = before;
    const uint64_t b = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const bool in = b < nblocks;
    const float s7 = in ? div7(s[b]) : 1.0f;
    const uint32_t mine = in ? th4_block_ties(cnt[b], s7, tau) : 0u;
    const uint32_t incl = wave_scan_incl(mine);
    if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint32_t rank = gsum[0] + gsum[1] + gsum[2] + gsum[3] + chunk_ties[blockIdx.x] + incl - mine;
    for (int w = 0; w < (int)(threadIdx.x >> 6); w++) rank += wsum[w];
    if (!in) return;
    uint32_t lo_t = 0, hi_t = 0;                                   // magnitudes >= hi_t are above tau, [lo_t, hi_t) equal it
#pragma unroll
    for (int m = 0; m <= 8; m++) {
        const uint32_t key = cand_key(s7, m);
        lo_t += key < tau;
        hi_t += key <= tau;
    }
    const u32x4 lo = q[2 * b], hi = q[2 * b + 1];
    uint32_t w[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const uint32_t ab = abs_nibbles(swap_nibbles(w[j]));
        const uint64_t first = b * 64 + 8 * j;
        const uint32_t valid = first_nibbles(first >= n ? 0u : (n - first < 8 ? (uint32_t)(n - first) : 8u));
        const uint32_t above = ge_nibbles(ab, hi_t);
        uint32_t kb = above | (0x88888888u & ~valid);              // padding is left alone
        uint32_t t = ge_nibbles(ab, lo_t) & ~above & valid;
        const uint32_t nt = __popc(t), room = keep > rank ? keep - rank : 0;
        if (room >= nt) kb |= t;
        else for (uint32_t r = 0; r < room; r++) { kb |= t & (0u - t); t &= t - 1; }
        rank += nt;
        w[j] &= swap_nibbles((kb >> 3) * 0xFu);
    }
    q[2 * b] = u32x4{w[0], w[1], w[2], w[3]};
    q[2 * b + 1] = u32x4{w[4], w[5], w[6], w[7]};
}


// ---- round 6: the same call in THREE launches -------------------------------------------------------------------------------------
//   k_th4_count_hist0     the pass over the nibbles writes the per-block tables AND bins their nine candidates for radix level 0 (the LDS
//                         atomics run beside the streaming loads);
//   k_th4_select_persist  ONE workgroup per CU, all resident: level 1, level 2 and the tie prefixes with two grid-wide arrivals in between
//                         (an agent-scope counter per arrival; what crosses workgroups -- the level histograms -- is written with agent-
//                         scope atomics and read back with agent-scope loads, so no L2 write-back is needed: a wave waits for its own
//                         atomics' acknowledgements, the workgroup's barrier collects the waves, one lane arrives).
//                         Once level 0 has chosen a bin of NORMAL numbers (12 bits = sign, exponent, 3 mantissa bits: a bin spans a
//                         factor <= 9/8), at most ONE of a block's magnitudes m = 1 .. 8 can lie in it (consecutive magnitudes are a factor
//                         >= 8/7 apart; m = 0 has key 0), so the block shrinks to one word {its 20 low key bits, its weight}: level 1
//                         costs 8 multiplies per block instead of 9 binned candidates, level 2 and the tie counts read 4 bytes per
//                         block.  A level-0 bin of zeros / subnormals (tau = 0: k beyond the non-zero elements) can hold several
//                         magnitudes of one block: that case -- uniform over the grid -- takes the general loops of k_th4_hist6;
//   k_th4_apply6          as before (reads tau and the tie ranks), and clears the control block for the next call.
// Same keys, same selection, same tie rule as the six-launch form: the results are the same bit for bit (tests/test_threshold_large3.py).
#define TH4_CTL_WORDS (3u * 4096u + 16u)               // hist0, hist1 (16 copies of 256 bins), hist2, two result words: library-owned, all zero between calls
#define TH4_P_MAX_CPG 2048u                            // chunks (of 256 blocks) per workgroup of the persistent kernel at most

#ifndef TH4_K1_U
#define TH4_K1_U 2           // blocks per thread and step
#endif
#ifndef TH4_K1_WAVES
#define TH4_K1_WAVES 4       // waves per SIMD the register budget allows (8: two workgroups of 1024 per CU)
#endif
__global__ __launch_bounds__(1024, TH4_K1_WAVES) void k_th4_count_hist0(const u32x4 *__restrict__ q, const float *__restrict__ s, uint64_t n,
                                                          unsigned long long *__restrict__ cnt, uint64_t nblocks, uint32_t *__restrict__ hist0)
{
    __shared__ uint32_t lh[4096];
    for (int i = threadIdx.x; i < 4096; i += 1024) lh[i] = 0;
    __syncthreads();
    const uint32_t m0 = (threadIdx.x & 63) % 9u;
    const uint64_t stride = (uint64_t)gridDim.x * 1024;
    for (uint64_t b = (uint64_t)blockIdx.x * 1024 + threadIdx.x; b < nblocks; b += TH4_K1_U * stride) {
        u32x4 lo[TH4_K1_U], hi[TH4_K1_U];
        float sc[TH4_K1_U];
#pragma unroll
        for (int u = 0; u < TH4_K1_U; u++) {                     // clamped addresses, unconditional loads: all six go out together
            const uint64_t bu = b + u * stride, bc = bu < nblocks ? bu : b;
            lo[u] = q[2 * bc];
            hi[u] = q[2 * bc + 1];
            sc[u] = s[bc];
        }
#pragma unroll
        for (int u = 0; u < TH4_K1_U; u++) {
            const uint64_t bu = b + u * stride;
            if (bu >= nblocks) continue;
            const uint32_t w[8] = {lo[u].x, lo[u].y, lo[u].z, lo[u].w, hi[u].x, hi[u].y, hi[u].z, hi[u].w};
            unsigned long long c = 0;
            if (bu * 64 + 64 <= n) {
                c = th4_count_full_block(w);
            } else {                                             // the block that n cuts: element by element, the first n - 64 b of them
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const uint32_t ab = abs_nibbles(swap_nibbles(w[j]));
                    const uint64_t first = bu * 64 + 8 * j;
                    const uint32_t valid = first >= n ? 0u : (n - first < 8 ? (uint32_t)(n - first) : 8u);
#pragma unroll
                    for (uint32_t e = 0; e < 8; e++)
                        if (e < valid) c += 1ull << (7u * ((ab >> (4 * e)) & 0xFu));
                }
            }
            cnt[bu] = c;
            th4_add_block<0>(lh, c, sc[u], 0u, m0);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 4096; i += 1024)
        if (lh[i]) atomicAdd(&hist0[i], lh[i]);
}

// One radix level's hand-over.  Every workgroup flushes its non-empty bins with agent-scope atomics and goes on to poll the level's result
// word; nobody counts arrivals.  Workgroup 0 instead reads the histogram (agent-scope loads, past its L2) until its bins ADD UP to the
// number of elements the level distributes -- the size of the bin chosen one level up, known from that level's histogram -- which they do
// exactly when every flush has landed (bins only grow).  It then selects the level's bin and publishes {prefix, remaining} as one 64-bit
// word (never zero: remaining >= 1).  Two memory round trips per level on the critical path (flush -> visible, result -> visible) where an
// arrival counter and a re-read of the histogram by every workgroup took four and a hot spot of 256 x 16 KiB on the same 128 lines.
// LEVEL 1: TH4_L1_COPIES copies of 256 bins, workgroup g flushes into copy g mod TH4_L1_COPIES (atomics on ONE address serialise at
// ~40 ns each: 256 workgroups on one copy cost 10 us); LEVEL 2: 4096 bins, one copy (a workgroup has ~20 candidates left).
#define TH4_L1_COPIES 16u
template <int LEVEL>
__device__ __forceinline__ Th4Sel th4_handover(const uint32_t *hist, unsigned long long *result, uint32_t expected, uint32_t need, uint32_t prev,
                                               uint32_t *lh /* LDS[4096]: level 1: bins 0..255 zero on entry */, uint32_t *sel, uint32_t *wsum,
                                               uint32_t *wtot /* LDS[16] */)
{
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (blockIdx.x == 0) {
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(hist), 0, 4096 * 4, 0x00020000);
        u32x4 v;
        uint32_t spins = 0;
        unsigned long long t_start = 0;
        while (true) {
            v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, 16u * tid, 0, /*sc1*/ 16);
            const uint32_t part = wave_scan_incl(v.x + v.y + v.z + v.w);
            if (lane == 63) wtot[wave] = part;
            __syncthreads();
            uint32_t total = 0;
#pragma unroll
            for (int w = 0; w < 16; w++) total += wtot[w];
            __syncthreads();                                     // wtot has been read: the next round may overwrite it
            if (total == expected) break;
            __builtin_amdgcn_s_sleep(2);
            if ((++spins & 1023u) == 0) {                        // bounded: 4 s on the 100 MHz wall clock, then a trap
                const unsigned long long now = __builtin_amdgcn_s_memrealtime();
                if (!t_start) t_start = now;
                else if (now - t_start > 400000000ull) __builtin_trap();
            }
        }
        if (LEVEL == 1) {                                        // thread = copy tid >> 6, bins 4 (tid & 63) ..: add the copies up
            if (v.x) atomicAdd(&lh[4 * lane], v.x);
            if (v.y) atomicAdd(&lh[4 * lane + 1], v.y);
            if (v.z) atomicAdd(&lh[4 * lane + 2], v.z);
            if (v.w) atomicAdd(&lh[4 * lane + 3], v.w);
        } else {
            *reinterpret_cast<u32x4 *>(lh + 4 * tid) = v;
        }
        __syncthreads();
        const Th4Sel r = LEVEL == 1 ? th4_wg_select<2>(lh, need, prev, sel, wsum) : th4_wg_select<1>(lh, need, prev, sel, wsum);
        if (tid == 0) __hip_atomic_store(result, ((unsigned long long)r.prefix << 32) | r.remaining, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (LEVEL == 1) {                                        // the bins go back to zero for level 2
            if (tid < 256) lh[tid] = 0;
            __syncthreads();
        }
        return r;
    }
    if (tid == 0) {
        uint32_t spins = 0;
        unsigned long long t_start = 0, v;
        while ((v = __hip_atomic_load(result, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0ull) {
            __builtin_amdgcn_s_sleep(2);
            if ((++spins & 1023u) == 0) {
                const unsigned long long now = __builtin_amdgcn_s_memrealtime();
                if (!t_start) t_start = now;
                else if (now - t_start > 400000000ull) __builtin_trap();
            }
        }
        sel[0] = (uint32_t)(v >> 32);
        sel[1] = (uint32_t)v;
    }
    __syncthreads();
    return Th4Sel{sel[0], sel[1], 0u};
}

// REG: a workgroup's range is at most 16 x 1024 blocks (n <= 2^28 on 256 CUs): thread t owns blocks b0 + t + 1024 u, loads their tables and
// scales in one go in front of everything else, and keeps their candidate words in registers from level 1 to the tie counts -- no load sits
// between the hand-overs; otherwise the words go through `cand`.
// Radix split behind level 0's 12 bits: 8 bits, then 12 (the six-launch form takes 12, then 8): at level 1 a workgroup's ~5000 candidates
// fill whatever bins there are, and every non-empty bin is one agent-scope atomic per workgroup; at level 2 only the ~20 candidates of the
// chosen bin are left per workgroup.  The selection is the same.
template <bool REG>
__global__ __launch_bounds__(1024) void k_th4_select_persist(const unsigned long long *__restrict__ cnt, const float *__restrict__ s, uint64_t nblocks,
                                                             uint32_t *__restrict__ ctl, uint32_t k, ThreshState *__restrict__ ts,
                                                             uint32_t *__restrict__ chunk_ties, uint32_t *__restrict__ group_ties, uint32_t nchunks,
                                                             uint32_t cpg, uint32_t *__restrict__ cand, unsigned long long *dbg)
{
#define TH4_STAMP(i) do { if (dbg && threadIdx.x == 0) dbg[blockIdx.x * 16 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
    TH4_STAMP(0);
    __shared__ __attribute__((aligned(16))) uint32_t lh[4096];
    __shared__ uint32_t tot[TH4_P_MAX_CPG];
    __shared__ uint32_t sel[4], wsum[4], wtot[16];
    uint32_t *hist0 = ctl, *hist1 = ctl + 4096, *hist2 = ctl + 8192;
    unsigned long long *result = reinterpret_cast<unsigned long long *>(ctl + 12288);
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t c0 = blockIdx.x * cpg, c1 = c0 + cpg < nchunks ? c0 + cpg : nchunks;
    const uint64_t b0 = (uint64_t)c0 * 256, b1 = (uint64_t)c1 * 256 < nblocks ? (uint64_t)c1 * 256 : nblocks;
    const uint32_t m0 = lane % 9u;
    unsigned long long cr[REG ? 16 : 1];
    float scr[REG ? 16 : 1];
    auto load_batch = [&](int bt) {                               // four blocks: eight loads
#pragma unroll
        for (int u = 4 * bt; u < 4 * bt + 4; u++) {
            const uint64_t bu = b0 + tid + 1024u * u, bc = bu < b1 ? bu : b0;
            cr[REG ? u : 0] = cnt[bc];
            scr[REG ? u : 0] = s[bc];
        }
    };
    // level 0 was finished by the launch in front: ordinary loads.  The first two batches of tables go out behind them and arrive under the
    // selection (a workgroup's 192 KiB are 8 us of HBM time for the grid: every batch is loaded two batches ahead of its arithmetic)
    Th4Bins<0> bins0;
    bins0.load(hist0);
    asm volatile("" ::: "memory");
    if (REG) { load_batch(0); load_batch(1); }
    asm volatile("" ::: "memory");
    for (int i = tid; i < 4096; i += 1024) lh[i] = 0;
    for (uint32_t i = tid; i < TH4_P_MAX_CPG; i += 1024) tot[i] = 0;
    const Th4Sel s0 = bins0.select(k, 0, sel, wsum);                                // ends with a barrier: lh is clear for everybody
    TH4_STAMP(1);
    // a bin of finite normal numbers holds one candidate per block at most (subnormals: linear bins; the bin of inf / NaN: every magnitude
    // of a block with such a scale)
    const uint32_t bexp = (s0.prefix >> 3) & 0xFFu;
    const bool one = bexp != 0 && bexp != 0xFFu;
    const float lo = __uint_as_float(s0.prefix << 20);                             // the bin's lower edge
    // The block's candidate in the bin, {20 low key bits, weight << 20}, or 0.  Keys grow with m, so it can only be m* = the first m whose key
    // reaches `lo`, and m* is floor or ceil of lo / |s7|: three magnitudes around round(lo * rcp(|s7|)) are tried with the exact key
    // expression (rcp's last-bit error moves the quotient by 1e-6 at most).  Subnormal scales (where rcp is not to be trusted): all eight.
    auto candidate = [&](unsigned long long c, float sc) -> uint32_t {
        const float s7 = div7(sc), a = __builtin_fabsf(s7);
        uint32_t cw = 0;
        if (a < 1.17549435e-38f) {
            if (a == 0.0f) return 0u;
#pragma unroll
            for (int m = 1; m <= 8; m++) {
                const uint32_t key = cand_key(s7, m), wgt = (uint32_t)(c >> (7 * m)) & 0x7Fu;
                if ((key >> 20) == s0.prefix && wgt) cw = (key & 0xFFFFFu) | (wgt << 20);
            }
            return cw;
        }
        const float mid = __builtin_fminf(__builtin_fmaxf(__builtin_rintf(lo * __builtin_amdgcn_rcpf(a)), 2.0f), 7.0f);
#pragma unroll
        for (int d = -1; d <= 1; d++) {
            const float mf = mid + (float)d;
            const uint32_t key = __float_as_uint(__builtin_fabsf(s7 * mf)), wgt = (uint32_t)(c >> (7u * (uint32_t)mf)) & 0x7Fu;
            if ((key >> 20) == s0.prefix && wgt) cw = (key & 0xFFFFFu) | (wgt << 20);
        }
        return cw;
    };
    uint32_t cwr[REG ? 16 : 1];
    // ---- level 1: the 8 bits behind level 0's 12 (and the blocks' candidate words) ----
    if (REG) {
#pragma unroll
        for (int bt = 0; bt < 4; bt++) {
            if (bt + 2 < 4) load_batch(bt + 2);
            asm volatile("" ::: "memory");
#pragma unroll
            for (int u = 4 * bt; u < 4 * bt + 4; u++) {
                cwr[REG ? u : 0] = 0;
                if (b0 + 1024u * u >= b1) continue;               // uniform: a short range (small n) skips the arithmetic
                const unsigned long long c = b0 + tid + 1024u * u < b1 ? cr[REG ? u : 0] : 0ull;   // an all-zero table has no candidate
                if (one) {
                    const uint32_t cw = candidate(c, scr[REG ? u : 0]);
                    cwr[REG ? u : 0] = cw;
                    if (cw) atomicAdd(&lh[(cw >> 12) & 0xFFu], cw >> 20);
                } else {
                    th4_add_block<3>(lh, c, scr[REG ? u : 0], s0.prefix, m0);
                }
            }
        }
    } else {
        for (uint64_t b = b0 + tid; b < b1; b += 4 * 1024) {
            unsigned long long c[4];
            float sc[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const uint64_t bu = b + 1024 * u, bc = bu < b1 ? bu : b;
                c[u] = cnt[bc];
                sc[u] = s[bc];
                if (bu >= b1) c[u] = 0ull;
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                if (!one) { th4_add_block<3>(lh, c[u], sc[u], s0.prefix, m0); continue; }
                const uint32_t cw = candidate(c[u], sc[u]);
                if (b + 1024 * u < b1) cand[b + 1024 * u] = cw;
                if (cw) atomicAdd(&lh[(cw >> 12) & 0xFFu], cw >> 20);
            }
        }
    }
    __syncthreads();
    TH4_STAMP(2);
    if (tid < 256 && lh[tid]) {
        __hip_atomic_fetch_add(&hist1[(blockIdx.x % TH4_L1_COPIES) * 256u + tid], lh[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        lh[tid] = 0;
    }
    __syncthreads();
    TH4_STAMP(3);
    const Th4Sel s1 = th4_handover<1>(hist1, result, s0.size, s0.remaining, s0.prefix, lh, sel, wsum, wtot);     // 256 bins: prefix = 20 bits
    TH4_STAMP(4);
    // ---- level 2: the last 12 bits ----
    if (REG && one) {
#pragma unroll
        for (int u = 0; u < 16; u++) {
            const uint32_t cw = cwr[REG ? u : 0];
            if (cw && ((cw >> 12) & 0xFFu) == (s1.prefix & 0xFFu)) atomicAdd(&lh[cw & 0xFFFu], cw >> 20);
        }
    } else {
        for (uint64_t b = b0 + tid; b < b1; b += 4 * 1024) {
            if (one) {
                uint32_t cw[4];
#pragma unroll
                for (int u = 0; u < 4; u++) { const uint64_t bu = b + 1024 * u; cw[u] = cand[bu < b1 ? bu : b]; if (bu >= b1) cw[u] = 0; }
#pragma unroll
                for (int u = 0; u < 4; u++)
                    if (cw[u] && ((cw[u] >> 12) & 0xFFu) == (s1.prefix & 0xFFu)) atomicAdd(&lh[cw[u] & 0xFFFu], cw[u] >> 20);
            } else {
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const uint64_t bu = b + 1024 * u;
                    if (bu < b1) th4_add_block<4>(lh, cnt[bu], s[bu], s1.prefix, m0);
                }
            }
        }
    }
    __syncthreads();
    TH4_STAMP(5);
    for (int i = tid; i < 4096; i += 1024)
        if (lh[i]) __hip_atomic_fetch_add(&hist2[i], lh[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();                                              // workgroup 0 writes the whole histogram over lh next
    TH4_STAMP(6);
    const Th4Sel s2 = th4_handover<2>(hist2, result + 1, s1.size, s1.remaining, s1.prefix, lh, sel, wsum, wtot);   // 4096 bins: all 32 bits
    TH4_STAMP(7);
    const uint32_t tau = s2.prefix, keep = s2.remaining;
    if (blockIdx.x == 0 && tid == 0) *ts = ThreshState{tau, keep, tau, keep};
    // ---- the ties: per chunk of 256 blocks (one workgroup of the apply kernel) the number in front of it inside this workgroup's range ----
    if (REG && one) {
#pragma unroll
        for (int u = 0; u < 16; u++) {                            // block b0 + tid + 1024 u lies in chunk c0 + 4 u + (tid >> 8)
            if (b0 + 1024u * u >= b1) continue;                   // uniform: a short range has no such blocks
            const uint32_t cw = cwr[REG ? u : 0];
            uint32_t t = (cw && (cw & 0xFFFFFu) == (tau & 0xFFFFFu)) ? cw >> 20 : 0u;
            t = wave_scan_incl(t);
            if (lane == 63 && t) atomicAdd(&tot[4 * u + (tid >> 8)], t);
        }
    } else {
        for (uint32_t c = c0 + (tid >> 8); c < c1; c += 4) {
            const uint64_t b = (uint64_t)c * 256 + (tid & 255);
            uint32_t t = 0;
            if (b < b1) {
                if (one) { const uint32_t cw = cand[b]; t = (cw && (cw & 0xFFFFFu) == (tau & 0xFFFFFu)) ? cw >> 20 : 0u; }
                else t = th4_block_ties(cnt[b], div7(s[b]), tau);
            }
            t = wave_scan_incl(t);
            if (lane == 63 && t) atomicAdd(&tot[c - c0], t);
        }
    }
    __syncthreads();
    TH4_STAMP(8);
    const uint32_t i0 = 2 * tid, n_here = c1 > c0 ? c1 - c0 : 0;
    const uint32_t a = i0 < n_here ? tot[i0] : 0u, a2 = i0 + 1 < n_here ? tot[i0 + 1] : 0u;
    const uint32_t incl = wave_scan_incl(a + a2);
    if (lane == 63) wtot[wave] = incl;
    __syncthreads();
    uint32_t before = incl - (a + a2), total = 0;
    for (uint32_t w = 0; w < 16; w++) { if (w < wave) before += wtot[w]; total += wtot[w]; }
    if (i0 < n_here) chunk_ties[c0 + i0] = before;
    if (i0 + 1 < n_here) chunk_ties[c0 + i0 + 1] = before + a;
    if (tid == 0) group_ties[blockIdx.x] = total;
    TH4_STAMP(9);
}

// ---- the apply pass of the three-launch form --------------------------------------------------------------------------------------
// What k_th4_apply6 does, with a third of its arithmetic (that kernel is VALU-bound: ~400 operations per block at 5.3 TB/s):
//  * a block's ties are counted on its own nibbles, which the pass holds anyway, not on its table (8 bytes per block less to read);
//  * FULL blocks are handled on bit planes, as the count pass does: two 4 x 4 bit transposes turn the block into the planes of its 64
//    nibbles, |v| is taken on the planes, "magnitude >= t" is a 4-bit comparator against the lane's constants (13 boolean operations for
//    32 nibbles), survivors are an AND on the planes, and the same transpose brings the words back;
//  * only a block whose ties are cut by the budget (the ONE block where the running rank crosses `keep`) and the block that n cuts take
//    the word-by-word walk of k_th4_apply6 (the order of the elements inside a word matters there only).
// mag >= t on planes (t in 0 .. 9; T[i] = all ones where bit i of t is set): from the lowest bit up
__device__ __forceinline__ uint32_t th4_planes_ge(const uint32_t A[4], const uint32_t T[4])
{
    uint32_t g = A[0] | ~T[0];
#pragma unroll
    for (int i = 1; i < 4; i++) g = (A[i] & ~T[i]) | (~(A[i] ^ T[i]) & g);
    return g;
}

// word-by-word: the block's words with everything but its survivors cleared; `rank` = ties in front of the block
__device__ __forceinline__ void th4_apply_block_words(uint32_t w[8], uint64_t b, uint64_t n, uint32_t lo_t, uint32_t hi_t, uint32_t rank, uint32_t keep)
{
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const uint32_t ab = abs_nibbles(swap_nibbles(w[j]));
        const uint64_t first = b * 64 + 8 * j;
        const uint32_t valid = first_nibbles(first >= n ? 0u : (n - first < 8 ? (uint32_t)(n - first) : 8u));
        const uint32_t above = ge_nibbles(ab, hi_t);
        uint32_t kb = above | (0x88888888u & ~valid);              // padding is left alone
        uint32_t t = ge_nibbles(ab, lo_t) & ~above & valid;
        const uint32_t nt = __popc(t), room = keep > rank ? keep - rank : 0;
        if (room >= nt) kb |= t;
        else for (uint32_t r = 0; r < room; r++) { kb |= t & (0u - t); t &= t - 1; }
        rank += nt;
        w[j] &= swap_nibbles((kb >> 3) * 0xFu);
    }
}

__global__ __launch_bounds__(256) void k_th4_apply3(u32x4 *__restrict__ q, const float *__restrict__ s, uint64_t n, uint64_t nblocks,
                                                    const ThreshState *__restrict__ ts, const uint32_t *__restrict__ chunk_ties,
                                                    const uint32_t *__restrict__ group_ties, uint32_t cpg, uint32_t *__restrict__ clean,
                                                    uint32_t clean_words)
{
    __shared__ uint32_t wsum[4], gsum[4];
    if (blockIdx.x == 0)                                           // the control block goes back all zero for the next call
        for (uint32_t i = threadIdx.x; i < clean_words; i += 256) clean[i] = 0;
    const uint64_t b = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const bool in = b < nblocks, full = b * 64 + 64 <= n;
    const uint64_t bc = in ? b : nblocks - 1;
    const u32x4 lo = q[2 * bc], hi = q[2 * bc + 1];                  // clamped, unconditional: the loads go out first
    const float sc = s[bc];
    const uint32_t tau = ts->tau, keep = ts->ties_keep;
    const uint32_t group = blockIdx.x / cpg;
    uint32_t before = 0;
    for (uint32_t g = threadIdx.x; g < group; g += 256) before += group_ties[g];
    before = wave_scan_incl(before);
    if ((threadIdx.x & 63) == 63) gsum[threadIdx.x >> 6] = before;
    const float s7 = div7(sc);
    uint32_t lo_t = 0, hi_t = 0;                                   // magnitudes >= hi_t are above tau, [lo_t, hi_t) equal it
#pragma unroll
    for (int m = 0; m <= 8; m++) {
        const uint32_t key = cand_key(s7, m);
        lo_t += key < tau;
        hi_t += key <= tau;
    }
    uint32_t w[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    uint32_t V[2][4], above[2], tie[2], mine = 0;
    if (full) {
        uint32_t TH[4], TL[4];
#pragma unroll
        for (int i = 0; i < 4; i++) { TH[i] = 0u - ((hi_t >> i) & 1u); TL[i] = 0u - ((lo_t >> i) & 1u); }
#pragma unroll
        for (int h = 0; h < 2; h++) {
            th4_planes(w[4 * h], w[4 * h + 1], w[4 * h + 2], w[4 * h + 3], V[h]);
            const uint32_t low = V[h][1] | V[h][0];
            const uint32_t A[4] = {V[h][0], V[h][1] ^ (V[h][3] & V[h][0]), V[h][2] ^ (V[h][3] & low), V[h][3] & ~(V[h][2] | low)};
            above[h] = th4_planes_ge(A, TH);
            tie[h] = th4_planes_ge(A, TL) & ~above[h];
            mine += __popc(tie[h]);
        }
    } else if (in) {                                               // the block that n cuts
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const uint32_t ab = abs_nibbles(swap_nibbles(w[j]));
            const uint64_t first = b * 64 + 8 * j;
            const uint32_t valid = first_nibbles(first >= n ? 0u : (n - first < 8 ? (uint32_t)(n - first) : 8u));
            mine += __popc(ge_nibbles(ab, lo_t) & ~ge_nibbles(ab, hi_t) & valid);
        }
    }
    const uint32_t incl = wave_scan_incl(mine);
    if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint32_t rank = gsum[0] + gsum[1] + gsum[2] + gsum[3] + chunk_ties[blockIdx.x] + incl - mine;
    for (int wv = 0; wv < (int)(threadIdx.x >> 6); wv++) rank += wsum[wv];
    if (!in) return;
    const uint32_t room = keep > rank ? keep - rank : 0;
    if (full && (room >= mine || room == 0)) {                     // all of the block's ties survive, or none
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const uint32_t K = room ? above[h] | tie[h] : above[h];
            uint32_t P[4];
            th4_planes(V[h][0] & K, V[h][1] & K, V[h][2] & K, V[h][3] & K, P);        // the transpose is its own inverse
            w[4 * h] = P[0]; w[4 * h + 1] = P[1]; w[4 * h + 2] = P[2]; w[4 * h + 3] = P[3];
        }
    } else {
        th4_apply_block_words(w, b, n, lo_t, hi_t, rank, keep);
    }
    q[2 * b] = u32x4{w[0], w[1], w[2], w[3]};
    q[2 * b + 1] = u32x4{w[4], w[5], w[6], w[7]};
}

// (TH4_GROUPS / TH4_MAX_CPG are defined in front of k_th4_ties6)

// workspace layout: [3 histograms of 4096 u32][ThreshState, 256 B][group_ties: 512 u32][chunk_ties: nblocks/256 + 1 u32, padded to
// 256 B][cnt: nblocks u64]
static int threshold4_large(uint32_t *q, const float *s, uint64_t n, uint64_t n_pad, uint64_t k, void *workspace, hipStream_t st)
{
    const uint64_t nblocks = (n + 63) / 64;
    const uint32_t nchunks = (uint32_t)((nblocks + 255) / 256);
    uint32_t *hists = (uint32_t *)workspace;
    ThreshState *ts = (ThreshState *)(hists + 3 * 4096);
    uint32_t *group_ties = (uint32_t *)((char *)ts + 256);
    uint32_t *chunk_ties = group_ties + TH4_GROUPS;
    unsigned long long *cnt = (unsigned long long *)((char *)chunk_ties + (((uint64_t)(n_pad / 64 / 256 + 1) * 4 + 255) & ~255ull));
    // the three-launch form (round 6): k != 0, a zeroed control block on this stream, at most TH4_P_MAX_CPG chunks per CU
    const int three = [] { const char *e = getenv("CLV_THRESHOLD_THREE_LAUNCH"); return e ? atoi(e) : 1; }();      // read per call: A/B runs flip it
    if (three && k != 0) {
        const uint32_t cus = (uint32_t)clv_cu_count();
        const uint32_t grid2 = nchunks < cus ? nchunks : cus, cpg2 = (nchunks + grid2 - 1) / grid2, groups2 = (nchunks + cpg2 - 1) / cpg2;
        void *slots = nullptr;
        if (cpg2 <= TH4_P_MAX_CPG && groups2 <= TH4_GROUPS && clv_internal_sync_slots(&slots, CLV_SYNC_SLOT_BYTES_TOTAL, st) == CLV_OK) {
            uint32_t *ctl = (uint32_t *)((char *)slots + CLV_SYNC_SLOT_THRESHOLD_OFFSET);
            uint32_t *cand = (uint32_t *)(cnt + n_pad / 64);
            const uint64_t want1 = (nblocks + 1024 * TH4_K1_U - 1) / (1024 * TH4_K1_U), cap1 = (uint64_t)cus * 2;
            hipLaunchKernelGGL(k_th4_count_hist0, dim3((unsigned)(want1 < cap1 ? want1 : cap1)), dim3(1024), 0, st, (const u32x4 *)q, s, n, cnt, nblocks, ctl);
            int rc = clv_internal_persist_enter(st);                      // resident workgroups that wait for each other: one such launch at a time
            if (rc) {
                (void)hipMemsetAsync(ctl, 0, TH4_CTL_WORDS * sizeof(uint32_t), st);
                return rc;
            }
            const bool in_regs = cpg2 <= 64 && !getenv("CLV_THRESHOLD_FORCE_CAND");      // (the variable: tests run the other form at small sizes)
            unsigned long long *dbg = nullptr;
            if (const char *e = getenv("CLV_THRESHOLD_DEBUG_STAMPS")) dbg = (unsigned long long *)strtoull(e, nullptr, 0);   // probe only: groups x 16 words
            if (in_regs)
                hipLaunchKernelGGL(k_th4_select_persist<true>, dim3(groups2), dim3(1024), 0, st, cnt, s, nblocks, ctl, (uint32_t)k, ts, chunk_ties,
                                   group_ties, nchunks, cpg2, cand, dbg);
            else
                hipLaunchKernelGGL(k_th4_select_persist<false>, dim3(groups2), dim3(1024), 0, st, cnt, s, nblocks, ctl, (uint32_t)k, ts, chunk_ties,
                                   group_ties, nchunks, cpg2, cand, dbg);
            clv_internal_persist_leave();
            hipLaunchKernelGGL(k_th4_apply3, dim3(nchunks), dim3(256), 0, st, (u32x4 *)q, s, n, nblocks, ts, chunk_ties, group_ties, cpg2, ctl,
                               TH4_CTL_WORDS);
            hipError_t e = hipGetLastError();
            if (e != hipSuccess) {
                (void)hipMemsetAsync(ctl, 0, TH4_CTL_WORDS * sizeof(uint32_t), st);     // whatever ran: the next call finds the block clean
                clv_set_error("clv4_threshold: launch failed: %s", hipGetErrorString(e));
                return CLV_ERR_HIP;
            }
            return CLV_OK;
        }
    }
    const uint64_t want = (nblocks + 255) / 256, cap = (uint64_t)clv_cu_count() * 8;
    hipLaunchKernelGGL(k_th4_count6, dim3((unsigned)(want < cap ? want : cap)), dim3(256), 0, st, (const u32x4 *)q, n, cnt, nblocks, hists);
    if (k != 0) {
        // few, fat workgroups: every workgroup clears 16 KiB of bins and flushes its non-empty ones with global atomics, and those land on
        // the SAME few dozen addresses from every workgroup (the keys of one vector cluster), where they serialise at ~50 ns each -- the
        // flush, not the table read, was what a level cost at n = 2^28 (round 3: 1024 workgroups of 256 threads, 38 us per 12-bit level).
        // So: as many THREADS as before but in workgroups of 1024 -- a quarter of the flushes per address -- one per 4096 blocks, at
        // most two per CU.
        // Below 2^20 blocks (n < 2^26) a level is all fixed cost and the round-3 shape -- 256-thread workgroups, one per 1024 blocks --
        // is the faster one (n = 2^24: 6-9 us per level either way, 41 us per call against 49).
        const bool fat = nblocks >= (1ull << 20);
        const uint64_t hwant = fat ? (nblocks + 4095) / 4096 : (nblocks + 1023) / 1024;
        const uint64_t hcap = fat ? (uint64_t)clv_cu_count() * TH4_HWG_PER_CU_X2 / 2 : (uint64_t)clv_cu_count() * 4;
        const dim3 grid((unsigned)(hwant < hcap ? hwant : hcap));
        const dim3 hthreads(fat ? 1024 : 256);
        hipLaunchKernelGGL(k_th4_hist6<0>, grid, hthreads, 0, st, cnt, s, nblocks, hists, (uint32_t)k);
        hipLaunchKernelGGL(k_th4_hist6<1>, grid, hthreads, 0, st, cnt, s, nblocks, hists, (uint32_t)k);
        hipLaunchKernelGGL(k_th4_hist6<2>, grid, hthreads, 0, st, cnt, s, nblocks, hists, (uint32_t)k);
    }
    const uint32_t cpg = (nchunks + TH4_GROUPS - 1) / TH4_GROUPS;
    const uint32_t groups = (nchunks + cpg - 1) / cpg;
    hipLaunchKernelGGL(k_th4_ties6, dim3(groups), dim3(256), 0, st, cnt, s, nblocks, hists, (uint32_t)k, ts, chunk_ties, group_ties, nchunks, cpg);
    hipLaunchKernelGGL(k_th4_apply6, dim3(nchunks), dim3(256), 0, st, (u32x4 *)q, s, n, nblocks, cnt, ts, chunk_ties, group_ties, cpg);
    CLV_LAUNCH_CHECK();
    return CLV_OK;
}

__global__ void k_thresh_state_none(ThreshState *ts) { *ts = ThreshState{0, 0, 0x7F800000u, 0}; }

template <int BITS>
static int threshold_large(uint32_t *q, const float *s, uint64_t n, uint64_t k, void *workspace, hipStream_t st)
{
    uint32_t *hist = (uint32_t *)workspace;
    ThreshState *ts = (ThreshState *)(hist + 4096);
    uint32_t *block_ties = (uint32_t *)((char *)ts + 256);
    const uint64_t nwords = (n + ThreshElems<BITS>::EPW - 1) / ThreshElems<BITS>::EPW;
    const uint32_t nblocks = (uint32_t)((nwords + TH_WORDS_PER_BLOCK - 1) / TH_WORDS_PER_BLOCK);
    CLV_HIP(hipMemsetAsync(hist, 0, 4096 * sizeof(uint32_t) + 256, st));
    if (k == 0) {
        // keep nothing: tau = +inf pattern beyond any finite magnitude, no ties kept (written by a one-thread kernel: the call only enqueues)
        hipLaunchKernelGGL(k_thresh_state_none, dim3(1), dim3(1), 0, st, ts);
    } else {
        const uint64_t want = (nwords + 255) / 256, cap = (uint64_t)clv_cu_count() * 4;
        const dim3 grid((unsigned)(want < cap ? want : cap));
        hipLaunchKernelGGL((k_thresh_hist<0, BITS>), grid, dim3(256), 0, st, (const uint32_t *)q, s, n, ts, hist);
        hipLaunchKernelGGL(k_thresh_select<0>, dim3(1), dim3(256), 0, st, hist, ts, (uint32_t)k);
        hipLaunchKernelGGL((k_thresh_hist<1, BITS>), grid, dim3(256), 0, st, (const uint32_t *)q, s, n, ts, hist);
        hipLaunchKernelGGL(k_thresh_select<1>, dim3(1), dim3(256), 0, st, hist, ts, (uint32_t)k);
        hipLaunchKernelGGL((k_thresh_hist<2, BITS>), grid, dim3(256), 0, st, (const uint32_t *)q, s, n, ts, hist);
        hipLaunchKernelGGL(k_thresh_select<2>, dim3(1), dim3(256), 0, st, hist, ts, (uint32_t)k);
        CLV_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(k_thresh_count_ties<BITS>, dim3(nblocks), dim3(256), 0, st, (const uint32_t *)q, s, n, ts, block_ties);
    hipLaunchKernelGGL(k_thresh_scan, dim3(1), dim3(256), 0, st, block_ties, nblocks);
    hipLaunchKernelGGL(k_thresh_apply<BITS>, dim3(nblocks), dim3(256), 0, st, q, s, n, ts, block_ties);
    CLV_LAUNCH_CHECK();
    return CLV_OK;
}

extern "C" uint64_t clv4_threshold_workspace_bytes(uint64_t n_pad)
{
    const uint64_t chunk_bytes = ((n_pad / 64 / 256 + 1) * 4 + 255) & ~255ull;
    return 3 * 4096 * sizeof(uint32_t) + 256 + TH4_GROUPS * sizeof(uint32_t) + chunk_bytes + (n_pad / 64) * sizeof(unsigned long long) +
           (n_pad / 64) * sizeof(uint32_t) /* the blocks' candidate words */ + 256;
}

extern "C" uint64_t clv8_threshold_workspace_bytes(uint64_t n_pad)
{
    const uint64_t blocks = (n_pad / 4 + TH_WORDS_PER_BLOCK - 1) / TH_WORDS_PER_BLOCK;
    return 4096 * sizeof(uint32_t) + 256 + blocks * sizeof(uint32_t) + 256;
}

extern "C" int clv4_threshold(int8_t *q, const float *s, uint64_t n, uint64_t n_pad, uint64_t k, void *workspace, void *stream)
{
    CLV_REQUIRE(q && s, "clv4_threshold: null pointer");
    CLV_REQUIRE(n_pad % 128 == 0 && n <= n_pad, "clv4_threshold: n=%llu n_pad=%llu", (unsigned long long)n, (unsigned long long)n_pad);
    CLV_REQUIRE(n < (1ull << 32), "clv4_threshold: vectors of 2^32 or more elements are not supported");
    hipStream_t st = as_stream(stream);
    if (k >= n || n == 0) return CLV_OK;                       // everything survives
    if (n_pad <= (uint64_t)TS_THREADS * TS_MAXW * 8) {
        const uint64_t wpt = ((n + 7) / 8 + TS_THREADS - 1) / TS_THREADS;         // words per thread
#define T4_LAUNCH(W) hipLaunchKernelGGL(k_thresh_small<W>, dim3(1), dim3(TS_THREADS), 0, st, (uint32_t *)q, s, (uint32_t)n, (uint32_t)k)
        // (fewer, fatter threads -- W = 8 so that every block is one thread's and the tables need no atomics -- were measured slower at
        //  N = 8192 and 32768: 7.7 / 12.4 us against 5.4 / 10.9)
        if (wpt <= 1) T4_LAUNCH(1);
        else if (wpt <= 2) T4_LAUNCH(2);
        else if (wpt <= 4) T4_LAUNCH(4);
        else if (wpt <= 8) T4_LAUNCH(8);
        else T4_LAUNCH(16);
#undef T4_LAUNCH
        CLV_LAUNCH_CHECK();
        return CLV_OK;
    }
    if (!workspace) {
        int rc = clv_internal_workspace(&workspace, clv4_threshold_workspace_bytes(n_pad), as_stream(stream));
        if (rc) return rc;
    }
    return threshold4_large((uint32_t *)q, s, n, n_pad, k, workspace, st);
}

// =================================================================================================
// f3'  threshold in the REFERENCE's survivor order (CLV_THRESHOLD_REFERENCE): the min-heap walk of CloverVector4.h:1927-1972 /
//      CloverVector8.h:1680-1740 with the heap helpers of CloverBase.h:208-249 (std::make_heap under gt_idx_t = libstdc++'s bottom-up
//      __adjust_heap; min_heapify with left-first ties) reproduced step by step.  Which of several EQUAL magnitudes survive is decided by
//      where they sit in the heap when a larger value arrives, i.e. by the whole history: the walk is sequential by definition, like the
//      16 fma chains of dot EXACT.  One wavefront runs it: the heap lives in LDS (k <= 20000 entries of {value, index}; beyond that its top
//      14 levels, the deeper entries in global memory); the stream of the n - k later elements is taken 64 at a time and a ballot against the current root skips every
//      chunk -- or chunk remainder -- that cannot enter the heap (the root only grows), so only the inserts cost a sift (five heap
//      levels per LDS round trip, see k_thr_ref_walk).  Around it: a parallel pass that writes |value| per element
//      (the same expression as CloverVector4::get / CloverVector8::get) and a parallel pass that clears every nibble / byte whose index
//      is not in the final heap.  Cost: ~0.4 us per insert: N = 8192, K = 1024 0.85 ms (round 4: 3.4).  The C ABI's default is the radix
//      select above; the C++ containers default to this mode (clover_device.h: the exactness switch).
// =================================================================================================
template <int BITS>
__global__ __launch_bounds__(256) void k_thr_ref_keys(const uint32_t *__restrict__ q, const float *__restrict__ s, uint64_t n, float *__restrict__ vals,
                                                      uint32_t *__restrict__ keep, uint64_t keep_words)
{
    typedef ThreshElems<BITS> E;
    const uint64_t nwords = (n + E::EPW - 1) / E::EPW, stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nwords; i += stride) {
        const uint32_t w = q[i];
        const float sc = s[i / E::WPB];
#pragma unroll
        for (int e = 0; e < E::EPW; e++)
            if (i * E::EPW + e < n) vals[i * E::EPW + e] = __uint_as_float(E::key(w, e, sc));
    }
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < keep_words; i += stride) keep[i] = 0;
}

#define THR_LDS_ENTRIES 16384u      // the LDS part of a heap that does not fit: the top 14 levels
#define THR_LDS_WHOLE_MAX 20000u    // largest k whose heap (+ sentinel) lives in LDS whole: 160 008 of the CU's 160 KiB
// the heap's LDS part: the whole heap + its sentinel when k <= THR_LDS_WHOLE_MAX (IN_LDS), else entries 0 .. THR_LDS_ENTRIES - 1 -- the top 14
// levels, where every sift spends two of its rounds -- with the deeper entries in global memory (r5: all of it was global before, 2.3 x slower)
extern __shared__ __attribute__((aligned(16))) uint2 thr_lheap[];

struct ThrHeap {            // heap storage: LDS (ds_read / ds_write), beyond its capacity global memory read past the vector L1
    // global-memory entries: a lane's store and ANOTHER lane's later load of the same entry are ordered by the memory model, not by the
    // in-order issue of one wavefront: release / acquire at wavefront scope (no instruction on gfx950 beyond a wait for the
    // store; the LDS part is ordered by the wave's own lgkmcnt waits)
    template <bool IN_LDS> __device__ static __forceinline__ void wave_fence()
    {
        if (!IN_LDS) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    }
    template <bool IN_LDS> __device__ static __forceinline__ uint2 ld(const uint2 *g, unsigned long long i)
    {
        if (IN_LDS || i < THR_LDS_ENTRIES) return thr_lheap[i];
        const unsigned long long v = __hip_atomic_load((const unsigned long long *)(g + i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return make_uint2((uint32_t)v, (uint32_t)(v >> 32));
    }
    // this lane's own store, no fence (make_heap: a lane works inside its own subtree; wave_fence() separates the levels)
    template <bool IN_LDS> __device__ static __forceinline__ void st1(uint2 *g, unsigned long long i, uint2 v)
    {
        if (IN_LDS || i < THR_LDS_ENTRIES) thr_lheap[i] = v;
        else __hip_atomic_store((unsigned long long *)(g + i), (unsigned long long)v.x | ((unsigned long long)v.y << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // every lane with `on` stores its own entry at its own index (the moves of one sift step); then every lane may read them
    template <bool IN_LDS> __device__ static __forceinline__ void st_lanes(uint2 *g, bool on, unsigned long long i, uint2 v)
    {
        if (on) st1<IN_LDS>(g, i, v);
        wave_fence<IN_LDS>();
    }
    // one wave-uniform entry, written by lane 0
    template <bool IN_LDS> __device__ static __forceinline__ void st(uint2 *g, unsigned long long i, uint2 v) { st_lanes<IN_LDS>(g, threadIdx.x == 0, i, v); }
};
#define THR_VAL(e) __uint_as_float((e).x)
// gt_idx_t (CloverBase.h:216-218): (a.value > b.value) || isnan(a.value).  The NaN clause decides only where a NaN magnitude (a block
// scale that is NaN, or infinite over a zero nibble) sits in the initial heap -- kept so that the walk is the reference's for every input
#define THR_GT(a, b) ((THR_VAL(a) > THR_VAL(b)) || THR_VAL(a) != THR_VAL(a))

// bit `lane` and the bits of its ancestors inside a 62-node subtree laid out as lane = 2^depth - 2 + offset (depth 1 .. 5): the lanes
// that must ALL have been chosen for `lane` to lie on the sift path
__device__ __forceinline__ unsigned long long thr_ancestors(uint32_t lane)
{
    if (lane >= 62) return ~0ull;                                       // never satisfied: only bits 0 .. 61 are ever set
    unsigned long long anc = 0;
    for (uint32_t l = lane;; l = (l - 2) >> 1) {
        anc |= 1ull << l;
        if (l < 2) break;
    }
    return anc;
}

template <bool IN_LDS>
__global__ __launch_bounds__(64) void k_thr_ref_walk(const float *__restrict__ vals, uint32_t n, uint32_t k, uint2 *__restrict__ gheap,
                                                     uint32_t *__restrict__ keep, uint2 *__restrict__ heap_out)
{
    // heap indices: 32 bits while the heap fits LDS (k <= 20000: (pos + 1) << 5 stays small), 64 when part of it is in global memory
    typedef typename std::conditional<IN_LDS, uint32_t, unsigned long long>::type hidx_t;
    uint2 *h = gheap;                                                   // the global part (unused when the heap fits LDS)
    const uint32_t lane = threadIdx.x;
    // "Copy the first K-elements" (CloverVector4.h:1933-1940); entry k is a sentinel (+inf: never smaller than anything) that every
    // fetch beyond the heap is clamped to
    for (uint32_t i = lane; i <= k; i += 64) {
        ThrHeap::st1<IN_LDS>(h, i, i < k ? make_uint2(__float_as_uint(vals[i]), i) : make_uint2(0x7F800000u, 0xFFFFFFFFu));
    }
    ThrHeap::wave_fence<IN_LDS>();
    __syncthreads();
    // std::make_heap(min_heap, min_heap + k, gt_idx_t) (:1944): libstdc++ __make_heap = __adjust_heap(first, parent, len, value) for
    // parent = (len - 2) / 2 ... 0, comp = gt_idx_t.  An adjust touches the subtree under its parent only, and the parents of ONE level have
    // disjoint subtrees: their adjusts commute, so a level is done by all lanes at once (lane = parent, the scalar algorithm per lane) and
    // the levels follow each other bottom-up -- the same heap as the sequential order, entry for entry (r5; one parent at a time before:
    // 0.4 ms at k = 1024).
    if (k >= 2) {
        const uint32_t last_parent = (k - 2) / 2;
        for (int level = 31 - __builtin_clz(last_parent + 1); level >= 0; level--) {
            const uint32_t lo = (1u << level) - 1u, hi = (2u << level) - 2u < last_parent ? (2u << level) - 2u : last_parent;
            for (uint32_t parent = lo + lane; parent <= hi; parent += 64) {
                const uint2 v = ThrHeap::ld<IN_LDS>(h, parent);
                const uint32_t top = parent;
                uint32_t hole = parent, child = parent;
                while (child < (k - 1) / 2) {
                    child = 2 * (child + 1);
                    const uint2 a = ThrHeap::ld<IN_LDS>(h, child);
                    const uint2 b = ThrHeap::ld<IN_LDS>(h, child - 1);
                    if (THR_GT(a, b)) {                                        // comp(first + child, first + (child - 1))
                        child--;
                        ThrHeap::st1<IN_LDS>(h, hole, b);
                    } else {
                        ThrHeap::st1<IN_LDS>(h, hole, a);
                    }
                    hole = child;
                }
                if ((k & 1) == 0 && child == (k - 2) / 2) {
                    child = 2 * (child + 1);
                    ThrHeap::st1<IN_LDS>(h, hole, ThrHeap::ld<IN_LDS>(h, child - 1));
                    hole = child - 1;
                }
                while (hole > top) {                                           // __push_heap
                    const uint32_t par = (hole - 1) / 2;
                    const uint2 pe = ThrHeap::ld<IN_LDS>(h, par);
                    if (!THR_GT(pe, v)) break;                                 // comp(first + parent, value)
                    ThrHeap::st1<IN_LDS>(h, hole, pe);
                    hole = par;
                }
                ThrHeap::st1<IN_LDS>(h, hole, v);
            }
            ThrHeap::wave_fence<IN_LDS>();                                     // the next level reads what other lanes wrote in this one
        }
    }
    // the walk over elements k ... n-1 (:1952-1962): strictly larger than the root -> replace the root, min_heapify(0)
    float root = THR_VAL(ThrHeap::ld<IN_LDS>(h, 0));
    float vnext = (k + lane < n) ? vals[k + lane] : -1.0f;
    // min_heapify(heap, 0, k) with heap[0] = m (CloverBase.h:226-249): the smaller child moves up while it is smaller than m, the LEFT
    // child on equal children.  FIVE levels per memory round trip (r5; one pair of dependent round trips PER level in round 4): the 62
    // descendants of `pos` down to depth 5 are fetched by 62 lanes at once -- lane = 2^depth - 2 + offset, so a left child sits on an even
    // lane with its right sibling beside it; indices beyond the heap are clamped to the +inf sentinel.  What a sibling pair decides depends
    // on the pair and on m only (a < m, b < a, b < m), never on the path: three compares in all lanes give every pair's verdict, a
    // handful of mask operations the set S of chosen children, and a lane lies on the sift path iff it and all its ancestors are in S
    // (one masked compare against a per-lane constant).  The entries on the path move up one level in ONE masked store.  A wave issues
    // one instruction per ~4 cycles whatever it is, so the instruction count of this loop IS its time: ~45 per round.
    const uint32_t ld2 = lane + 2, dep = lane < 62 ? 31u - (uint32_t)__builtin_clz(ld2) : 0u, off = lane < 62 ? ld2 - (1u << dep) : 0u;
    const unsigned long long anc = thr_ancestors(lane);
    const unsigned long long LEFT = 0x1555555555555555ull;                 // even lanes 0 .. 60: the left children
    for (uint32_t base = k; base < n; base += 64) {
        const float v = vnext;
        const uint32_t nb = base + 64 + lane;
        vnext = (base + 64 < n && nb < n) ? vals[nb] : -1.0f;             // |value| >= 0: -1 never enters
        unsigned long long mask = __ballot(v > root);
        while (mask) {
            const int j = __builtin_ctzll(mask);
            const uint2 m = make_uint2((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(v), j), base + (uint32_t)j);
            const float mv = THR_VAL(m);
            hidx_t pos = 0;
            float new_root = mv;
            for (;;) {
                hidx_t mine = ((pos + 1) << dep) - 1 + off;              // lanes 62, 63 fetch `pos` itself: harmless, never on a path
                mine = mine < (hidx_t)k ? mine : (hidx_t)k;
                const uint2 e = ThrHeap::ld<IN_LDS>(h, mine);
                const float a = THR_VAL(e);
                // (__builtin_amdgcn_mov_dpp saves the zero-initialising v_mov of update_dpp -- and runs 5.6 % SLOWER on the same box: 895 against 847 us)
                const float b = __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp(0, (int)e.x, 0xB1, 0xF, 0xF, false));     // lane ^ 1
                const unsigned long long alm = __ballot(a < mv), bla = __ballot(b < a), blm = __ballot(b < mv);
                const unsigned long long r = LEFT & ((alm & bla) | (~alm & blm));      // pairs whose RIGHT child moves up
                const unsigned long long S = (LEFT & alm & ~r) | (r << 1);             // the chosen child of every pair that has one
                const bool on_path = (S & anc) == anc;
                const unsigned long long path = __ballot(on_path);
                if (path == 0) break;
                const int last = 63 - __builtin_clzll(path);
                if (pos == 0) new_root = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)e.x, __builtin_ctzll(path)));
                ThrHeap::st_lanes<IN_LDS>(h, on_path, (mine - 1) >> 1, e);
                if (IN_LDS) pos = (hidx_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)mine, last);
                else pos = (hidx_t)(((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)((unsigned long long)mine >> 32), last) << 32) |
                                    (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)mine, last));
                if (last < 30) break;                                        // the path ended above depth 5: m stops at `pos`
            }
            ThrHeap::st<IN_LDS>(h, pos, m);
            root = new_root;
            mask = __ballot(v > root) & ~((2ull << j) - 1ull);             // later lanes of this chunk, against the new root
        }
    }
    __syncthreads();
    // "Only copy the max K elements" (:1966-1969): the indices left in the heap survive
    for (uint32_t i = lane; i < k; i += 64) {
        const uint2 e = ThrHeap::ld<IN_LDS>(h, i);
        atomicOr(&keep[e.y >> 5], 1u << (e.y & 31));
        if (heap_out) heap_out[i] = e;                                     // threshold_min_heap's caller keeps the heap (CloverVector4.h:1929)
    }
}

template <int BITS>
__global__ __launch_bounds__(256) void k_thr_ref_apply(uint32_t *__restrict__ q, uint64_t n, const uint32_t *__restrict__ keep)
{
    typedef ThreshElems<BITS> E;
    const uint64_t nwords = (n + E::EPW - 1) / E::EPW, stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nwords; i += stride) {
        const uint64_t e0 = i * E::EPW;
        const uint32_t bits = keep[e0 >> 5] >> (e0 & 31);                 // EPW divides 32: the word's bits sit in one bitmap word
        uint32_t w = q[i];
#pragma unroll
        for (int e = 0; e < E::EPW; e++)
            if (e0 + e < n && !((bits >> e) & 1u)) w &= ~E::mask(e);       // setBits(i, 0) for i < length only (:1939, 1961)
        q[i] = w;
    }
}

extern "C" uint64_t clv_threshold_reference_workspace_bytes_k(uint64_t n_pad, uint64_t k)
{
    // [|value| per element: 4 n][survivor bitmap: n / 8][the heap's entries, ONLY when it does not fit LDS (k > 20000): 8 (k + 1)] + slack
    const uint64_t kk = k < n_pad ? k : n_pad;
    return n_pad * 4 + ((n_pad / 8 + 255) & ~255ull) + (kk > THR_LDS_WHOLE_MAX ? (kk + 1) * 8 : 0) + 512;
}
extern "C" uint64_t clv_threshold_reference_workspace_bytes(uint64_t n_pad) { return clv_threshold_reference_workspace_bytes_k(n_pad, n_pad); }   // any k

template <int BITS>
static int threshold_reference(uint32_t *q, const float *s, uint64_t n, uint64_t n_pad, uint64_t k, void *workspace, hipStream_t st,
                               uint2 *heap_out = nullptr)
{
    // the walk steps a 32-bit element index 64 at a time (k_thr_ref_walk): base + 64 must not wrap
    CLV_REQUIRE(n <= 0xFFFFFFFFull - 64, "threshold (reference order): n=%llu, at most 2^32 - 65 elements", (unsigned long long)n);
    if (!workspace) {
        // the library's own scratch is sized for THIS k: the 8-bytes-per-element heap region exists only beyond the LDS heap (ADVICE r5:
        // a 2^30-element vector used to pin 13 GB of grow-only scratch for any k)
        int rc = clv_internal_workspace(&workspace, clv_threshold_reference_workspace_bytes_k(n_pad, k), st);
        if (rc) return rc;
    }
    float *vals = (float *)workspace;
    uint32_t *keep = (uint32_t *)((char *)workspace + n_pad * 4);
    uint2 *gheap = (uint2 *)((char *)keep + ((n_pad / 8 + 255) & ~255ull));
    const uint64_t keep_words = (n + 31) / 32;
    const uint64_t nwords = (n + ThreshElems<BITS>::EPW - 1) / ThreshElems<BITS>::EPW;
    const uint64_t want = (nwords + 255) / 256, cap = (uint64_t)clv_cu_count() * 8;
    const dim3 grid((unsigned)(want < cap ? want : cap));
    hipLaunchKernelGGL(k_thr_ref_keys<BITS>, grid, dim3(256), 0, st, (const uint32_t *)q, s, n, vals, keep, keep_words);
    if (k == 0) {
        // nothing survives (the oracle's reading of k = 0): the bitmap the keys pass has just cleared goes to the apply pass as it is,
        // which clears the first n elements and leaves the padding alone
    } else if (k <= THR_LDS_WHOLE_MAX) {
        const size_t lds = ((size_t)k + 1) * sizeof(uint2);                                  // + the sentinel entry
        if (lds > 64 * 1024) CLV_HIP(hipFuncSetAttribute((const void *)k_thr_ref_walk<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k_thr_ref_walk<true>, dim3(1), dim3(64), lds, st, vals, (uint32_t)n, (uint32_t)k, gheap, keep, heap_out);
    } else {
        const size_t lds = (size_t)THR_LDS_ENTRIES * sizeof(uint2);                          // the top 14 levels; the rest in the workspace
        CLV_HIP(hipFuncSetAttribute((const void *)k_thr_ref_walk<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k_thr_ref_walk<false>, dim3(1), dim3(64), lds, st, vals, (uint32_t)n, (uint32_t)k, gheap, keep, heap_out);
    }
    hipLaunchKernelGGL(k_thr_ref_apply<BITS>, grid, dim3(256), 0, st, q, n, keep);
    CLV_LAUNCH_CHECK();
    return CLV_OK;
}

extern "C" int clv4_threshold_mode(int8_t *q, const float *s, uint64_t n, uint64_t n_pad, uint64_t k, int mode, void *workspace, void *stream)
{
    if (mode == CLV_THRESHOLD_FAST) return clv4_threshold(q, s, n, n_pad, k, workspace, stream);
    CLV_REQUIRE(mode == CLV_THRESHOLD_REFERENCE, "clv4_threshold_mode: unknown mode %d", mode);
    CLV_REQUIRE(q && s, "clv4_threshold_mode: null pointer");
    CLV_REQUIRE(n_pad % 128 == 0 && n <= n_pad, "clv4_threshold_mode: n=%llu n_pad=%llu", (unsigned long long)n, (unsigned long long)n_pad);
    CLV_REQUIRE(n < (1ull << 32), "clv4_threshold_mode: vectors of 2^32 or more elements are not supported");
    if (k >= n || n == 0) return CLV_OK;
    return threshold_reference<4>((uint32_t *)q, s, n, n_pad, k, workspace, as_stream(stream));
}

// threshold_min_heap (CloverVector4.h:1929-1970, CloverVector8.h:1696-1737): the REFERENCE walk, and the K-entry heap as the walk leaves it
template <int BITS>
static int threshold_heap(const char *fn, int8_t *q, const float *s, uint64_t n, uint64_t n_pad, uint64_t k, void *heap_dev, void *workspace, void *stream)
{
    CLV_REQUIRE(q && s && heap_dev, "%s: null pointer", fn);
    CLV_REQUIRE(n_pad % 128 == 0 && n <= n_pad, "%s: n=%llu n_pad=%llu", fn, (unsigned long long)n, (unsigned long long)n_pad);
    CLV_REQUIRE(n < (1ull << 32), "%s: vectors of 2^32 or more elements are not supported", fn);
    // the reference's loop copies the first k elements into the heap unconditionally: k > n reads beyond the vector there -- rejected here
    CLV_REQUIRE(k >= 1 && k <= n, "%s: k=%llu must lie in 1 .. n=%llu", fn, (unsigned long long)k, (unsigned long long)n);
    return threshold_reference<BITS>((uint32_t *)q, s, n, n_pad, k, workspace, as_stream(stream), (uint2 *)heap_dev);
}
extern "C" int clv4_threshold_heap(int8_t *q, const float *s, uint64_t n, uint64_t n_pad, uint64_t k, void *heap_dev, void *workspace, void *stream)
{
    return threshold_heap<4>("clv4_threshold_heap", q, s, n, n_pad, k, heap_dev, workspace, stream);
}
extern "C" int clv8_threshold_heap(int8_t *q, const float *s, uint64_t n, uint64_t n_pad, uint64_t k, void *heap_dev, void *workspace, void *stream)
{
    return threshold_heap<8>("clv8_threshold_heap", q, s, n, n_pad, k, heap_dev, workspace, stream);
}

// CloverVector8::threshold(K) (CloverVector8.h:1680-1740): same algorithm and tie rule on |q * scale / 127|
extern "C" int clv8_threshold(int8_t *q, const float *s, uint64_t n, uint64_t n_pad, uint64_t k, void *workspace, void *stream)
{
    CLV_REQUIRE(q && s, "clv8_threshold: null pointer");
    CLV_REQUIRE(n_pad % 128 == 0 && n <= n_pad, "clv8_threshold: n=%llu n_pad=%llu", (unsigned long long)n, (unsigned long long)n_pad);
    CLV_REQUIRE(n < (1ull << 32), "clv8_threshold: vectors of 2^32 or more elements are not supported");
    hipStream_t st = as_stream(stream);
    if (k >= n || n == 0) return CLV_OK;
    if (n_pad <= (uint64_t)TS_THREADS * TS8_MAXW * 4) {
        const uint64_t w = ((n + 3) / 4 + TS_THREADS - 1) / TS_THREADS;
#define T8_LAUNCH(W) hipLaunchKernelGGL(k_thresh8_small<W>, dim3(1), dim3(TS_THREADS), 0, st, (uint32_t *)q, s, (uint32_t)n, (uint32_t)k)
        if (w <= 1) T8_LAUNCH(1);
        else if (w <= 2) T8_LAUNCH(2);
        else if (w <= 4) T8_LAUNCH(4);
        else T8_LAUNCH(8);
#undef T8_LAUNCH
        CLV_LAUNCH_CHECK();
        return CLV_OK;
    }
    if (!workspace) {
        int rc = clv_internal_workspace(&workspace, clv8_threshold_workspace_bytes(n_pad), as_stream(stream));
        if (rc) return rc;
    }
    return threshold_large<8>((uint32_t *)q, s, n, k, workspace, st);
}

extern "C" int clv8_threshold_mode(int8_t *q, const float *s, uint64_t n, uint64_t n_pad, uint64_t k, int mode, void *workspace, void *stream)
{
    if (mode == CLV_THRESHOLD_FAST) return clv8_threshold(q, s, n, n_pad, k, workspace, stream);
    CLV_REQUIRE(mode == CLV_THRESHOLD_REFERENCE, "clv8_threshold_mode: unknown mode %d", mode);
    CLV_REQUIRE(q && s, "clv8_threshold_mode: null pointer");
    CLV_REQUIRE(n_pad % 128 == 0 && n <= n_pad, "clv8_threshold_mode: n=%llu n_pad=%llu", (unsigned long long)n, (unsigned long long)n_pad);
    CLV_REQUIRE(n < (1ull << 32), "clv8_threshold_mode: vectors of 2^32 or more elements are not supported");
    if (k >= n || n == 0) return CLV_OK;
    return threshold_reference<8>((uint32_t *)q, s, n, n_pad, k, workspace, as_stream(stream));
}
