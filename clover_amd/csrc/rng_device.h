// rng_device.h -- device-side pieces of the XORShift stream shared by rng4.hip, scale_add4.hip, mixed8.hip and matrix4.hip.
//
// A draw never reads part1 (simdxorshift128plus.h:97-109 as written), so each generator lane k is one 64-bit
// word a with  n = T(a), out = n + a, a <- n  and T linear over GF(2).  T^e is applied by square-and-multiply
// over a table of T^(2^k); see rng4.hip for the tables.
//
// State buffer (CLV_RNG_STATE_BYTES = 256 = 32 x u64), two slots so that ONE launch can both read the state in
// every workgroup and write the advanced state, without a grid-wide hand-over:
//     slot i (i = 0, 1) at word 16 i:  [0..3] part1 lanes, [4..7] part2 lanes (the reference's two __m256i keys,
//     CloverRandom.h:90-94), [8] = sequence number of the launch (or clv_rng_set call) that wrote the slot.
// Every launch carries a host-side, process-wide increasing sequence number `seq`.  Readers use the slot with the
// largest stamp below `seq`; workgroup 0 writes the state after the launch's draws into the OTHER slot and stamps
// it with `seq` -- which readers of the same launch ignore whether or not they already see it.  Launches on one
// state must be stream-ordered (they are a sequential stream by definition), and a captured hipGraph would replay
// a stale `seq`: stochastic calls are not graph-capturable.
#pragma once

#include "common.h"

#define RNG_POW_LEVELS 56
#define RNG_SEG_MATS 64          // T^(16*e), e = 0..63: start of 8-block segment e relative to a workgroup base
                                 // (and, in a second table, T^(64*e): 32-block segments of the large-vector shape)
#define RNG_SLOT_WORDS 16        // u64 stride between the two state slots
#define RNG_STAMP_WORD 8         // u64 index of the slot's sequence stamp
#define RNG_TICK_WORD 9          // u64 index (slot 0 only) of the DEVICE-side launch counter: graph mode (clv_rng_graph_mode)

// device tables, one allocation, all in ROW form (bit i of row j = bit j of the image of bit i), which is what the
// whole-wave product below wants: pow_rows[56][64] = T^(2^k), seg_rows[64][64] = T^(16 e)
struct RngTables {
    const uint64_t *pow_rows;
    const uint64_t *seg_rows;
    const uint64_t *seg_rows_long;      // seg_rows_long[64][64] = T^(64 e)
};
int clv_rng_tables(RngTables *t);      // rng4.hip; builds the tables on first use per device
uint64_t clv_rng_next_seq();           // runtime.hip; process-wide launch sequence number (>= 1)
// The sequence number a stochastic launch on `state` carries.  Ordinary states: the next host number.  States in graph mode
// (clv_rng_graph_mode): 0 -- after enqueueing a one-thread kernel that increments the state's own counter (RNG_TICK_WORD) on `stream`;
// the kernels read that counter when they are handed 0 (rng_effective_seq), so a captured graph replays with fresh numbers.
uint64_t clv_rng_seq_for(uint64_t *state, hipStream_t stream);      // runtime.hip

__host__ __device__ __forceinline__ uint64_t xs_T(uint64_t a)
{
    const uint64_t t = a ^ (a << 23);
    return t ^ a ^ (t >> 18) ^ (a >> 5);
}

// r = M * v over GF(2); M is 64 columns (column i = image of bit i).  Per-lane form: every lane its own v.
__host__ __device__ __forceinline__ uint64_t gf2_matvec(const uint64_t *M, uint64_t v)
{
    uint64_t r = 0;
#pragma unroll 32
    for (int i = 0; i < 64; i++) r ^= (0 - ((v >> i) & 1ull)) & M[i];
    return r;
}

#ifdef __HIPCC__
__device__ __forceinline__ uint64_t uniform64(uint64_t v)
{
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return ((uint64_t)hi << 32) | lo;
}

// bit j of M*v for the lane holding row j of M; the ballot of it over the wave is M*v
__device__ __forceinline__ uint64_t wave_matvec(uint64_t row, uint64_t v)
{
    // parity(row & v) = parity((row.lo & v.lo) ^ (row.hi & v.hi)): one popcount instead of two (v_and, v_bitop3, v_bcnt, v_and, v_cmp)
    const uint32_t x = ((uint32_t)row & (uint32_t)v) ^ ((uint32_t)(row >> 32) & (uint32_t)(v >> 32));
    return __ballot(__builtin_popcount(x) & 1);
}
// the same, with the product written into lane DST of (lo, hi): v_writelane takes the ballot straight from its SGPR pair -- no lane
// compare, no selects.  DST is an immediate: a second SGPR beside the ballot would break the one-constant-bus-read rule.
template <int DST>
__device__ __forceinline__ void wave_matvec_to_lane(uint64_t row, uint64_t v, uint32_t &lo, uint32_t &hi)
{
    const uint64_t r = wave_matvec(row, v);
    // gfx940+: a VALU that reads an SGPR needs two wait states behind the VALU that wrote it (here: v_cmp -> VCC); hipcc inserts
    // them for its own instructions, not inside an asm statement
    asm("s_nop 1\n\tv_writelane_b32 %0, %2, %4\n\tv_writelane_b32 %1, %3, %4"
        : "+v"(lo), "+v"(hi)
        : "s"((uint32_t)r), "s"((uint32_t)(r >> 32)), "n"(DST));
}

// Whole-wave form: T^(e << k0)(v) for wave-uniform v and e.  Lane j holds ROW j of the level's matrix, so bit j of
// the product is the parity of (row & v) and the new v is one ballot: ~6 instructions per set bit of e, no
// cross-lane traffic.  The rows of up to 8 levels are fetched together (one memory round trip per 8 bits of e),
// and before v is touched, so that round trip overlaps the load that produces v.
__device__ __forceinline__ uint64_t wave_pow_apply(const uint64_t *__restrict__ pow_rows, uint64_t v, uint64_t e, int k0)
{
    const int lane = threadIdx.x & 63;
    e = uniform64(e);
    const uint64_t *row = pow_rows + (size_t)k0 * 64 + lane;
    uint64_t R[8];
#pragma unroll
    for (int i = 0; i < 8; i++) R[i] = ((e >> i) & 1ull) ? row[64 * i] : 0ull;
    v = uniform64(v);
    for (;;) {
#pragma unroll
        for (int i = 0; i < 8; i++)
            if ((e >> i) & 1ull) v = wave_matvec(R[i], v);
        e >>= 8;
        if (!e) break;
        row += 8 * 64;
#pragma unroll
        for (int i = 0; i < 8; i++) R[i] = ((e >> i) & 1ull) ? row[64 * i] : 0ull;
    }
    return v;
}

// seq == 0: the launch belongs to a state in graph mode; its number is the device counter its tick kernel has just advanced (stable for
// the whole launch: the next tick is stream-ordered behind it)
__device__ __forceinline__ uint64_t rng_effective_seq(const uint64_t *state, uint64_t seq)
{
    return seq ? seq : state[RNG_TICK_WORD];
}

// the slot a launch with sequence number `seq` reads: largest stamp below seq
__device__ __forceinline__ int rng_read_slot(const uint64_t *state, uint64_t seq)
{
    const uint64_t s0 = state[RNG_STAMP_WORD], s1 = state[RNG_SLOT_WORDS + RNG_STAMP_WORD];
    return (s1 < seq && (s0 >= seq || s1 > s0)) ? 1 : 0;
}

// noise lane: byte `sh` of W, as the reference builds it (mask 0x7F7F7F7F, shift left by 8 sh, int -> float, * 2^-31;
// CloverVector4.h:690-734).  Computed without the shift: (V << 8 sh) as an int32 is (V & (0x7F7F7F7F >> 8 sh)) * 2^(8 sh) -- the
// bytes shifted out are dropped, bit 31 is a cleared bit 7 -- and int -> float rounding commutes with a power of two, so
//     noise = (float)(int)(W & (0x7F7F7F7F >> 8 sh)) * 2^(8 sh - 31)
// bit for bit: three VALU (mask and factor are compile-time or per-lane constants) instead of four.
__device__ __forceinline__ float noise_of(uint32_t W, int sh)
{
    const uint32_t mask = 0x7F7F7F7Fu >> (8 * sh);
    const float scale = __uint_as_float((uint32_t)(96 + 8 * sh) << 23);       // 2^(8 sh - 31)
    return (float)(int)(W & mask) * scale;
}

// all four noises of one word (sh = 0..3), for kernels in which a lane uses every shift of its word: the 0x7F mask once for the
// word, the two low fields picked by the conversion itself (SDWA source select) -- 6 VALU + 4 multiplies instead of 8 + 4.
__device__ __forceinline__ void noise4_of(uint32_t W, float n[4])
{
    const uint32_t m = W & 0x7F7F7F7Fu;
    float f2, f3;
    asm("v_cvt_f32_u32_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0" : "=v"(f2) : "v"(m));
    asm("v_cvt_f32_u32_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0" : "=v"(f3) : "v"(m));
    n[0] = (float)(int)m * __uint_as_float(96u << 23);                    // 2^-31
    n[1] = (float)(int)(m & 0x00FFFFFFu) * __uint_as_float(104u << 23);   // 2^-23
    n[2] = f2 * __uint_as_float(112u << 23);                              // 2^-15
    n[3] = f3 * __uint_as_float(120u << 23);                              // 2^-7
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

// workgroup 0 writes the state after `total` (>= 1) draws into the other slot: wave k (0..3) advances generator lane k
// from a0k (part2 = T^total(a0), part1 = T^(total-1)(a0), exactly what `total` sequential draws leave behind).
// Contains a barrier that EVERY workgroup executes: call from uniform control flow.
__device__ __forceinline__ void rng_commit(uint64_t *state, uint64_t seq, int slot, const uint64_t *__restrict__ pow_rows, uint64_t a0k,
                                           uint64_t total)
{
    const int k = threadIdx.x >> 6;
    if (blockIdx.x == 0 && blockIdx.y == 0) {
        uint64_t *next = state + (slot ^ 1) * RNG_SLOT_WORDS;
        const uint64_t f = wave_pow_apply(pow_rows, a0k, total - 1, 0);
        if ((threadIdx.x & 63) == 0) {
            next[k] = f;
            next[4 + k] = xs_T(f);
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            __threadfence();
            next[RNG_STAMP_WORD] = seq;
        }
    } else {
        __syncthreads();
    }
}

// Start-of-kernel step for 256-thread workgroups whose 4 waves map to the 4 generator lanes:
// lds_base[k] = T^(index << shift)(a0[k]), then rng_commit.  Ends with a barrier: call from uniform control flow.
// Returns a0 of this wave's generator lane.
__device__ __forceinline__ uint64_t rng_workgroup_begin(uint64_t *state, uint64_t seq, const uint64_t *__restrict__ pow_rows,
                                                        uint64_t index, int shift, uint64_t total, uint64_t *lds_base)
{
    const int k = threadIdx.x >> 6;
    seq = rng_effective_seq(state, seq);
    const int slot = rng_read_slot(state, seq);
    const uint64_t a0 = state[slot * RNG_SLOT_WORDS + 4 + k];
    const uint64_t b = wave_pow_apply(pow_rows, a0, index, shift);
    if ((threadIdx.x & 63) == 0) lds_base[k] = b;
    rng_commit(state, seq, slot, pow_rows, a0, total);
    return a0;
}

// lane (seg = l>>2, k = l&3) <- SEG[seg0 + seg] * base[k]: the segment starts of one wave as NSEG*4 whole-wave
// products.  Lane j holds row j of each of the wave's NSEG segment matrices (coalesced loads); load them BEFORE
// rng_workgroup_begin so that they share its memory round trip.  All 64 lanes of the wave must take part.
template <int NSEG>
struct SegRows {
    uint64_t R[NSEG];
    __device__ __forceinline__ void load(const uint64_t *__restrict__ seg_rows, int seg0)
    {
        const int lane = threadIdx.x & 63;
#pragma unroll
        for (int s = 0; s < NSEG; s++) R[s] = seg_rows[(size_t)(seg0 + s) * 64 + lane];
    }
    __device__ __forceinline__ uint64_t starts(const uint64_t *lds_base) const
    {
        uint64_t b[4];
#pragma unroll
        for (int k = 0; k < 4; k++) b[k] = uniform64(lds_base[k]);
        return starts_from(b);
    }
    template <int I>
    __device__ __forceinline__ void fill(const uint64_t (&b)[4], uint32_t &lo, uint32_t &hi) const
    {
        if constexpr (I < 4 * NSEG) {
            wave_matvec_to_lane<I>(R[I / 4], b[I % 4], lo, hi);
            fill<I + 1>(b, lo, hi);
        }
    }
    // b[k]: wave-uniform base of generator lane k
    __device__ __forceinline__ uint64_t starts_from(const uint64_t (&b)[4]) const
    {
        uint32_t lo = 0, hi = 0;
        fill<0>(b, lo, hi);
        return ((uint64_t)hi << 32) | lo;
    }
};

// shape of the vector stochastic kernels (quantize, scaleAndAdd): a wave = NSEG segments of SEGLEN consecutive blocks, each walked
// by 4 generator lanes.  S = 1, 4, 16: S segments of 8 blocks (small vectors want many short waves: a wave walks its blocks
// serially).  S = 64: 16 segments of 32 blocks -- the per-wave jump-ahead (64 whole-wave GF(2) products, ~450 VALU) is then
// amortised over 512 blocks instead of 128: it was 16 % of the instructions of the VALU-bound scaleAndAdd.
template <int S>
struct StShape {
    static constexpr bool LONG = S == 64;
    static constexpr int NSEG = LONG ? 16 : S;
    static constexpr int SEGLEN = LONG ? 32 : 8;
    static constexpr int BPR = NSEG == 16 ? 4 : 8;      // blocks per segment per round
    static constexpr int ROUNDS = SEGLEN / BPR;
    static constexpr int NBR = NSEG * BPR;              // blocks per wave per round (<= 64)
    static constexpr int STEPS = NBR / 8;               // 8 blocks (64 lanes x 8 elements) per step
    static constexpr int SHIFT = LONG ? 12 : 6 + (S == 16 ? 4 : S == 4 ? 2 : 0);     // log2(draws per workgroup = 4 waves x NSEG x SEGLEN x 2)
    // global block of (round r, local block bl): segment bl / BPR, block BPR r + bl % BPR inside it
    __device__ static __forceinline__ uint64_t block(uint64_t blk0, int r, int bl) { return blk0 + (uint64_t)(bl / BPR) * SEGLEN + BPR * r + (bl % BPR); }
    __device__ static __forceinline__ const uint64_t *seg_table(const RngTables &T) { return LONG ? T.seg_rows_long : T.seg_rows; }
};

int clv_st_segments(uint64_t nblocks, bool long_ok);     // rng4.hip: S by size (long_ok: may pick the 16 x 32-block shape)
#endif
