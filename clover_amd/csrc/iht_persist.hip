// iht_persist.hip -- Q_IHT / Q_GD (test/performance/01_measure.h:923-946, 999-1021) as ONE persistent launch for the sizes Clover
// publishes (doc/results/performance.txt:564-585: N = 256 ... 8192 fit): Phi and PhiT stay in LDS for all iterations.
//
// Why: an iteration of the launch-per-step loop (iht4.hip) is three launches of 5.5-7.6 us at N = 8192, and Phi / PhiT (16 MiB each)
// evict each other from the 8 x 4 MiB of L2 between launches.  gfx950 has 256 CUs x 160 KiB of LDS = 40 MiB: at N <= 8192 both
// matrices fit ON CHIP.  One launch, one workgroup per CU:
//   * start: workgroup g copies its R1 rows of Phi and R2 rows of PhiT into LDS (once per call);
//   * P1: t1's row dots from LDS -- lane = (row, fma chain j): the reference's 16 chains per row (CloverMatrix4.h:816-898) are 16 lanes,
//     `v_dot8_i32_i4` + cvt + fma per 32-bit word, then the fixed add tree of CloverBase.h:149-157 over the 16 lanes (DPP);
//   * the fp32 dots are PUBLISHED as 8-byte {epoch, bits} granules (one agent-scope store each, data and flag arrive together);
//   * E1: every workgroup gathers ALL m dots (thread = one word of t1: 8 granules, re-polled until their epoch matches) and
//     re-quantises the whole vector for itself: t1 = quantize(d), t2 = quantize(y - t1) (CloverVector4.h:1196-1478) -- the deferred
//     re-quantisation: 64 dots of a row group meet in every consumer instead of in one producer, so no second hop;
//   * P2 / E2: the same with PhiT: t3 = quantize(PhiT t2), x = quantize(x + mu t3), then threshold(K) (FAST: the one-workgroup radix
//     select of threshold4.hip, here on registers, candidates per 8-lane group) -- redundantly in every workgroup, so the new x is
//     already in every workgroup's LDS when the next P1 starts: TWO all-gathers per iteration and no other grid-wide step.
// Bit-exact with the launch-per-step loop: same chains, same tree, same re-quantisation arithmetic (matrix4.hip's epilogue).
// Residency: the grid is <= the CU count and a workgroup takes more than half a CU's LDS or 1024 threads... the host checks the occupancy
// query before launching; every spin is bounded (a trap after 4 s turns a scheduling anomaly into an error, not a hang).
#include "common.h"
#include "thresh_device.h"

#include <mutex>
#include <stdlib.h>

#define IHTP_THREADS 1024
#define IHTP_MAXLEN 8192u          // vectors up to 8192 elements: one 32-bit word (8 elements) per thread

typedef unsigned long long u64;
typedef __attribute__((address_space(1))) u64 gu64;          // granules are touched through GLOBAL agent-scope accesses only, never flat ones

struct IhtpArgs {
    const uint8_t *Phi;
    const float *sPhi;
    const uint8_t *PhiT;
    const float *sPhiT;
    uint32_t m, n, x_len;
    uint32_t R1, R2;              // rows of Phi / PhiT per workgroup (powers of two, <= 64)
    uint32_t *x;
    float *sx;
    const uint32_t *y;
    const float *sy;
    uint32_t *t1;
    float *st1;
    uint32_t *t2;
    float *st2;
    uint32_t *t3;
    float *st3;
    uint32_t iterations, K;
    float mu;
    int threshold;                // 0: Q_GD, 1: Q_IHT with the FAST threshold
    u64 *g1, *g2;                 // granules: m and n of them, zero before the launch
    u64 *dbg;                     // NULL, or 16 wall-clock stamps (100 MHz) per iteration and workgroup for tools/iht_persist_probe.py
};
#define IHTP_STAMP(k) do { if (A.dbg && threadIdx.x == 0 && it < 16) A.dbg[((size_t)g * 16 + it) * 16 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)

// ---- LDS layout (bytes; every offset a multiple of 16) -----------------------------------------------------------------------------
// A row of Phi in LDS is re-dealt so that a chain lane reads FOUR consecutive steps with one ds_read_b128: row word wi = 16 t + j
// (step t, chain j) sits at word (t >> 2) * 64 + j * 4 + (t & 3).  x / t2 use the same dealing; the per-block factors c[b]
// (b = 2 t + a, a = j >> 3) sit at float (t >> 2) * 8 + a * 4 + (t & 3).
struct IhtpLayout {
    uint32_t TG1, TG2;            // step groups (4 steps = 512 columns) of Phi's / PhiT's rows
    uint32_t offA1, offA2, offXV, offC1, offTV, offC2, offP1, offP2, offHist, offWtot, offPub, offStage, total;
};

__host__ __device__ inline IhtpLayout ihtp_layout(uint32_t m, uint32_t n, uint32_t R1, uint32_t R2)
{
    IhtpLayout L;
    L.TG1 = (n / 128 + 3) / 4;
    L.TG2 = (m / 128 + 3) / 4;
    uint32_t o = 0;
    L.offA1 = o; o += R1 * L.TG1 * 256;
    L.offA2 = o; o += R2 * L.TG2 * 256;
    L.offXV = o; o += L.TG1 * 256;
    L.offC1 = o; o += L.TG1 * 32;
    L.offTV = o; o += L.TG2 * 256;
    L.offC2 = o; o += L.TG2 * 32;
    L.offP1 = o; o += L.TG1 * 32;          // f32(sPhi[rg][b] * 1/49), dealt like c
    L.offP2 = o; o += L.TG2 * 32;
    L.offHist = o; o += 4 * 256 * 4;
    L.offWtot = o; o += 64;
    L.offPub = o; o += 64 * 4;                                        // this workgroup's dots on their way to the publishing wave
    L.offStage = o; o += ((m > n ? m : n) < 4096u ? (m > n ? m : n) : 4096u) * 4;      // gathered dots -> owner layout, 4096 per pass
    L.total = o;
    return L;
}

__device__ __forceinline__ uint32_t dealt_word(uint32_t wi) { const uint32_t t = wi >> 4, j = wi & 15; return (t >> 2) * 64 + j * 4 + (t & 3); }
__device__ __forceinline__ uint32_t dealt_factor(uint32_t b) { const uint32_t t = b >> 1, a = b & 1; return (t >> 2) * 8 + a * 4 + (t & 3); }

__device__ __forceinline__ float group8_max(float v)
{
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, false)));       // quad_perm [1,0,3,2]
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, false)));       // quad_perm [2,3,0,1]
    return fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, false)));   // row_half_mirror
}
__device__ __forceinline__ uint32_t group8_add(uint32_t v)
{
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, false);
    return v + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, false);
}

// copy R rows of a row-major nibble matrix (row = T * 64 bytes) into LDS in the dealt layout: work item = (row, step group, chain
// quad jq): four 16-byte loads 64 B apart (steps 4 tg .. 4 tg + 3, chains 4 jq .. 4 jq + 3), a 4 x 4 word transposition in
// registers, four ds_write_b128
__device__ __forceinline__ void ihtp_load_rows(const uint8_t *__restrict__ A, uint64_t row0, uint32_t R, uint32_t T, uint32_t TG, uint32_t *lds)
{
    const uint32_t items = R * TG * 4;
    for (uint32_t it = threadIdx.x; it < items; it += IHTP_THREADS) {
        const uint32_t jq = it & 3, tg = (it >> 2) % TG, r = (it >> 2) / TG;
        const u32x4 *src = reinterpret_cast<const u32x4 *>(A + (row0 + r) * (uint64_t)T * 64);
        u32x4 v[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint32_t t = 4 * tg + i;
            v[i] = t < T ? src[4 * t + jq] : u32x4{0u, 0u, 0u, 0u};
        }
        u32x4 *dst = reinterpret_cast<u32x4 *>(lds + (size_t)(r * TG + tg) * 64 + jq * 16);
        dst[0] = u32x4{v[0].x, v[1].x, v[2].x, v[3].x};
        dst[1] = u32x4{v[0].y, v[1].y, v[2].y, v[3].y};
        dst[2] = u32x4{v[0].z, v[1].z, v[2].z, v[3].z};
        dst[3] = u32x4{v[0].w, v[1].w, v[2].w, v[3].w};
    }
}

// v_dot8_i32_i4 with a zero addend as ONE instruction (the builtin with c = 0 becomes v_mov + v_dot8c)
__device__ __forceinline__ int sdot8z(uint32_t a, uint32_t b)
{
    int r;
    asm("v_dot8_i32_i4 %0, %1, %2, 0" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
#define IHTP_DPP_F(v, ctrl) __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), ctrl, 0xF, 0xF, false))

struct IhtpStepRegs { u32x4 a, x; f32x4 c; };

__device__ __forceinline__ float ihtp_steps4(const IhtpStepRegs &r, float acc)
{
    acc = __builtin_fmaf(r.c.x, (float)sdot8z(r.a.x, r.x.x), acc);
    acc = __builtin_fmaf(r.c.y, (float)sdot8z(r.a.y, r.x.y), acc);
    acc = __builtin_fmaf(r.c.z, (float)sdot8z(r.a.z, r.x.z), acc);
    return __builtin_fmaf(r.c.w, (float)sdot8z(r.a.w, r.x.w), acc);
}

// the row dot of lane (row, chain j) from LDS and the reference's add tree over the row's 16 lanes; every lane of the row returns the dot.
// The step groups are read two ahead of their use (registers double-buffered by hand: a lone wave per SIMD has nobody to hide the LDS
// latency behind; the first version, read-then-use, spent 1.6 us on 64 steps).
__device__ __forceinline__ float ihtp_row_dot(const uint32_t *Arow, const uint32_t *xv, const float *cf, uint32_t T, int j)
{
    const u32x4 *Ap = reinterpret_cast<const u32x4 *>(Arow) + j;
    const u32x4 *Xp = reinterpret_cast<const u32x4 *>(xv) + j;
    const f32x4 *Cp = reinterpret_cast<const f32x4 *>(cf) + (j >> 3);
    float acc = 0.0f;
    const uint32_t full = T >> 2, groups = (T + 3) >> 2, lastg = groups - 1;
#define IHTP_LOAD(R, G) do { const uint32_t g_ = (G) < lastg ? (G) : lastg; R.a = Ap[g_ * 16]; R.x = Xp[g_ * 16]; R.c = Cp[g_ * 2]; } while (0)
    IhtpStepRegs r0, r1, r2, r3;
    IHTP_LOAD(r0, 0);
    IHTP_LOAD(r1, 1);
    uint32_t tg = 0;
    for (; tg + 4 <= full; tg += 4) {
        IHTP_LOAD(r2, tg + 2);
        IHTP_LOAD(r3, tg + 3);
        acc = ihtp_steps4(r0, acc);
        acc = ihtp_steps4(r1, acc);
        IHTP_LOAD(r0, tg + 4);
        IHTP_LOAD(r1, tg + 5);
        acc = ihtp_steps4(r2, acc);
        acc = ihtp_steps4(r3, acc);
    }
    for (; tg < full; tg++) {                                           // r0 = group tg, r1 = group tg + 1
        acc = ihtp_steps4(r0, acc);
        r0 = r1;
        IHTP_LOAD(r1, tg + 2);
    }
    if (T & 3) {                                                       // the last, partial step group (cols % 512 != 0): it is r0
        const uint32_t rem = T & 3;
        acc = __builtin_fmaf(r0.c.x, (float)sdot8z(r0.a.x, r0.x.x), acc);
        if (rem > 1) acc = __builtin_fmaf(r0.c.y, (float)sdot8z(r0.a.y, r0.x.y), acc);
        if (rem > 2) acc = __builtin_fmaf(r0.c.z, (float)sdot8z(r0.a.z, r0.x.z), acc);
    }
#undef IHTP_LOAD
    // chain j: accumulator a = j >> 3, AVX lane w = j & 7.  v[w] = acc[0][w] + acc[1][w]; x[i] = v[i + 4] + v[i]; (x0 + x2) + (x1 + x3).
    // DPP inside the row of 16 lanes: row_ror:8 pairs j with j ^ 8; v is then the same in j and j ^ 8, so row_ror:4 (lane j - 4 mod 16)
    // delivers v[j ^ 4] to every lane; quad_perm for ^ 2 and ^ 1.  fp32 addition commutes, so the bits are the tree's.
    const float v = acc + IHTP_DPP_F(acc, 0x128);
    const float x4 = v + IHTP_DPP_F(v, 0x124);
    const float y2 = x4 + IHTP_DPP_F(x4, 0x4E);
    return y2 + IHTP_DPP_F(y2, 0xB1);
}

// Gather a published vector of `len` dots.  Granule e = {epoch, bits of d[e]} sits at slot e: a producer's R rows are R consecutive
// slots = whole 128-byte lines of its own, written by one wave instruction -- no line is shared between producers.  Thread t polls
// slots t + 1024 k (coalesced, 8 bytes per lane), re-polling what has not arrived, and returns them in v[k] (k < ceil(len / 1024)).
__device__ __forceinline__ void ihtp_gather(const u64 *g, uint32_t t, uint32_t len, uint32_t epoch, float v[8])
{
    uint32_t pending = 0;
#pragma unroll
    for (int k = 0; k < 8; k++)
        if (t + 1024u * k < len) pending |= 1u << k;
    uint32_t spins = 0;
    u64 t_start = 0;
    while (true) {
        u64 raw[8];
#pragma unroll
        for (int k = 0; k < 8; k++)
            if (pending & (1u << k)) raw[k] = __hip_atomic_load((const gu64 *)g + t + 1024u * k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int k = 0; k < 8; k++)
            if ((pending & (1u << k)) && (uint32_t)(raw[k] >> 32) == epoch) {
                v[k] = __uint_as_float((uint32_t)raw[k]);
                pending &= ~(1u << k);
            }
        if (!__any(pending != 0)) break;
        __builtin_amdgcn_s_sleep(2);
        if ((++spins & 1023u) == 0) {                                  // bounded: 4 s on the 100 MHz wall clock, then a trap
            const u64 now = __builtin_amdgcn_s_memrealtime();
            if (!t_start) t_start = now;
            else if (now - t_start > 400000000ull) __builtin_trap();
        }
    }
}

// the gathered dots (thread t holds elements t + 1024 k) -> the owner layout (thread w holds elements 8 w .. 8 w + 7) through the
// 4096-float staging buffer, 4096 elements per pass.  Contains 1 + 2 (passes - 1) workgroup barriers... the caller must have a barrier
// between this call's last read of `stage` and its next write (there always is one).
__device__ __forceinline__ void ihtp_to_owner(const float v[8], uint32_t t, uint32_t len, float *stage, float d[8])
{
    const uint32_t passes = (len + 4095u) / 4096u;
    for (uint32_t p = 0; p < passes; p++) {
        if (p) __syncthreads();                                         // the previous pass has been read
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const float val = p ? v[4 + k] : v[k];
            if (4096u * p + t + 1024u * k < len) stage[t + 1024u * k] = val;
        }
        __syncthreads();
        if ((t >> 9) == p && 8u * t < len) {
            const f32x4 lo = *reinterpret_cast<const f32x4 *>(stage + 8u * (t & 511u)), hi = *reinterpret_cast<const f32x4 *>(stage + 8u * (t & 511u) + 4);
            d[0] = lo.x; d[1] = lo.y; d[2] = lo.z; d[3] = lo.w;
            d[4] = hi.x; d[5] = hi.y; d[6] = hi.z; d[7] = hi.w;
        }
    }
}

// quantise 8 values with the block factor k (rounding disabled: noise 0) -> one word
__device__ __forceinline__ uint32_t ihtp_quant8(const float v[8], float k, int q[8])
{
#pragma unroll
    for (int e = 0; e < 8; e++) q[e] = quant1(v[e], k, 0.0f);
    return pack8_perm(q);
}

// One vector step after a gather, for the thread that owns word w of the vectors (8 lanes = one 64-element block):
//   r = quantize(d)                                (the mvm's re-quantisation, CloverMatrix4.h:919-1080)
//   o = quantize(u + a * r)                        (scaleAndAdd, CloverVector4.h:1196-1478; the arithmetic of matrix4.hip's fused epilogue)
// returns r's word / scale and o's word / scale
__device__ __forceinline__ void ihtp_requant_saa(const float d[8], uint32_t uw, float us, float a, uint32_t &rw, float &rs, uint32_t &ow, float &os)
{
    float mx = 0.0f;
#pragma unroll
    for (int e = 0; e < 8; e++) mx = fmaxf(mx, __builtin_fabsf(d[e]));
    rs = fix_zero_max(group8_max(mx));
    int q[8];
    rw = ihtp_quant8(d, 7.0f / rs, q);
    const float su7 = div7(us), sv7 = div7(rs * a);
    float val[8];
    float m2 = 0.0f;
#pragma unroll
    for (int e = 0; e < 8; e++) {
        val[e] = __builtin_fmaf((float)q[e], sv7, (float)unpack1(uw, e) * su7);
        m2 = fmaxf(m2, __builtin_fabsf(val[e]));
    }
    os = fix_zero_max(group8_max(m2));
    int q2[8];
    ow = ihtp_quant8(val, 7.0f / os, q2);
}

// CloverVector4::threshold(K), FAST rule (threshold4.hip k_thresh_small: same keys, same selection, lowest-index ties), for a vector
// held one word per thread, blocks = groups of 8 lanes.  n = logical length; s = this thread's block scale.  Returns the new word.
// Uses hist[1024] (zero on entry, zero again on exit) and wtot[16]; contains 5 workgroup barriers.
__device__ __forceinline__ uint32_t ihtp_threshold(uint32_t w, float s, uint32_t tid_, uint32_t n, uint32_t k, uint32_t *hist, uint32_t *wtot)
{
    const int tid = (int)tid_, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t first = 8u * tid;
    const uint32_t valid = first >= n ? 0u : (n - first < 8u ? n - first : 8u);
    const float s7 = div7(s);
    uint32_t tau = 0x7F800000u, keep = 0;
    if (k != 0) {
        // magnitude counts of the block: 9 fields of 7 bits summed over the block's 8 lanes (two 32-bit halves; no carries between fields:
        // a field is at most 64 only when it is the only non-zero one -- 7 bits hold 64)
        const u64 cw = th4_count_word(w, valid);
        const uint32_t lo = group8_add((uint32_t)cw), hi = group8_add((uint32_t)(cw >> 32));
        const u64 cnt = ((u64)hi << 32) | lo;
        // candidates (block, magnitude): lane i of the group takes magnitude i, lane 0 also magnitude 8
        const int i = tid & 7;
        const uint32_t wgt0 = (uint32_t)(cnt >> (7 * i)) & 0x7Fu, key0 = cand_key(s7, i);
        const uint32_t wgt1 = i == 0 ? (uint32_t)(cnt >> 56) & 0x7Fu : 0u, key1 = cand_key(s7, 8);
        uint32_t prefix = 0, need = k;
#pragma unroll
        for (int level = 0; level < 4; level++) {
            const int shift = 24 - 8 * level;
            uint32_t *h = hist + 256 * level;
            if (wgt0 && (level == 0 || (key0 >> (shift + 8)) == prefix)) atomicAdd(&h[(key0 >> shift) & 0xFFu], wgt0);
            if (wgt1 && (level == 0 || (key1 >> (shift + 8)) == prefix)) atomicAdd(&h[(key1 >> shift) & 0xFFu], wgt1);
            __syncthreads();
            // every wave selects for itself: lane l owns bins 255 - 4 l ... 252 - 4 l (from the top)
            const u32x4 h4 = *reinterpret_cast<const u32x4 *>(h + 252 - 4 * lane);
            const uint32_t t0 = h4.w, t1 = h4.z, t2 = h4.y, t3 = h4.x;
            const uint32_t sum = t0 + t1 + t2 + t3;
            const uint32_t incl = wave_scan_incl(sum);
            const u64 hit = __ballot(incl >= need && incl - sum < need);
            const int L = __builtin_ctzll(hit);                          // exactly one lane: the level's total weight is >= need
            uint32_t above = (uint32_t)__builtin_amdgcn_readlane((int)(incl - sum), L);      // L is wave-uniform: v_readlane, no LDS round trip
            const uint32_t T0 = (uint32_t)__builtin_amdgcn_readlane((int)t0, L), T1 = (uint32_t)__builtin_amdgcn_readlane((int)t1, L),
                           T2 = (uint32_t)__builtin_amdgcn_readlane((int)t2, L);
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
    // per block: magnitudes >= hi_t are above tau, [lo_t, hi_t) equal it
    uint32_t lo_t = 0, hi_t = 0;
#pragma unroll
    for (int mth = 0; mth <= 8; mth++) {
        const uint32_t key = cand_key(s7, mth);
        lo_t += key < tau;
        hi_t += key <= tau;
    }
    const uint32_t ab = abs_nibbles(swap_nibbles(w));
    const uint32_t vmask = first_nibbles(valid);
    const uint32_t above = ge_nibbles(ab, hi_t);
    uint32_t kb = above | (0x88888888u & ~vmask);                        // padding is left alone
    uint32_t tb = ge_nibbles(ab, lo_t) & ~above & vmask;
    const uint32_t c = __popc(tb);
    const uint32_t v = wave_scan_incl(c);
    if (lane == 63) wtot[wave] = v;
    __syncthreads();
    hist[tid] = 0;                                                       // every wave is past the level scans: ready for the next call
    const uint32_t tot = lane < 16 ? wtot[lane] : 0;
    const uint32_t inc = wave_scan_incl(tot);
    const uint32_t rank = v - c + (uint32_t)__builtin_amdgcn_readlane((int)(inc - tot), wave);
    const uint32_t room = keep > rank ? keep - rank : 0;
    if (room >= c) kb |= tb;
    else for (uint32_t r = 0; r < room; r++) { kb |= tb & (0u - tb); tb &= tb - 1; }
    const uint32_t full = (kb >> 3) * 0xFu;                              // bit 3 -> whole nibble, then back to the stored nibble order
    return w & swap_nibbles(full);
}

__global__ __launch_bounds__(IHTP_THREADS) void k_iht4_persist(const IhtpArgs A)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const IhtpLayout L = ihtp_layout(A.m, A.n, A.R1, A.R2);
    uint32_t *A1 = reinterpret_cast<uint32_t *>(smem + L.offA1), *A2 = reinterpret_cast<uint32_t *>(smem + L.offA2);
    uint32_t *xv = reinterpret_cast<uint32_t *>(smem + L.offXV), *tv = reinterpret_cast<uint32_t *>(smem + L.offTV);
    float *c1 = reinterpret_cast<float *>(smem + L.offC1), *c2 = reinterpret_cast<float *>(smem + L.offC2);
    float *p1 = reinterpret_cast<float *>(smem + L.offP1), *p2 = reinterpret_cast<float *>(smem + L.offP2);
    uint32_t *hist = reinterpret_cast<uint32_t *>(smem + L.offHist), *wtot = reinterpret_cast<uint32_t *>(smem + L.offWtot);
    float *pub = reinterpret_cast<float *>(smem + L.offPub), *stage = reinterpret_cast<float *>(smem + L.offStage);

    const uint32_t tid0 = threadIdx.x, g = blockIdx.x;
    const uint32_t m = A.m, n = A.n, T1 = n / 128, T2 = m / 128;
    const uint32_t row1 = g * A.R1, row2 = g * A.R2;                     // this workgroup's first row of Phi / PhiT
    const bool has1 = row1 < m, has2 = row2 < n;

    // ---- once per call: the matrix slices, the per-block factors' constant halves, y, x = 0 ----
    if (has1) ihtp_load_rows(A.Phi, row1, A.R1, T1, L.TG1, A1);
    if (has2) ihtp_load_rows(A.PhiT, row2, A.R2, T2, L.TG2, A2);
    for (uint32_t i = tid0; i < L.TG1 * 64; i += IHTP_THREADS) xv[i] = 0;                       // x.clear(): nibbles 0 ...
    for (uint32_t i = tid0; i < L.TG1 * 8; i += IHTP_THREADS) { c1[i] = 0.0f; p1[i] = 0.0f; }
    for (uint32_t i = tid0; i < L.TG2 * 8; i += IHTP_THREADS) { c2[i] = 0.0f; p2[i] = 0.0f; }
    for (uint32_t i = tid0; i < L.TG2 * 64; i += IHTP_THREADS) tv[i] = 0;
    hist[tid0] = 0;
    __syncthreads();
    if (has1)
        for (uint32_t b = tid0; b < n / 64; b += IHTP_THREADS) {
            const float p = A.sPhi[(size_t)(row1 >> 6) * (n / 64) + b] * CLV_RCP49;
            p1[dealt_factor(b)] = p;
            c1[dealt_factor(b)] = p * 1.0f;                                                       // ... scales 1.0
        }
    if (has2)
        for (uint32_t b = tid0; b < m / 64; b += IHTP_THREADS) p2[dealt_factor(b)] = A.sPhiT[(size_t)(row2 >> 6) * (m / 64) + b] * CLV_RCP49;
    const bool own_m = tid0 < m / 8, own_n = tid0 < n / 8;                 // this thread owns word tid0 of the m- / n-element vectors
    const uint32_t yw = own_m ? A.y[tid0] : 0u;
    const float ys = own_m ? A.sy[tid0 >> 3] : 1.0f;
    uint32_t xw = 0;
    float xs = 1.0f;
    __syncthreads();

    for (uint32_t it = 0; it < A.iterations; it++) {
        const uint32_t epoch = it + 1;
        // the thread index, opaque to the optimiser once per iteration: everything derived from it (LDS addresses, granule addresses,
        // ownership masks) is recomputed here instead of being hoisted out of the loop and spilled (128 VGPRs at 1024 threads)
        uint32_t tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        const bool last = it + 1 == A.iterations;
        IHTP_STAMP(0);
        // ---- P1: t1's row dots -> LDS -> one wave publishes this workgroup's lines ----
        if (has1 && tid < 16 * A.R1) {
            const float dot = ihtp_row_dot(A1 + (size_t)(tid >> 4) * L.TG1 * 64, xv, c1, T1, tid & 15);
            if ((tid & 15) == 0) pub[tid >> 4] = dot;
        }
        IHTP_STAMP(1);
        __syncthreads();
        if (has1 && tid < A.R1)
            __hip_atomic_store((gu64 *)A.g1 + row1 + tid, ((u64)epoch << 32) | __float_as_uint(pub[tid]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // ---- E1: all of d1 -> t1 = quantize(d1), t2 = quantize(y - t1) ----
        float d[8];
        {
            float v[8];
            ihtp_gather(A.g1, tid, m, epoch, v);
            IHTP_STAMP(2);
            ihtp_to_owner(v, tid, m, stage, d);
        }
        if (tid < m / 8) {
            uint32_t t1w, t2w;
            float t1s, t2s;
            ihtp_requant_saa(d, yw, ys, -1.0f, t1w, t1s, t2w, t2s);
            tv[dealt_word(tid)] = t2w;
            if ((tid & 7) == 0) c2[dealt_factor(tid >> 3)] = p2[dealt_factor(tid >> 3)] * t2s;
            if (last && g == 0) {
                A.t1[tid] = t1w;
                A.t2[tid] = t2w;
                if ((tid & 7) == 0) { A.st1[tid >> 3] = t1s; A.st2[tid >> 3] = t2s; }
            }
        }
        IHTP_STAMP(3);
        __syncthreads();
        IHTP_STAMP(4);
        // ---- P2: t3's row dots ----
        if (has2 && tid < 16 * A.R2) {
            const float dot = ihtp_row_dot(A2 + (size_t)(tid >> 4) * L.TG2 * 64, tv, c2, T2, tid & 15);
            if ((tid & 15) == 0) pub[tid >> 4] = dot;
        }
        IHTP_STAMP(5);
        __syncthreads();
        if (has2 && tid < A.R2)
            __hip_atomic_store((gu64 *)A.g2 + row2 + tid, ((u64)epoch << 32) | __float_as_uint(pub[tid]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // ---- E2: all of d3 -> t3 = quantize(d3), x = quantize(x + mu t3), threshold ----
        {
            float v[8];
            ihtp_gather(A.g2, tid, n, epoch, v);
            IHTP_STAMP(6);
            ihtp_to_owner(v, tid, n, stage, d);
        }
        uint32_t t3w = 0;
        float t3s = 1.0f;
        if (tid < n / 8) ihtp_requant_saa(d, xw, xs, A.mu, t3w, t3s, xw, xs);
        IHTP_STAMP(7);
        if (A.threshold && A.K < A.x_len) xw = ihtp_threshold(xw, xs, tid, A.x_len, A.K, hist, wtot);       // all 1024 threads: it has barriers
        IHTP_STAMP(8);
        if (tid < n / 8) {
            xv[dealt_word(tid)] = xw;
            if ((tid & 7) == 0) c1[dealt_factor(tid >> 3)] = p1[dealt_factor(tid >> 3)] * xs;
            if (last && g == 0) {
                A.t3[tid] = t3w;
                A.x[tid] = xw;
                if ((tid & 7) == 0) { A.st3[tid >> 3] = t3s; A.sx[tid >> 3] = xs; }
            }
        }
        __syncthreads();
        IHTP_STAMP(9);
    }
}

// ---- host ---------------------------------------------------------------------------------------------------------------------------
static uint32_t pow2_ceil(uint32_t v) { uint32_t p = 1; while (p < v) p <<= 1; return p; }

// Persistent launches need every workgroup resident at once: two of them on one device at the same time (two streams) could each hold
// part of the chip and wait for the rest forever.  They are therefore chained: a launch on another stream than the previous one waits
// for an event recorded behind that one.  (Another PROCESS on the same device is not covered: the kernel's bounded spin then traps.)
static std::mutex g_persist_mutex;
static struct { bool valid; hipStream_t stream; hipEvent_t ev; } g_persist_last[64];

void clv_internal_persist_forget(hipStream_t stream)
{
    std::lock_guard<std::mutex> lock(g_persist_mutex);
    for (auto &e : g_persist_last)
        if (e.valid && e.stream == stream && stream != nullptr) e.valid = false;
}

static int persist_chain(hipStream_t st)
{
    int dev = 0;
    CLV_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) return CLV_OK;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess) { (void)hipGetLastError(); cs = hipStreamCaptureStatusNone; }
    if (cs != hipStreamCaptureStatusNone) return CLV_OK;                  // inside a capture the graph's own edges order the launches
    std::lock_guard<std::mutex> lock(g_persist_mutex);
    auto &e = g_persist_last[dev];
    if (e.valid && e.stream != st) {
        if (!e.ev) CLV_HIP(hipEventCreateWithFlags(&e.ev, hipEventDisableTiming));
        CLV_HIP(hipEventRecord(e.ev, e.stream));
        CLV_HIP(hipStreamWaitEvent(st, e.ev, 0));
    }
    e.valid = true;
    e.stream = st;
    return CLV_OK;
}

// returns 1 if the persistent kernel was launched, 0 if the problem does not qualify (the caller runs the launch-per-step loop), < 0 on error
int clm4_iht_persistent(const int8_t *Phi, const float *sPhi, const int8_t *PhiT, const float *sPhiT, uint64_t m, uint64_t n, int8_t *x,
                        float *sx, uint64_t x_len, const int8_t *y, const float *sy, int8_t *t1, float *st1, int8_t *t2, float *st2, int8_t *t3,
                        float *st3, uint64_t iterations, uint64_t K, float mu, int threshold, uint64_t *rng, hipStream_t st)
{
    const int mode = [] { const char *e = getenv("CLV_IHT_PERSISTENT"); return e ? atoi(e) : 1; }();      // read per call: A/B runs flip it
    if (!mode || rng || threshold < 0 || threshold > 1 || !iterations || iterations >= 0x7FFFFFFFull) return 0;
    if (m > IHTP_MAXLEN || n > IHTP_MAXLEN || m % 128 || n % 128 || !m || !n) return 0;
    const int cus = clv_cu_count();
    const uint32_t rows_min = [] { const char *e = getenv("CLV_IHT_ROWS_MIN"); const int v = e ? atoi(e) : 4; return (uint32_t)(v < 1 ? 1 : v > 64 ? 64 : v); }();
    uint32_t R1 = pow2_ceil((uint32_t)((m + cus - 1) / cus)), R2 = pow2_ceil((uint32_t)((n + cus - 1) / cus));
    if (R1 < rows_min) R1 = pow2_ceil(rows_min);
    if (R2 < rows_min) R2 = pow2_ceil(rows_min);
    if (R1 > 64 || R2 > 64) return 0;
    const IhtpLayout L = ihtp_layout((uint32_t)m, (uint32_t)n, R1, R2);
    if (L.total > 160u * 1024u) return 0;
    const uint32_t grid = (uint32_t)((m / R1 > n / R2) ? (m + R1 - 1) / R1 : (n + R2 - 1) / R2);
    if ((int)grid > cus) return 0;

    static std::mutex attr_mutex;
    static uint32_t attr_set[64];
    int dev = 0;
    CLV_HIP(hipGetDevice(&dev));
    {
        std::lock_guard<std::mutex> lock(attr_mutex);
        if (dev >= 0 && dev < 64 && attr_set[dev] < L.total) {
            CLV_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_iht4_persist), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            attr_set[dev] = 160u * 1024u;
        }
    }
    void *ws = nullptr;
    int rc = clv_internal_workspace(&ws, (m + n) * sizeof(u64), st);
    if (rc) return -1;
    if (hipMemsetAsync(ws, 0, (m + n) * sizeof(u64), st) != hipSuccess) {
        clv_set_error("clm4_iht: hipMemsetAsync failed: %s", hipGetErrorString(hipGetLastError()));
        return -1;
    }
    if (persist_chain(st)) return -1;
    IhtpArgs a;
    a.Phi = (const uint8_t *)Phi; a.sPhi = sPhi; a.PhiT = (const uint8_t *)PhiT; a.sPhiT = sPhiT;
    a.m = (uint32_t)m; a.n = (uint32_t)n; a.x_len = (uint32_t)x_len; a.R1 = R1; a.R2 = R2;
    a.x = (uint32_t *)x; a.sx = sx; a.y = (const uint32_t *)y; a.sy = sy;
    a.t1 = (uint32_t *)t1; a.st1 = st1; a.t2 = (uint32_t *)t2; a.st2 = st2; a.t3 = (uint32_t *)t3; a.st3 = st3;
    a.iterations = (uint32_t)iterations; a.K = (uint32_t)(K > 0xFFFFFFFFull ? 0xFFFFFFFFull : K); a.mu = mu; a.threshold = threshold;
    a.g1 = (u64 *)ws; a.g2 = (u64 *)ws + m;
    a.dbg = nullptr;
    if (const char *e = getenv("CLV_IHT_DEBUG_STAMPS")) a.dbg = (u64 *)strtoull(e, nullptr, 0);      // probe only: a device buffer of grid * 16 * 16 words
    hipLaunchKernelGGL(k_iht4_persist, dim3(grid), dim3(IHTP_THREADS), L.total, st, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        clv_set_error("clm4_iht: persistent launch failed: %s", hipGetErrorString(e));
        return -1;
    }
    return 1;
}
