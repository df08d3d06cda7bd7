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
#include "rng_device.h"
#include "thresh_device.h"

#include <atomic>
#include <mutex>
#include <stdlib.h>
#include <string.h>

#define IHTP_THREADS 1024
#define IHTP_MAXLEN 8192u          // vectors up to 8192 elements: one 32-bit word (8 elements) per thread

typedef uint64_t u64;
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
    // stochastic rounding (k_iht4_persist<true>): the XORShift state (rng_device.h), this launch's sequence number, T^(16 e) in row form,
    // and the per-iteration jumps T^(D - 16), T^(D - 17) in column form (D = draws per iteration = 4 (m + n) / 64)
    u64 *rng;
    u64 seq;
    const u64 *seg_rows;
    u64 jump[64], jump1[64];
    uint32_t nap0, nap;           // s_sleep(1) units before the first poll round of a gather / between two rounds
    u64 *dbg;                     // NULL, or 16 wall-clock stamps (100 MHz) per iteration and workgroup for tools/iht_persist_probe.py
};
#define IHTP_STAMP(k) do { if (A.dbg && (threadIdx.x == 0 || threadIdx.x == IHTP_THREADS - 1) && it < 16) A.dbg[((size_t)g * 16 + it) * 32 + (threadIdx.x ? 16 : 0) + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)

// ---- who owns which rows ------------------------------------------------------------------------------------------------------------
// A consumer thread owns ONE WORD of a vector (elements 8 w .. 8 w + 7) and wants its 8 dots with coalesced 16-byte loads (two granules
// each): element e = 8 w + k lives in granule slot (len / 4) * (k >> 1) + 2 w + (k & 1).  A producer wants whole 128-byte lines of its
// own (16 slots: a line shared between producers is invalidated under the pollers once per producer -- the first version of this kernel,
// natural row order: 6-9 us per gather at N = 8192; this dealing: 1.5-1.9).  Both hold when a workgroup owns "units": unit u =
// (a = u >> 2, kk = u & 3) = the 16 rows 64 a + 8 j + 2 kk + i, j = 0..7, i = 0..1 = slots (len / 4) kk + 16 a + 2 j + i.  A workgroup
// takes R / 16 consecutive units (R = 16, 32 or 64 rows); they share `a`: all its rows lie in the 64-row group a.
__device__ __forceinline__ uint32_t unit_row(uint32_t u, uint32_t l) { return 64u * (u >> 2) + 8u * (l >> 1) + 2u * (u & 3u) + (l & 1u); }
__device__ __forceinline__ uint32_t unit_slot(uint32_t u, uint32_t l, uint32_t len) { return (len >> 2) * (u & 3u) + 16u * (u >> 2) + l; }

// ---- LDS layout (bytes; every offset a multiple of 16) -----------------------------------------------------------------------------
// A row of Phi in LDS is re-dealt so that a chain lane reads FOUR consecutive steps with one ds_read_b128: row word wi = 16 t + j
// (step t, chain j) sits at word (t >> 2) * 64 + j * 4 + (t & 3).  x / t2 use the same dealing; the per-block factors c[b]
// (b = 2 t + a, a = j >> 3) sit at float (t >> 2) * 8 + a * 4 + (t & 3).
struct IhtpLayout {
    uint32_t TG1, TG2;            // step groups (4 steps = 512 columns) of Phi's / PhiT's rows
    uint32_t offA1, offA2, offXV, offC1, offTV, offC2, offP1, offP2, offHist, offWtot, offSel, offPub, offLut, offRaw, total;
};

__host__ __device__ inline IhtpLayout ihtp_layout(uint32_t m, uint32_t n, uint32_t R1, uint32_t R2, bool st)
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
    L.offSel = o; o += 64;                 // the scanning wave's (prefix, need) per radix level
    L.offPub = o; o += 64 * 4;             // this workgroup's dots on their way to the publishing wave
    L.offLut = o; o += 256 * 8;            // byte -> magnitude counts of its two nibbles (9 fields of 7 bits)
    // stochastic: the raw draws of one vector phase (4 per block: 2 for the mvm's re-quantisation, 2 for the scaleAndAdd), generated in
    // segments of 16 draws; the two phases of an iteration use the buffer one after the other
    const uint32_t gmax = (m > n ? m : n) / 64;
    L.offRaw = o; o += st ? ((gmax + 3) / 4) * 16 * 32 : 0;
    L.total = o;                           // (the row-dot loop reads, never uses, up to two step groups past an array's end: all of them
                                           //  have other arrays behind them)
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
// the same for the 64-bit magnitude tables: a 7-bit field straddles bit 32, so the halves must be added WITH their carry
__device__ __forceinline__ u64 group8_add64(u64 v)
{
#define IHTP_DPP64(x, ctrl) (((u64)(uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)((x) >> 32), ctrl, 0xF, 0xF, false) << 32) | \
                             (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(x), ctrl, 0xF, 0xF, false))
    v += IHTP_DPP64(v, 0xB1);
    v += IHTP_DPP64(v, 0x4E);
    return v + IHTP_DPP64(v, 0x141);
#undef IHTP_DPP64
}
__device__ __forceinline__ uint32_t group8_add(uint32_t v)
{
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, false);
    return v + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, false);
}

// copy this workgroup's R rows (units u0 .., 16 rows each) of a row-major nibble matrix (row = T * 64 bytes) into LDS in the dealt layout:
// work item = (local row, step group, chain quad jq): four 16-byte loads 64 B apart (steps 4 tg .. 4 tg + 3, chains 4 jq .. 4 jq + 3),
// a 4 x 4 word transposition in registers, four ds_write_b128
__device__ __forceinline__ void ihtp_load_rows(const uint8_t *__restrict__ A, uint32_t u0, uint32_t R, uint32_t T, uint32_t TG, uint32_t *lds)
{
    const uint32_t items = R * TG * 4;
    for (uint32_t it = threadIdx.x; it < items; it += IHTP_THREADS) {
        const uint32_t jq = it & 3, tg = (it >> 2) % TG, lr = (it >> 2) / TG;
        const uint64_t row = unit_row(u0 + (lr >> 4), lr & 15);
        const u32x4 *src = reinterpret_cast<const u32x4 *>(A + row * (uint64_t)T * 64);
        u32x4 v[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint32_t t = 4 * tg + i;
            v[i] = t < T ? src[4 * t + jq] : u32x4{0u, 0u, 0u, 0u};
        }
        u32x4 *dst = reinterpret_cast<u32x4 *>(lds + (size_t)(lr * TG + tg) * 64 + jq * 16);
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

// the four exact word integers of a step group as floats (order-free work: it is interleaved with the previous group's fma chain below)
__device__ __forceinline__ f32x4 ihtp_ints(const IhtpStepRegs &r)
{
    return f32x4{(float)sdot8z(r.a.x, r.x.x), (float)sdot8z(r.a.y, r.x.y), (float)sdot8z(r.a.z, r.x.z), (float)sdot8z(r.a.w, r.x.w)};
}
// acc through the four steps of group `cur` (its integers in ic) WHILE the next group's integers are formed: the fma chain is the only
// dependent sequence of the loop, and a lone wave on its SIMD needs independent instructions between two links of it
__device__ __forceinline__ float ihtp_chain4(float acc, const f32x4 c, const f32x4 ic, const IhtpStepRegs &nx, f32x4 &in)
{
    const int n0 = sdot8z(nx.a.x, nx.x.x);
    acc = __builtin_fmaf(c.x, ic.x, acc);
    const int n1 = sdot8z(nx.a.y, nx.x.y);
    in.x = (float)n0;
    acc = __builtin_fmaf(c.y, ic.y, acc);
    const int n2 = sdot8z(nx.a.z, nx.x.z);
    in.y = (float)n1;
    acc = __builtin_fmaf(c.z, ic.z, acc);
    const int n3 = sdot8z(nx.a.w, nx.x.w);
    in.z = (float)n2;
    acc = __builtin_fmaf(c.w, ic.w, acc);
    in.w = (float)n3;
    return acc;
}

// the row dot of lane (row, chain j) from LDS and the reference's add tree over the row's 16 lanes; every lane of the row returns the dot.
// Step groups are read two ahead of their use (registers double-buffered by hand: a lone wave per SIMD has nobody to hide the LDS latency
// behind; the first version, read-then-use, spent 1.6 us on 64 steps).  Reads run up to two groups past the row's end (inside LDS, unused).
__device__ __forceinline__ float ihtp_row_dot(const uint32_t *Arow, const uint32_t *xv, const float *cf, uint32_t T, int j)
{
    const u32x4 *Ap = reinterpret_cast<const u32x4 *>(Arow) + j;
    const u32x4 *Xp = reinterpret_cast<const u32x4 *>(xv) + j;
    const f32x4 *Cp = reinterpret_cast<const f32x4 *>(cf) + (j >> 3);
    float acc = 0.0f;
    const uint32_t full = T >> 2;
#define IHTP_LOAD(R, G) do { R.a = Ap[(G) * 16]; R.x = Xp[(G) * 16]; R.c = Cp[(G) * 2]; } while (0)
    IhtpStepRegs r0, r1;
    IHTP_LOAD(r0, 0);
    IHTP_LOAD(r1, 1);
    f32x4 i0 = ihtp_ints(r0), i1;
    uint32_t tg = 0;
    for (; tg + 2 <= full; tg += 2) {                                   // r0 = group tg (integers in i0), r1 = group tg + 1
        const f32x4 c0 = r0.c;
        IHTP_LOAD(r0, tg + 2);
        acc = ihtp_chain4(acc, c0, i0, r1, i1);
        const f32x4 c1 = r1.c;
        IHTP_LOAD(r1, tg + 3);
        acc = ihtp_chain4(acc, c1, i1, r0, i0);
    }
    if (tg < full) {                                                   // one more full group: r0; the partial one, if any, is then r1
        acc = ihtp_chain4(acc, r0.c, i0, r1, i1);
        r0 = r1;
        i0 = i1;
    }
    if (T & 3) {                                                       // the last, partial step group (cols % 512 != 0): it is r0 / i0
        const uint32_t rem = T & 3;
        acc = __builtin_fmaf(r0.c.x, i0.x, acc);
        if (rem > 1) acc = __builtin_fmaf(r0.c.y, i0.y, acc);
        if (rem > 2) acc = __builtin_fmaf(r0.c.z, i0.z, acc);
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

// gather word w's 8 dots: granules {epoch, bits of d[8 w + k]} sit at slots (len / 4) * (k >> 1) + 2 w + (k & 1): four coalesced 16-byte
// sc1 loads (two granules each) through a buffer descriptor, re-polled until every lane of the wave has everything.  A poll round moves
// the whole vector into every CU (64 KiB at n = 8192: 16 MiB chip-wide), so rounds are not free: the first one is issued `nap0` sleeps
// after the workgroup's own publication -- about when the slowest producer's lines land -- and a missed one is followed after `nap`.
// (Two rounds in flight half a round trip apart, to cut the quantisation of the waiting time, measured SLOWER: 14.0 against 10.8 us per
// iteration -- the polls are the traffic.)
__device__ __forceinline__ void ihtp_gather8(const u64 *g, uint32_t len, uint32_t w, uint32_t epoch, bool active, uint32_t nap0, uint32_t nap, float d[8])
{
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<u64 *>(g), 0, (int)(len * 8u), 0x00020000);
    const uint32_t off = active ? 16u * w : 0u, region = 2u * len;     // bytes: slot 2 w; (len / 4) slots per kk
    uint32_t pending = active ? 0xFu : 0u;
    uint32_t spins = 0;
    u64 t_start = 0;
    for (uint32_t z = 0; z < nap0; z++) __builtin_amdgcn_s_sleep(1);
    while (true) {
        u32x4 v[4];
#pragma unroll
        for (int kk = 0; kk < 4; kk++)
            if (pending & (1u << kk)) v[kk] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off + region * kk, 0, /*sc1*/ 16);
#pragma unroll
        for (int kk = 0; kk < 4; kk++)
            if ((pending & (1u << kk)) && v[kk].y == epoch && v[kk].w == epoch) {
                d[2 * kk] = __uint_as_float(v[kk].x);
                d[2 * kk + 1] = __uint_as_float(v[kk].z);
                pending &= ~(1u << kk);
            }
        if (!__any(pending != 0)) break;
        for (uint32_t z = 0; z < nap; z++) __builtin_amdgcn_s_sleep(1);
        if ((++spins & 1023u) == 0) {                                  // bounded: 4 s on the 100 MHz wall clock, then a trap
            const u64 now = __builtin_amdgcn_s_memrealtime();
            if (!t_start) t_start = now;
            else if (now - t_start > 400000000ull) __builtin_trap();
        }
    }
}

// One vector step after a gather, for the thread that owns word w of the vectors (8 lanes = one 64-element block):
//   r = quantize(d)                                (the mvm's re-quantisation, CloverMatrix4.h:919-1080)
//   o = quantize(u + a * r)                        (scaleAndAdd, CloverVector4.h:1196-1478; the arithmetic of matrix4.hip's fused epilogue)
// returns r's word / scale and o's word / scale.  ST: W[0..1] = this word's dword of the block's two mvm draws, W[2..3] = of its two
// scaleAndAdd draws: element e = 8 i + e' of the block takes group e' of lane i -- draw e' >> 2, byte e' & 3 -- in the mvm (lane map
// 8 j + g, CloverMatrix4.h:925-932) and group e' ^ 1 in scaleAndAdd (8 j + (g ^ 1), CloverVector4.h:1236-1243).
template <bool ST>
__device__ __forceinline__ void ihtp_requant_saa(const float d[8], uint32_t uw, float us, float a, const uint32_t W[4], uint32_t &rw, float &rs,
                                                 uint32_t &ow, float &os)
{
    float nm[8], ns[8];
#pragma unroll
    for (int e = 0; e < 8; e++) nm[e] = ns[e] = 0.0f;
    if (ST) {
        float t[4];
        noise4_of(W[0], nm);
        noise4_of(W[1], nm + 4);
        noise4_of(W[2], t);
        ns[0] = t[1]; ns[1] = t[0]; ns[2] = t[3]; ns[3] = t[2];
        noise4_of(W[3], t);
        ns[4] = t[1]; ns[5] = t[0]; ns[6] = t[3]; ns[7] = t[2];
    }
    float mx = 0.0f;
#pragma unroll
    for (int e = 0; e < 8; e++) mx = fmaxf(mx, __builtin_fabsf(d[e]));
    rs = fix_zero_max(group8_max(mx));
    int q[8];
    {
        const float k = 7.0f / rs;
#pragma unroll
        for (int e = 0; e < 8; e++) q[e] = quant1(d[e], k, nm[e]);
        rw = pack8_perm(q);
    }
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
    {
        const float k = 7.0f / os;
#pragma unroll
        for (int e = 0; e < 8; e++) q2[e] = quant1(val[e], k, ns[e]);
        ow = pack8_perm(q2);
    }
}

// ---- stochastic rounding: the generator lanes ----
// v -> M v for a matrix in ROW form in global memory (bit j of the result = parity of row j & v): the one-off jump to a segment start
__device__ __forceinline__ u64 ihtp_rows_matvec(const u64 *__restrict__ rows, u64 v)
{
    u64 r = 0;
    for (int j = 0; j < 64; j++) r |= (u64)(__builtin_popcountll(rows[j] & v) & 1) << j;
    return r;
}
// v -> M v for a matrix in COLUMN form held in the kernel arguments (scalar loads): the per-iteration jump, 3 VALU per column
__device__ __forceinline__ u64 ihtp_cols_matvec(const u64 (&cols)[64], u64 v)
{
    uint32_t lo = 0, hi = 0;
#pragma unroll 16
    for (int i = 0; i < 64; i++) {
        const uint32_t mask = 0u - (uint32_t)((v >> i) & 1ull);
        lo ^= mask & (uint32_t)cols[i];
        hi ^= mask & (uint32_t)(cols[i] >> 32);
    }
    return ((u64)hi << 32) | lo;
}
// 16 draws of generator lane k from state a: raw[(16 seg + s) * 4 + k] = the 64-bit output (dwords W[2k], W[2k+1] of draw 16 seg + s)
__device__ __forceinline__ u64 ihtp_gen16(u64 a, u64 *raw, uint32_t seg, uint32_t k)
{
#pragma unroll
    for (int s_ = 0; s_ < 16; s_++) {
        const u64 n = xs_T(a);
        raw[(16 * seg + s_) * 4 + k] = n + a;
        a = n;
    }
    return a;
}

// byte -> the magnitude counts of its two nibbles, th4_count_word's field layout (9 fields of 7 bits)
__device__ __forceinline__ u64 ihtp_lut_entry(uint32_t b)
{
    const uint32_t h = b >> 4, l = b & 15u;
    const uint32_t mh = h < 8 ? h : 16u - h, ml = l < 8 ? l : 16u - l;
    return (1ull << (7u * mh)) + (1ull << (7u * ml));
}

// CloverVector4::threshold(K), FAST rule (threshold4.hip k_thresh_small: same keys, same selection, lowest-index ties), for a vector
// held one word per thread, blocks = groups of 8 lanes.  n = logical length; s = this thread's block scale.  Returns the new word.
// All 1024 threads redo this in every workgroup, so instructions and LDS round trips count: the magnitude counts of a word come from a
// byte table, the 256 bins of a radix level are scanned by ONE wave (16 waves on 4 SIMDs share the issue slots; the others wait for
// (prefix, need) at a barrier), the per-block cut-offs come from the block's own candidate keys.  Measured and dropped (profiles/
// r06_iht_persist_notes.txt): every wave scanning for itself (same time); 9-bit levels below the candidates' common leading bits with the
// bins in 4 copies (3 levels instead of 4, but a level's scan + clear by one wave costs 0.68 us against 0.42).
// LDS: hist[4 * 256] zero on entry and again on exit; wtot[16], sel[8], lut[256].  9 workgroup barriers.
__device__ __forceinline__ uint32_t ihtp_threshold(uint32_t w, float s, uint32_t tid_, uint32_t n, uint32_t k, uint32_t *hist, uint32_t *wtot,
                                                   uint32_t *sel, const u64 *lut, u64 *dbg)
{
#define THR_STAMP(k_) do { if (dbg && threadIdx.x == 0) dbg[k_] = __builtin_amdgcn_s_memrealtime(); } while (0)
    const int tid = (int)tid_, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t first = 8u * tid;
    const uint32_t valid = first >= n ? 0u : (n - first < 8u ? n - first : 8u);
    const float s7 = div7(s);
    const int i = tid & 7;
    // candidates (block, magnitude): lane i of the group takes magnitude i, lane 0 also magnitude 8
    const uint32_t key0 = cand_key(s7, i), key1 = cand_key(s7, 8);
    uint32_t tau = 0x7F800000u, keep = 0;
    if (k != 0) {
        // magnitude counts of the block: 9 fields of 7 bits summed over the block's 8 lanes (no carries between fields: a field is at
        // most 64, and 7 bits hold 64; the field of magnitude 4 straddles bit 32: a true 64-bit addition)
        u64 cw;
        if (valid == 8) cw = (lut[w & 0xFFu] + lut[(w >> 8) & 0xFFu]) + (lut[(w >> 16) & 0xFFu] + lut[w >> 24]);
        else cw = th4_count_word(w, valid);
        const u64 cnt = group8_add64(cw);
        const uint32_t wgt0 = (uint32_t)(cnt >> (7 * i)) & 0x7Fu;
        const uint32_t wgt1 = i == 0 ? (uint32_t)(cnt >> 56) & 0x7Fu : 0u;
        uint32_t prefix = 0, need = k;
        THR_STAMP(10);
#pragma unroll
        for (int level = 0; level < 4; level++) {
            const int shift = 24 - 8 * level;
            uint32_t *h = hist + 256 * level;
            if (wgt0 && (level == 0 || (key0 >> (shift + 8)) == prefix)) atomicAdd(&h[(key0 >> shift) & 0xFFu], wgt0);
            if (wgt1 && (level == 0 || (key1 >> (shift + 8)) == prefix)) atomicAdd(&h[(key1 >> shift) & 0xFFu], wgt1);
            __syncthreads();
            THR_STAMP(11 + level);
            if (wave == 0) {
                // lane l owns bins 255 - 4 l ... 252 - 4 l (from the top)
                const u32x4 h4 = *reinterpret_cast<const u32x4 *>(h + 252 - 4 * lane);
                const uint32_t t0 = h4.w, t1 = h4.z, t2 = h4.y, t3 = h4.x;
                const uint32_t sum = t0 + t1 + t2 + t3;
                const uint32_t incl = wave_scan_incl(sum);
                const u64 hit = __ballot(incl >= need && incl - sum < need);
                const int L = __builtin_ctzll(hit);                      // exactly one lane: the level's total weight is >= need
                uint32_t above = (uint32_t)__builtin_amdgcn_readlane((int)(incl - sum), L);      // L is wave-uniform: v_readlane, no LDS round trip
                const uint32_t T0 = (uint32_t)__builtin_amdgcn_readlane((int)t0, L), T1 = (uint32_t)__builtin_amdgcn_readlane((int)t1, L),
                               T2 = (uint32_t)__builtin_amdgcn_readlane((int)t2, L);
                uint32_t pick = 0;
                if (above + T0 < need) { above += T0; pick = 1;
                    if (above + T1 < need) { above += T1; pick = 2;
                        if (above + T2 < need) { above += T2; pick = 3; } } }
                if (lane == 0) { sel[2 * level] = (prefix << 8) | (255u - 4u * (uint32_t)L - pick); sel[2 * level + 1] = need - above; }
            }
            __syncthreads();
            prefix = sel[2 * level];
            need = sel[2 * level + 1];
        }
        tau = prefix;
        keep = need;
    }
    THR_STAMP(15);
#undef THR_STAMP
    // per block: magnitudes >= hi_t are above tau, [lo_t, hi_t) equal it -- counted over the block's 9 candidate keys, one or two per lane
    uint32_t pk = (key0 < tau ? 1u : 0u) + (key0 <= tau ? 256u : 0u);
    if (i == 0) pk += (key1 < tau ? 1u : 0u) + (key1 <= tau ? 256u : 0u);
    pk = group8_add(pk);
    const uint32_t lo_t = pk & 0xFFu, hi_t = pk >> 8;
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

template <bool ST>
__global__ __launch_bounds__(IHTP_THREADS) void k_iht4_persist(const IhtpArgs A)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const IhtpLayout L = ihtp_layout(A.m, A.n, A.R1, A.R2, ST);
    uint32_t *A1 = reinterpret_cast<uint32_t *>(smem + L.offA1), *A2 = reinterpret_cast<uint32_t *>(smem + L.offA2);
    uint32_t *xv = reinterpret_cast<uint32_t *>(smem + L.offXV), *tv = reinterpret_cast<uint32_t *>(smem + L.offTV);
    float *c1 = reinterpret_cast<float *>(smem + L.offC1), *c2 = reinterpret_cast<float *>(smem + L.offC2);
    float *p1 = reinterpret_cast<float *>(smem + L.offP1), *p2 = reinterpret_cast<float *>(smem + L.offP2);
    uint32_t *hist = reinterpret_cast<uint32_t *>(smem + L.offHist), *wtot = reinterpret_cast<uint32_t *>(smem + L.offWtot);
    uint32_t *sel = reinterpret_cast<uint32_t *>(smem + L.offSel);
    float *pub = reinterpret_cast<float *>(smem + L.offPub);
    u64 *lut = reinterpret_cast<u64 *>(smem + L.offLut);
    u64 *raw = reinterpret_cast<u64 *>(smem + L.offRaw);

    const uint32_t tid0 = threadIdx.x, g = blockIdx.x;
    const uint32_t m = A.m, n = A.n, T1 = n / 128, T2 = m / 128;
    const uint32_t u1 = g * (A.R1 >> 4), u2 = g * (A.R2 >> 4);           // this workgroup's first unit of Phi's / PhiT's rows
    const bool has1 = u1 < m / 16, has2 = u2 < n / 16;

    // ---- once per call: the matrix slices, the per-block factors' constant halves, y, x = 0 ----
    if (has1) ihtp_load_rows(A.Phi, u1, A.R1, T1, L.TG1, A1);
    if (has2) ihtp_load_rows(A.PhiT, u2, A.R2, T2, L.TG2, A2);
    for (uint32_t i = tid0; i < L.TG1 * 64; i += IHTP_THREADS) xv[i] = 0;                      // x.clear(): nibbles 0 ...
    for (uint32_t i = tid0; i < L.TG1 * 8; i += IHTP_THREADS) { c1[i] = 0.0f; p1[i] = 0.0f; }
    for (uint32_t i = tid0; i < L.TG2 * 8; i += IHTP_THREADS) { c2[i] = 0.0f; p2[i] = 0.0f; }
    for (uint32_t i = tid0; i < L.TG2 * 64; i += IHTP_THREADS) tv[i] = 0;
    hist[tid0] = 0;
    if (tid0 < 256) lut[tid0] = ihtp_lut_entry(tid0);
    __syncthreads();
    if (has1)
        for (uint32_t b = tid0; b < n / 64; b += IHTP_THREADS) {
            const float p = A.sPhi[(size_t)(u1 >> 2) * (n / 64) + b] * CLV_RCP49;
            p1[dealt_factor(b)] = p;
            c1[dealt_factor(b)] = p * 1.0f;                                                       // ... scales 1.0
        }
    if (has2)
        for (uint32_t b = tid0; b < m / 64; b += IHTP_THREADS) p2[dealt_factor(b)] = A.sPhiT[(size_t)(u2 >> 2) * (m / 64) + b] * CLV_RCP49;
    const bool own_m = tid0 < m / 8;                                     // this thread owns word tid of the m- / n-element vectors
    const uint32_t yw = own_m ? A.y[tid0] : 0u;
    const float ys = own_m ? A.sy[tid0 >> 3] : 1.0f;
    uint32_t xw = 0;
    float xs = 1.0f;
    // stochastic: the LAST 4 * segs1 (segs2) threads are the generator lanes of the first (second) vector phase: thread -> (generator lane
    // k, segment of 16 draws).  They start at T^(position)(a0[k]) and advance by the iteration's D draws with one matrix product each
    const uint32_t G1 = m / 64, G2 = n / 64, segs1 = (G1 + 3) / 4, segs2 = (G2 + 3) / 4;
    const int gi1 = ST ? (int)tid0 - (int)(IHTP_THREADS - 4 * segs1) : -1, gi2 = ST ? (int)tid0 - (int)(IHTP_THREADS - 4 * segs2) : -1;
    u64 ga = 0, gb = 0;
    int rng_slot = 0;
    if (ST) {
        const u64 seq = rng_effective_seq(A.rng, A.seq);
        rng_slot = rng_read_slot(A.rng, seq);
        if (gi1 >= 0) ga = ihtp_rows_matvec(A.seg_rows + (size_t)(gi1 >> 2) * 64, A.rng[rng_slot * RNG_SLOT_WORDS + 4 + (gi1 & 3)]);
        if (gi2 >= 0) {
            gb = ihtp_rows_matvec(A.seg_rows + (size_t)((gi2 >> 2) + ((4 * G1) >> 4)) * 64, A.rng[rng_slot * RNG_SLOT_WORDS + 4 + (gi2 & 3)]);
            for (uint32_t z = 0; z < ((4 * G1) & 15u); z++) gb = xs_T(gb);      // m = 128 (mod 256): the phase starts half a segment in
        }
    }
    __syncthreads();

    if (A.dbg && tid0 == 0) { A.dbg[((size_t)g * 16 + 15) * 32 + 28] = __builtin_readcyclecounter(); A.dbg[((size_t)g * 16 + 15) * 32 + 29] = __builtin_amdgcn_s_memrealtime(); }
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
            const uint32_t lr = tid >> 4;
            const float dot = ihtp_row_dot(A1 + (size_t)lr * L.TG1 * 64, xv, c1, T1, tid & 15);
            if ((tid & 15) == 0) pub[lr] = dot;
        }
        if (ST && gi1 >= 0) {                                            // the first phase's draws, beside the row dots of other waves
            ga = ihtp_gen16(ga, raw, (uint32_t)gi1 >> 2, (uint32_t)gi1 & 3u);
            if (last && g == 0 && gi1 < 4) {                             // the state the launch leaves behind: part1 = T^(total - 1), part2 = T^total
                u64 *next = A.rng + (rng_slot ^ 1) * RNG_SLOT_WORDS;
                const u64 f = ihtp_cols_matvec(A.jump1, ga);
                next[gi1] = f;
                next[4 + gi1] = xs_T(f);
            }
            ga = ihtp_cols_matvec(A.jump, ga);
        }
        IHTP_STAMP(1);
        __syncthreads();
        if (has1 && tid < A.R1)
            __hip_atomic_store((gu64 *)A.g1 + unit_slot(u1 + (tid >> 4), tid & 15, m), ((u64)epoch << 32) | __float_as_uint(pub[tid]), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        // ---- E1: all of d1 -> t1 = quantize(d1), t2 = quantize(y - t1) ----
        if ((tid & ~63u) < m / 8) {                                      // wave-uniform: waves that own no word skip the gather
            float d[8];
            ihtp_gather8(A.g1, m, tid, epoch, tid < m / 8, A.nap0, A.nap, d);
            IHTP_STAMP(2);
            if (tid < m / 8) {
                uint32_t t1w, t2w, W[4] = {0u, 0u, 0u, 0u};
                float t1s, t2s;
                if (ST) {
                    const uint32_t *r32 = reinterpret_cast<const uint32_t *>(raw) + (tid & 7u), b = tid >> 3;
                    W[0] = r32[(2 * b) * 8]; W[1] = r32[(2 * b + 1) * 8]; W[2] = r32[(2 * G1 + 2 * b) * 8]; W[3] = r32[(2 * G1 + 2 * b + 1) * 8];
                }
                ihtp_requant_saa<ST>(d, yw, ys, -1.0f, W, t1w, t1s, t2w, t2s);
                tv[dealt_word(tid)] = t2w;
                if ((tid & 7) == 0) c2[dealt_factor(tid >> 3)] = p2[dealt_factor(tid >> 3)] * t2s;
                if (last && g == 0) {
                    A.t1[tid] = t1w;
                    A.t2[tid] = t2w;
                    if ((tid & 7) == 0) { A.st1[tid >> 3] = t1s; A.st2[tid >> 3] = t2s; }
                }
            }
        }
        IHTP_STAMP(3);
        __syncthreads();
        IHTP_STAMP(4);
        // ---- P2: t3's row dots ----
        if (has2 && tid < 16 * A.R2) {
            const uint32_t lr = tid >> 4;
            const float dot = ihtp_row_dot(A2 + (size_t)lr * L.TG2 * 64, tv, c2, T2, tid & 15);
            if ((tid & 15) == 0) pub[lr] = dot;
        }
        if (ST && gi2 >= 0) {                                            // the second phase's draws (the first phase's have been used)
            gb = ihtp_gen16(gb, raw, (uint32_t)gi2 >> 2, (uint32_t)gi2 & 3u);
            gb = ihtp_cols_matvec(A.jump, gb);
        }
        IHTP_STAMP(5);
        __syncthreads();
        if (has2 && tid < A.R2)
            __hip_atomic_store((gu64 *)A.g2 + unit_slot(u2 + (tid >> 4), tid & 15, n), ((u64)epoch << 32) | __float_as_uint(pub[tid]), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        // ---- E2: all of d3 -> t3 = quantize(d3), x = quantize(x + mu t3), threshold ----
        uint32_t t3w = 0;
        float t3s = 1.0f;
        if ((tid & ~63u) < n / 8) {
            float d[8];
            ihtp_gather8(A.g2, n, tid, epoch, tid < n / 8, A.nap0, A.nap, d);
            IHTP_STAMP(6);
            if (tid < n / 8) {
                uint32_t W[4] = {0u, 0u, 0u, 0u};
                if (ST) {
                    const uint32_t *r32 = reinterpret_cast<const uint32_t *>(raw) + (tid & 7u), b = tid >> 3;
                    W[0] = r32[(2 * b) * 8]; W[1] = r32[(2 * b + 1) * 8]; W[2] = r32[(2 * G2 + 2 * b) * 8]; W[3] = r32[(2 * G2 + 2 * b + 1) * 8];
                }
                ihtp_requant_saa<ST>(d, xw, xs, A.mu, W, t3w, t3s, xw, xs);
            }
        }
        IHTP_STAMP(7);
        if (A.threshold && A.K < A.x_len) xw = ihtp_threshold(xw, xs, tid, A.x_len, A.K, hist, wtot, sel, lut,
                                                                       A.dbg && it < 16 ? A.dbg + ((size_t)g * 16 + it) * 32 : nullptr);       // all threads: barriers inside
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
    if (A.dbg && tid0 == 0) { A.dbg[((size_t)g * 16 + 15) * 32 + 30] = __builtin_readcyclecounter(); A.dbg[((size_t)g * 16 + 15) * 32 + 31] = __builtin_amdgcn_s_memrealtime(); }
    if (ST && g == 0) {                                                  // stamp the slot written in the last iteration (rng_device.h: rng_commit)
        __syncthreads();
        if (tid0 == IHTP_THREADS - 4 * segs1) {
            __threadfence();
            A.rng[(rng_slot ^ 1) * RNG_SLOT_WORDS + RNG_STAMP_WORD] = rng_effective_seq(A.rng, A.seq);
        }
    }
}

// =====================================================================================================================================
// The same loop for CloverVector8 vectors: clm4_iht_v8, the configuration the reference measures and publishes as its "4-bit" IHT / GD
// (CloverMatrix4 x CloverVector8: test/performance/02_bit04.cpp:140, doc/results/performance.txt:597-606).
//
// What differs from the 4-bit kernel above is the row dot (CloverMatrix4.h:1093-1441): per row EIGHT fp32 fma chains, chain L taking from
// EVERY 64-column block b the exact integer I = sum of q4 * q8 over elements 4L..4L+3 and 32+4L..32+4L+3, acc = fma(c_b, (float)I, acc)
// with c_b = f32(f32(sA/7) * f32(sx/127)) -- n / 64 dependent steps per chain -- and ((a0+a4)+(a2+a6)) + ((a1+a5)+(a3+a7)).
// Lane = (row, chain).  For v_dot8_i32_i4 both operands are laid out per (block, chain):
//   * the matrix slice is re-dealt ONCE, on its way into LDS: word (b, L) = the chain's 8 nibbles (two half words of the row's block);
//   * an int8 x = 16 (xh + xc) + xl with xl = its signed low nibble, xc = 1 only for x >= 120 (where the high part would be 8), xh in
//     [-8, 7]: three nibble images of the vector, written by the thread that re-quantises the element, and
//     I = 16 (dot8(A, xh) + dot8(A, xc)) + dot8(A, xl): three instructions, exact, instead of the two v_dot4_i32_i8 and four widening
//     operations per half word of the launch-per-step kernel (k_m4_mvm8).
// Everything else is the 4-bit kernel's: units, granules, gathers, one re-quantisation per consumer.  Deterministic rounding, threshold
// FAST (k_thresh8_small's algorithm: element keys, four 8-bit levels, 8 copies of the bins) or none; otherwise clm4_iht_v8 keeps its loop.
// words between two copies of the threshold's 256 bins: the eight lanes of a block add to eight copies, and the same bin of two copies must
// not share an LDS bank (256 would put all eight on one)
#ifndef IHTP8_HSTRIDE
#define IHTP8_HSTRIDE 260
#endif
struct Ihtp8Layout {
    uint32_t GB1, GB2;            // groups of 4 blocks of Phi's / PhiT's rows
    uint32_t offA1, offA2, offX, offT, offC1, offC2, offP1, offP2, offHist, offHsum, offWtot, offChain, total;
    uint32_t raw1_bytes, raw2_bytes, raw1_room, raw2_room;      // stochastic: the phases' raw draws and the regions they overlay
};
__host__ __device__ inline Ihtp8Layout ihtp8_layout(uint32_t m, uint32_t n, uint32_t R1, uint32_t R2)
{
    Ihtp8Layout L;
    L.GB1 = ((n / 64 + 3) / 4 + 1) & ~1u;                                // an even number of groups: swz8 moves a slot to its neighbour group's
    L.GB2 = ((m / 64 + 3) / 4 + 1) & ~1u;
    uint32_t o = 0;
    L.offA1 = o; o += R1 * L.GB1 * 128;                                 // [row][group][chain][4 blocks] words
    L.offA2 = o; o += R2 * L.GB2 * 128;
    L.offX = o; o += 3 * L.GB1 * 128;                                   // x: the high / carry / low nibble images, [group][chain][4 blocks]
    // t2's images, the radix bins and their sums lie side by side: with stochastic rounding the second vector phase's raw draws overlay all
    // three (t2 is dead behind the second row dots, the bins are cleared again in front of the threshold), the first phase's overlay x's
    // images (dead between the first row dots and the end of the iteration): at N = 8192 there is no LDS left for buffers of their own
    L.offT = o; o += 3 * L.GB2 * 128;                                   // t2 likewise
    L.offHist = o; o += 8 * IHTP8_HSTRIDE * 4;                          // one radix level, 8 copies
    L.offHsum = o; o += 2 * 256 * 4;
    L.offC1 = o; o += L.GB1 * 16 + 16;                                  // c_b, [group][4 blocks], and one group of zeros behind them (what a
    L.offC2 = o; o += L.GB2 * 16 + 16;                                  // helper lane's unused slots read)
    L.offP1 = o; o += L.GB1 * 16;                                       // f32(sA_b / 7)
    L.offP2 = o; o += L.GB2 * 16;
    L.offWtot = o; o += 64;
    L.offChain = o; o += (R1 > R2 ? R1 : R2) * 8 * 4;                    // the 8 chain sums of every local row, on their way to the row's tree
    L.total = o + 256;                                                  // the dot loop reads one group past an array's end
    L.raw1_bytes = ((m / 64 + 3) / 4) * 16 * 32;
    L.raw2_bytes = ((n / 64 + 3) / 4) * 16 * 32;
    L.raw1_room = 3 * L.GB1 * 128;
    L.raw2_room = 3 * L.GB2 * 128 + 8 * IHTP8_HSTRIDE * 4 + 2 * 256 * 4;
    return L;
}
// The 16-byte slot of (group g, chain L) inside a row / an image: g * 8 + L, with bits 2..3 flipped by the low two bits of the group's
// helper (g >> gl; ihtp8_row_dots_par: helper h reads groups (h << gl) + k with lanes ordered chain-major, helper-minor).  A ds_read_b128
// serves 16 lanes per LDS cycle from 64 banks (16 slots of 16 bytes per 256 bytes); unswizzled, the 4 helpers of a chain that fall into one
// such lane group read slots 32 apart -- the same banks, four cycles instead of one for four of a step's five reads.  With the flip the 16
// lanes of a group (4 chains x 4 helpers) hit 16 different slots mod 16.
__device__ __forceinline__ uint32_t swz8(uint32_t slot, uint32_t gl) { return slot ^ ((((slot >> 3) >> gl) & 3u) << 2); }
__device__ __forceinline__ uint32_t dealt8(uint32_t b, uint32_t L, uint32_t gl) { return swz8((b >> 2) * 8 + L, gl) * 4 + (b & 3); }

// this workgroup's R rows into LDS, re-dealt per (block, chain): work item = (local row, group of 4 blocks, word pair k): words k and
// 4 + k of four blocks -> chains 2k (their low halves) and 2k + 1 (their high halves), two ds_write_b128
__device__ __forceinline__ void ihtp8_load_rows(const uint8_t *__restrict__ A, uint32_t u0, uint32_t R, uint32_t NB, uint32_t GB, uint32_t gl, uint32_t *lds)
{
    const uint32_t items = R * GB * 4;
    for (uint32_t it = threadIdx.x; it < items; it += IHTP_THREADS) {
        const uint32_t k = it & 3, g = (it >> 2) % GB, lr = (it >> 2) / GB;
        const uint64_t row = unit_row(u0 + (lr >> 4), lr & 15);
        const uint32_t *src = reinterpret_cast<const uint32_t *>(A + row * (uint64_t)NB * 32);
        uint32_t ev[4], od[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint32_t b = 4 * g + i;
            const uint32_t lo = b < NB ? src[b * 8 + k] : 0u, hi = b < NB ? src[b * 8 + 4 + k] : 0u;
            ev[i] = __builtin_amdgcn_perm(hi, lo, 0x05040100u);           // elements 8k..8k+3 | 32+8k..32+8k+3: chain 2k
            od[i] = __builtin_amdgcn_perm(hi, lo, 0x07060302u);           // elements 8k+4..8k+7 | 36+8k..39+8k: chain 2k + 1
        }
        u32x4 *dst = reinterpret_cast<u32x4 *>(lds + (size_t)lr * GB * 32);
        dst[swz8(g * 8 + 2 * k, gl)] = u32x4{ev[0], ev[1], ev[2], ev[3]};
        dst[swz8(g * 8 + 2 * k + 1, gl)] = u32x4{od[0], od[1], od[2], od[3]};
    }
}

// The row dots with helper lanes (late r6; a lane per chain kept 128 or 256 lanes busy for n / 64 dependent steps: 4.5 + 2.6 us of the
// iteration at N = 8192).  A chain is dealt over H = 128 / R consecutive lanes ("helpers"): helper h forms the exact
// integers (and their floats) of its four groups of four blocks -- the part that does not depend on the chain -- and then the fp32 chain
// itself walks through the helpers in block order: H stages of 16 dependent fmas, the running sum handed to the next lane by one DPP
// shift after each stage.  Every lane computes in every stage (what it computes outside its own stage is never used), so a wave issues
// 16 H fmas -- as many as one lane of the old form -- but 64 lanes' worth of chains at once, and the integer work is spread over all of
// them.  Same integers, same factors, same order of the fmas: the same bits.  A helper's slots beyond the row's blocks read a group of
// zero factors: fma(0, i, acc) = acc for any finite i (acc is never -0: it starts at +0 and x + (-x) rounds to +0).  A helper takes 2^gl <= 4 groups:
// GB <= 4 H (clm4_iht_v8_persistent checks; LDS bounds R GB to ~300, i.e. GB / H to 2.4).
// A lane carries TWO rows (rows 2 rp, 2 rp + 1 of the workgroup; 512 lanes work): the rows share the vector's nibble images and the
// factors, and their two chains advance in one v_pk_fma_f32 per step -- the H stages, which every lane of the wave walks for the one
// helper whose turn it is, are the larger half of the phase's instructions, and this halves them per row (14.1 -> 13.2 us per iteration
// at N = 8192 against one row per lane on all 1024).
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int H>
__device__ __forceinline__ void ihtp8_row_dots_par2(const uint32_t *Abase, uint32_t GB, uint32_t gl, const uint32_t *X, uint32_t xs, const float *cf,
                                                    uint32_t tid, float *chain /* LDS [R][8] */)
{
    const uint32_t h = tid & (H - 1), L = (tid / H) & 7u, rp = tid / (8 * H);
    const u32x4 *Ap0 = reinterpret_cast<const u32x4 *>(Abase + (size_t)(2 * rp) * GB * 32), *Ap1 = Ap0 + GB * 8;
    const u32x4 *Hp = reinterpret_cast<const u32x4 *>(X), *Cp = Hp + xs / 4, *Lp = Cp + xs / 4;
    const f32x4 *Fp = reinterpret_cast<const f32x4 *>(cf);
    f32x2 fi[16];
    float ff[16];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint32_t g = (h << gl) + k;
        const bool on = (uint32_t)k < (1u << gl) && g < GB;
        const uint32_t slot = swz8((on ? g : 0u) * 8 + L, gl);
        const u32x4 a = Ap0[slot], b = Ap1[slot], xh = Hp[slot], xc = Cp[slot], xl = Lp[slot];
        const f32x4 f = Fp[on ? g : GB];
        fi[4 * k] = f32x2{(float)sdot8(a.x, xl.x, sdot8(a.x, xc.x, sdot8z(a.x, xh.x)) << 4), (float)sdot8(b.x, xl.x, sdot8(b.x, xc.x, sdot8z(b.x, xh.x)) << 4)};
        fi[4 * k + 1] = f32x2{(float)sdot8(a.y, xl.y, sdot8(a.y, xc.y, sdot8z(a.y, xh.y)) << 4), (float)sdot8(b.y, xl.y, sdot8(b.y, xc.y, sdot8z(b.y, xh.y)) << 4)};
        fi[4 * k + 2] = f32x2{(float)sdot8(a.z, xl.z, sdot8(a.z, xc.z, sdot8z(a.z, xh.z)) << 4), (float)sdot8(b.z, xl.z, sdot8(b.z, xc.z, sdot8z(b.z, xh.z)) << 4)};
        fi[4 * k + 3] = f32x2{(float)sdot8(a.w, xl.w, sdot8(a.w, xc.w, sdot8z(a.w, xh.w)) << 4), (float)sdot8(b.w, xl.w, sdot8(b.w, xc.w, sdot8z(b.w, xh.w)) << 4)};
        ff[4 * k] = f.x; ff[4 * k + 1] = f.y; ff[4 * k + 2] = f.z; ff[4 * k + 3] = f.w;
    }
    f32x2 in = {0.0f, 0.0f}, acc = {0.0f, 0.0f};
#pragma unroll
    for (int j = 0; j < H; j++) {
        acc = in;
#pragma unroll
        for (int i = 0; i < 16; i++) acc = __builtin_elementwise_fma(f32x2{ff[i], ff[i]}, fi[i], acc);
        in.x = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(acc.x), 0x111 /* row_shr:1 */, 0xF, 0xF, false));
        in.y = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(acc.y), 0x111, 0xF, 0xF, false));
    }
    if (h == H - 1) { chain[(2 * rp) * 8 + L] = acc.x; chain[(2 * rp + 1) * 8 + L] = acc.y; }
}
// the reference's tree over a row's 8 chain sums (CloverMatrix4.h:1229-1234): ((a0 + a4) + (a2 + a6)) + ((a1 + a5) + (a3 + a7))
__device__ __forceinline__ float ihtp8_row_tree(const float *chain)
{
    const f32x4 lo = *reinterpret_cast<const f32x4 *>(chain), hi = *reinterpret_cast<const f32x4 *>(chain + 4);
    const float h0 = lo.x + hi.x, h1 = lo.y + hi.y, h2 = lo.z + hi.z, h3 = lo.w + hi.w;
    return (h0 + h2) + (h1 + h3);
}

// the three nibble images of 4 int8 values (one chain's half word, packed one per byte): x = 16 (xh + xc) + xl with xl the signed low nibble,
// xh + xc = (x + 8) >> 4 in [-8, 8] and xc = 1 where that is 8 (x >= 120).  All four bytes at once: biased to unsigned (u = x + 128), u + 8
// is added per byte without carries between bytes; the byte's own carry out IS xc, and the high nibble of the sum, re-biased (^ 8), minus xc
// is xh.  Each image's four nibbles go into 16 bits, even elements in the high nibble of their byte as the matrix stores them.
// (Element by element this was ~150 instructions per thread and store: 1.2 us of an iteration for x's images alone.)
__device__ __forceinline__ uint32_t ihtp8_pack16(uint32_t x) { const uint32_t y = (x >> 8) | (x << 4); return (y & 0xFFu) | ((y >> 8) & 0xFF00u); }
__device__ __forceinline__ void ihtp8_split4(uint32_t P, uint32_t &h, uint32_t &c, uint32_t &l)
{
    const uint32_t U = P ^ 0x80808080u, A = (U & 0x7F7F7F7Fu) + 0x08080808u, S = A ^ (U & 0x80808080u);
    const uint32_t cyb = (U & A & 0x80808080u) >> 7;
    h = ihtp8_pack16(((((S >> 4) & 0x0F0F0F0Fu) ^ 0x08080808u) - cyb) & 0x0F0F0F0Fu);
    c = ihtp8_pack16(cyb);
    l = ihtp8_pack16(P & 0x0F0F0F0Fu);
}
// p0 / p1: the thread's elements 0..3 / 4..7, one int8 per byte
__device__ __forceinline__ void ihtp8_store_images(uint32_t *X, uint32_t xs, uint32_t gl, uint32_t b, uint32_t i, uint32_t p0, uint32_t p1)
{
    uint16_t *X16 = reinterpret_cast<uint16_t *>(X);
#pragma unroll
    for (int half = 0; half < 2; half++) {
        uint32_t h, c, l;
        ihtp8_split4(half ? p1 : p0, h, c, l);
        const uint32_t w = dealt8(b, 2 * (i & 3) + half, gl) * 2 + (i >> 2);   // 16-bit index
        X16[w] = (uint16_t)h;
        X16[w + 2 * xs] = (uint16_t)c;
        X16[w + 4 * xs] = (uint16_t)l;
    }
}

// r = quantize8(d), o = quantize8(u + a r) for the thread's 8 elements (k_m4_mvm8's fused epilogue, mixed8.hip: CloverMatrix4.h:1246-1440,
// CloverVector8.h:1089-1358); u = the thread's two words of u.  ST: thread i of the block's 8 (elements l = 8 i + e) takes, for the mvm's
// re-quantisation, byte i & 3 of word e of the block's draw i >> 2 (noise group g = l >> 3, word l & 7: CloverMatrix4.h:1246-1440) and,
// for scaleAndAdd, byte e & 3 of word 2 (i & 3) + (e >> 2) of its draw i >> 2 (element l: draw l >> 5, word (l & 31) >> 2, byte l & 3:
// CloverVector8.h:1104-1126).  Wm = the 8 words of the mvm draw, Ws = the two words of the scaleAndAdd draw.
template <bool ST>
__device__ __forceinline__ void ihtp8_requant_saa(const float d[8], const uint32_t uw[2], float us, float a, uint32_t i, const uint32_t Wm[8],
                                                  const uint32_t Ws[2], int q[8], float &rs, int q2[8], float &os)
{
    float mx = 0.0f;
#pragma unroll
    for (int e = 0; e < 8; e++) mx = fmaxf(mx, __builtin_fabsf(d[e]));
    rs = fix_zero_max(group8_max(mx));
    const float k = 127.0f / rs;
#pragma unroll
    for (int e = 0; e < 8; e++) q[e] = quant1(d[e], k, ST ? noise_of(Wm[e], (int)(i & 3u)) : 0.0f);
    const float su127 = div127(us), sv127 = div127(rs * a);
    float val[8], m2 = 0.0f;
#pragma unroll
    for (int e = 0; e < 8; e++) {
        const float qu = (float)((int)(uw[e >> 2] << (24 - 8 * (e & 3))) >> 24);
        val[e] = __builtin_fmaf((float)q[e], sv127, qu * su127);
        m2 = fmaxf(m2, __builtin_fabsf(val[e]));
    }
    os = fix_zero_max(group8_max(m2));
    const float k2 = 127.0f / os;
#pragma unroll
    for (int e = 0; e < 8; e++) q2[e] = quant1(val[e], k2, ST ? noise_of(Ws[e >> 2], e & 3) : 0.0f);
}
__device__ __forceinline__ uint32_t pack4_i8(const int q[4]) { return ((uint32_t)q[0] & 0xFFu) | (((uint32_t)q[1] & 0xFFu) << 8) | (((uint32_t)q[2] & 0xFFu) << 16) | ((uint32_t)q[3] << 24); }

// CloverVector8::threshold(K), FAST rule: k_thresh8_small's algorithm (threshold4.hip) for a vector held 8 elements per thread.  q: the
// thread's elements; returns them with the losers zeroed.  hist[8][256] zero on entry and again on exit; hsum[2][256], wtot[16].
// (Measured and dropped: counting the keys that share the bin of the thread's first key in a register and adding them up over the block's 8
// lanes before ONE atomic -- the upper levels' bins are few -- changed nothing: 18.0 against 17.5 us per iteration at N = 8192.  Late r6,
// with phase stamps (5.2 us of the iteration): one pre-cleared pair of bin copies per level and ONE barrier per level instead of two --
// 6.2 us, two copies take the conflicts eight spread; a wave-wide single atomic for level 0's common top byte -- 5.7 us.  one scanning
// wave that also adds the copies up and hands {bin, rest} over through LDS -- 16.2 against 15.9 us per iteration.  None of the three parts
// of a level (atomics, barriers, scans) stands out; the kernel keeps the form below.  By the stamps: level 0 2.1 us -- mostly the wait for the
// workgroup's slowest wave to come out of the gather --, levels 1-3 0.8 us each, ranks and apply 0.8.  Listing the keys that are left after
// two levels (a dozen) and ranking them in one wave instead of levels 2 and 3: 1.4 against 1.6 us, not kept; levels 1-3 on ONE copy of the bins
// with one barrier each (no adding up): 0.63 against 0.79 us per level by the stamps, 13.4 against 13.4 us per iteration -- a level is
// its ~100 instructions in each of 16 waves, whatever the barriers.)
__device__ __forceinline__ void ihtp8_threshold(int q[8], float s, uint32_t tid_, uint32_t n, uint32_t k, uint32_t *hist, uint32_t *hsum, uint32_t *wtot,
                                                u64 *dbg = nullptr)
{
#define THR8_STAMP(k_) do { if (dbg && threadIdx.x == 0) dbg[k_] = __builtin_amdgcn_s_memrealtime(); } while (0)
    const int tid = (int)tid_, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    uint32_t keys[8], valid = 0;
#pragma unroll
    for (int e = 0; e < 8; e++) {
        keys[e] = __float_as_uint(__builtin_fabsf(div127((float)q[e] * s)));
        if (8u * tid + e < n) valid |= 1u << e;
    }
    uint32_t tau = 0x7F800000u, keep = 0;
    THR8_STAMP(10);
    if (k != 0) {
        uint32_t prefix = 0, need = k;
        uint32_t *hp = hist + IHTP8_HSTRIDE * (tid & 7);
#pragma unroll
        for (int level = 0; level < 4; level++) {
            const int shift = 24 - 8 * level;
#pragma unroll
            for (int e = 0; e < 8; e++)
                if (((valid >> e) & 1u) && (level == 0 || (keys[e] >> (shift + 8)) == prefix)) atomicAdd(&hp[(keys[e] >> shift) & 0xFFu], 1u);
            __syncthreads();
            if (tid < 256) {                                             // the copies are added up once, and cleared for the next level / call
                uint32_t mine = 0;
#pragma unroll
                for (int cpy = 0; cpy < 8; cpy++) { mine += hist[IHTP8_HSTRIDE * cpy + tid]; hist[IHTP8_HSTRIDE * cpy + tid] = 0; }
                hsum[256 * (level & 1) + tid] = mine;
            }
            __syncthreads();
            const u32x4 h4 = *reinterpret_cast<const u32x4 *>(hsum + 256 * (level & 1) + 252 - 4 * lane);
            const uint32_t t0 = h4.w, t1 = h4.z, t2 = h4.y, t3 = h4.x;
            const uint32_t sum = t0 + t1 + t2 + t3;
            const uint32_t incl = wave_scan_incl(sum);
            const u64 hit = __ballot(incl >= need && incl - sum < need);
            const int Ls = __builtin_ctzll(hit);
            uint32_t above = (uint32_t)__builtin_amdgcn_readlane((int)(incl - sum), Ls);
            const uint32_t T0 = (uint32_t)__builtin_amdgcn_readlane((int)t0, Ls), T1 = (uint32_t)__builtin_amdgcn_readlane((int)t1, Ls),
                           T2 = (uint32_t)__builtin_amdgcn_readlane((int)t2, Ls);
            uint32_t pick = 0;
            if (above + T0 < need) { above += T0; pick = 1;
                if (above + T1 < need) { above += T1; pick = 2;
                    if (above + T2 < need) { above += T2; pick = 3; } } }
            prefix = (prefix << 8) | (255u - 4u * (uint32_t)Ls - pick);
            need -= above;
            THR8_STAMP(11 + level);
        }
        tau = prefix;
        keep = need;
    }
    uint32_t c = 0;
#pragma unroll
    for (int e = 0; e < 8; e++) c += ((valid >> e) & 1u) && keys[e] == tau;
    THR8_STAMP(15);
    const uint32_t vinc = wave_scan_incl(c);
    if (lane == 63) wtot[wave] = vinc;
    __syncthreads();
    const uint32_t tot = lane < 16 ? wtot[lane] : 0;
    const uint32_t inc = wave_scan_incl(tot);
    uint32_t rank = vinc - c + (uint32_t)__builtin_amdgcn_readlane((int)(inc - tot), wave);
#pragma unroll
    for (int e = 0; e < 8; e++) {
        if (!((valid >> e) & 1u)) continue;                              // padding is left alone
        if (keys[e] > tau) continue;
        if (keys[e] == tau) { if (rank >= keep) q[e] = 0; rank++; }
        else q[e] = 0;
    }
#undef THR8_STAMP
}

struct Ihtp8Args {
    const uint8_t *Phi;
    const float *sPhi;
    const uint8_t *PhiT;
    const float *sPhiT;
    uint32_t m, n, x_len, R1, R2;
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
    int threshold;
    u64 *g1, *g2;
    uint32_t nap0, nap;
    u64 *dbg;                     // probe only: phase stamps, as IhtpArgs
    u64 *rng;                     // stochastic rounding (k_iht8_persist<true>): as IhtpArgs
    u64 seq;
    const u64 *seg_rows;
    u64 jump[64], jump1[64];
};

template <bool ST>
__global__ __launch_bounds__(IHTP_THREADS) void k_iht8_persist(const Ihtp8Args A)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const Ihtp8Layout L = ihtp8_layout(A.m, A.n, A.R1, A.R2);
    uint32_t *A1 = reinterpret_cast<uint32_t *>(smem + L.offA1), *A2 = reinterpret_cast<uint32_t *>(smem + L.offA2);
    uint32_t *X = reinterpret_cast<uint32_t *>(smem + L.offX), *T = reinterpret_cast<uint32_t *>(smem + L.offT);
    float *c1 = reinterpret_cast<float *>(smem + L.offC1), *c2 = reinterpret_cast<float *>(smem + L.offC2);
    float *p1 = reinterpret_cast<float *>(smem + L.offP1), *p2 = reinterpret_cast<float *>(smem + L.offP2);
    uint32_t *hist = reinterpret_cast<uint32_t *>(smem + L.offHist), *hsum = reinterpret_cast<uint32_t *>(smem + L.offHsum);
    uint32_t *wtot = reinterpret_cast<uint32_t *>(smem + L.offWtot);
    float *chain = reinterpret_cast<float *>(smem + L.offChain);

    const uint32_t tid0 = threadIdx.x, g = blockIdx.x;
    const uint32_t m = A.m, n = A.n, NB1 = n / 64, NB2 = m / 64, XS1 = L.GB1 * 32, XS2 = L.GB2 * 32;      // words per nibble image
    const uint32_t u1 = g * (A.R1 >> 4), u2 = g * (A.R2 >> 4);
    const bool has1 = u1 < m / 16, has2 = u2 < n / 16;

    // groups per helper lane of the row dots, as a power of two (H = 128 / R helpers per chain): 2^gl >= GB / H
    const uint32_t H1 = 128u / A.R1, H2 = 128u / A.R2;
    const uint32_t gl1 = L.GB1 <= H1 ? 0u : (L.GB1 <= 2 * H1 ? 1u : 2u), gl2 = L.GB2 <= H2 ? 0u : (L.GB2 <= 2 * H2 ? 1u : 2u);
    if (has1) ihtp8_load_rows(A.Phi, u1, A.R1, NB1, L.GB1, gl1, A1);
    if (has2) ihtp8_load_rows(A.PhiT, u2, A.R2, NB2, L.GB2, gl2, A2);
    for (uint32_t i = tid0; i < 3 * XS1; i += IHTP_THREADS) X[i] = 0;                          // x.clear(): every image of 0 is 0 ...
    for (uint32_t i = tid0; i < 3 * XS2; i += IHTP_THREADS) T[i] = 0;
    for (uint32_t i = tid0; i < L.GB1 * 4; i += IHTP_THREADS) { c1[i] = 0.0f; p1[i] = 0.0f; }
    for (uint32_t i = tid0; i < L.GB2 * 4; i += IHTP_THREADS) { c2[i] = 0.0f; p2[i] = 0.0f; }
    if (tid0 < 4) { c1[L.GB1 * 4 + tid0] = 0.0f; c2[L.GB2 * 4 + tid0] = 0.0f; }
    hist[tid0] = 0;
    hist[tid0 + 1024] = 0;
    if (tid0 < 8 * IHTP8_HSTRIDE - 2048) hist[2048 + tid0] = 0;
    __syncthreads();
    if (has1)
        for (uint32_t b = tid0; b < NB1; b += IHTP_THREADS) {
            const float p = A.sPhi[(size_t)(u1 >> 2) * NB1 + b] * (1.0f / 7.0f);              // CloverMatrix4.h:1147-1149
            p1[b] = p;
            c1[b] = p * (1.0f * (1.0f / 127.0f));                                                // ... scales 1.0
        }
    if (has2)
        for (uint32_t b = tid0; b < NB2; b += IHTP_THREADS) p2[b] = A.sPhiT[(size_t)(u2 >> 2) * NB2 + b] * (1.0f / 7.0f);
    const bool own_m = tid0 < m / 8;
    uint32_t yw[2] = {own_m ? A.y[2 * tid0] : 0u, own_m ? A.y[2 * tid0 + 1] : 0u};
    const float ys = own_m ? A.sy[tid0 >> 3] : 1.0f;
    uint32_t xw[2] = {0u, 0u};
    float xs = 1.0f;
    // stochastic: generator lanes as in k_iht4_persist (the last 4 * segs threads of a phase, 16 draws each, one GF(2) jump per iteration); the
    // draws of the first phase go where x's images lie (dead behind the first row dots), the second phase's where t2's images and the radix
    // bins lie (dead behind the second row dots / cleared again in front of the threshold)
    const uint32_t G1 = m / 64, G2 = n / 64, segs1 = (G1 + 3) / 4, segs2 = (G2 + 3) / 4;
    const int gi1 = ST ? (int)tid0 - (int)(IHTP_THREADS - 4 * segs1) : -1, gi2 = ST ? (int)tid0 - (int)(IHTP_THREADS - 4 * segs2) : -1;
    u64 ga = 0, gb = 0;
    int rng_slot = 0;
    u64 *raw1 = reinterpret_cast<u64 *>(X), *raw2 = reinterpret_cast<u64 *>(T);
    if (ST) {
        const u64 seq = rng_effective_seq(A.rng, A.seq);
        rng_slot = rng_read_slot(A.rng, seq);
        if (gi1 >= 0) ga = ihtp_rows_matvec(A.seg_rows + (size_t)(gi1 >> 2) * 64, A.rng[rng_slot * RNG_SLOT_WORDS + 4 + (gi1 & 3)]);
        if (gi2 >= 0) {
            gb = ihtp_rows_matvec(A.seg_rows + (size_t)((gi2 >> 2) + ((4 * G1) >> 4)) * 64, A.rng[rng_slot * RNG_SLOT_WORDS + 4 + (gi2 & 3)]);
            for (uint32_t z = 0; z < ((4 * G1) & 15u); z++) gb = xs_T(gb);
        }
    }
    __syncthreads();

    for (uint32_t it = 0; it < A.iterations; it++) {
        const uint32_t epoch = it + 1;
        uint32_t tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        const bool last = it + 1 == A.iterations;
        IHTP_STAMP(0);
        // ---- P1 ----
        if (has1 && tid < IHTP_THREADS / 2) {                            // two rows per lane (the host has checked GB <= 4 H)
            if (A.R1 == 16) ihtp8_row_dots_par2<8>(A1, L.GB1, gl1, X, XS1, c1, tid, chain);
            else if (A.R1 == 32) ihtp8_row_dots_par2<4>(A1, L.GB1, gl1, X, XS1, c1, tid, chain);
            else ihtp8_row_dots_par2<2>(A1, L.GB1, gl1, X, XS1, c1, tid, chain);
        }
        IHTP_STAMP(1);
        __syncthreads();
        if (has1 && tid < A.R1)
            __hip_atomic_store((gu64 *)A.g1 + unit_slot(u1 + (tid >> 4), tid & 15, m),
                               ((u64)epoch << 32) | __float_as_uint(ihtp8_row_tree(chain + 8 * tid)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (ST && gi1 >= 0) {                                            // behind the row dots: x's images are dead until the end of the iteration
            ga = ihtp_gen16(ga, raw1, (uint32_t)gi1 >> 2, (uint32_t)gi1 & 3u);
            if (last && g == 0 && gi1 < 4) {
                u64 *next = A.rng + (rng_slot ^ 1) * RNG_SLOT_WORDS;
                const u64 f = ihtp_cols_matvec(A.jump1, ga);
                next[gi1] = f;
                next[4 + gi1] = xs_T(f);
            }
            ga = ihtp_cols_matvec(A.jump, ga);
        }
        // ---- E1 ----
        float d[8];
        if ((tid & ~63u) < m / 8) ihtp_gather8(A.g1, m, tid, epoch, tid < m / 8, A.nap0, A.nap, d);
        IHTP_STAMP(2);
        if (ST) __syncthreads();                                         // the first phase's draws are in LDS
        {
            if (tid < m / 8) {
                int q1[8], q2[8];
                float t1s, t2s;
                uint32_t Wm[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u}, Ws[2] = {0u, 0u};
                if (ST) {
                    const uint32_t *r32 = reinterpret_cast<const uint32_t *>(raw1), b = tid >> 3, i = tid & 7u;
                    const u32x4 w0 = *reinterpret_cast<const u32x4 *>(r32 + (2 * b + (i >> 2)) * 8), w1 = *reinterpret_cast<const u32x4 *>(r32 + (2 * b + (i >> 2)) * 8 + 4);
                    Wm[0] = w0.x; Wm[1] = w0.y; Wm[2] = w0.z; Wm[3] = w0.w; Wm[4] = w1.x; Wm[5] = w1.y; Wm[6] = w1.z; Wm[7] = w1.w;
                    Ws[0] = r32[(2 * G1 + 2 * b + (i >> 2)) * 8 + 2 * (i & 3u)];
                    Ws[1] = r32[(2 * G1 + 2 * b + (i >> 2)) * 8 + 2 * (i & 3u) + 1];
                }
                ihtp8_requant_saa<ST>(d, yw, ys, -1.0f, tid & 7u, Wm, Ws, q1, t1s, q2, t2s);
                const uint32_t t2w0 = pack4_i8(q2), t2w1 = pack4_i8(q2 + 4);
                ihtp8_store_images(T, XS2, gl2, tid >> 3, tid & 7, t2w0, t2w1);
                if ((tid & 7) == 0) c2[tid >> 3] = p2[tid >> 3] * (t2s * (1.0f / 127.0f));
                if (last && g == 0) {
                    A.t1[2 * tid] = pack4_i8(q1); A.t1[2 * tid + 1] = pack4_i8(q1 + 4);
                    A.t2[2 * tid] = t2w0; A.t2[2 * tid + 1] = t2w1;
                    if ((tid & 7) == 0) { A.st1[tid >> 3] = t1s; A.st2[tid >> 3] = t2s; }
                }
            }
        }
        IHTP_STAMP(3);
        __syncthreads();
        // ---- P2 ----
        IHTP_STAMP(4);
        if (has2 && tid < IHTP_THREADS / 2) {
            if (A.R2 == 16) ihtp8_row_dots_par2<8>(A2, L.GB2, gl2, T, XS2, c2, tid, chain);
            else if (A.R2 == 32) ihtp8_row_dots_par2<4>(A2, L.GB2, gl2, T, XS2, c2, tid, chain);
            else ihtp8_row_dots_par2<2>(A2, L.GB2, gl2, T, XS2, c2, tid, chain);
        }
        IHTP_STAMP(5);
        __syncthreads();
        if (has2 && tid < A.R2)
            __hip_atomic_store((gu64 *)A.g2 + unit_slot(u2 + (tid >> 4), tid & 15, n),
                               ((u64)epoch << 32) | __float_as_uint(ihtp8_row_tree(chain + 8 * tid)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // ---- E2 ----
        int q3[8], qx[8];
        float t3s = 1.0f;
#pragma unroll
        for (int e = 0; e < 8; e++) q3[e] = qx[e] = 0;
        if (ST && gi2 >= 0) {                                            // behind the second row dots: t2's images and the bins are dead
            gb = ihtp_gen16(gb, raw2, (uint32_t)gi2 >> 2, (uint32_t)gi2 & 3u);
            gb = ihtp_cols_matvec(A.jump, gb);
        }
        {
            float d[8];
            if ((tid & ~63u) < n / 8) ihtp_gather8(A.g2, n, tid, epoch, tid < n / 8, A.nap0, A.nap, d);
            IHTP_STAMP(6);
            if (ST) __syncthreads();                                     // the second phase's draws are in LDS
            if (tid < n / 8) {
                uint32_t Wm[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u}, Ws[2] = {0u, 0u};
                if (ST) {
                    const uint32_t *r32 = reinterpret_cast<const uint32_t *>(raw2), b = tid >> 3, i = tid & 7u;
                    const u32x4 w0 = *reinterpret_cast<const u32x4 *>(r32 + (2 * b + (i >> 2)) * 8), w1 = *reinterpret_cast<const u32x4 *>(r32 + (2 * b + (i >> 2)) * 8 + 4);
                    Wm[0] = w0.x; Wm[1] = w0.y; Wm[2] = w0.z; Wm[3] = w0.w; Wm[4] = w1.x; Wm[5] = w1.y; Wm[6] = w1.z; Wm[7] = w1.w;
                    Ws[0] = r32[(2 * G2 + 2 * b + (i >> 2)) * 8 + 2 * (i & 3u)];
                    Ws[1] = r32[(2 * G2 + 2 * b + (i >> 2)) * 8 + 2 * (i & 3u) + 1];
                }
                ihtp8_requant_saa<ST>(d, xw, xs, A.mu, tid & 7u, Wm, Ws, q3, t3s, qx, xs);
            }
        }
        IHTP_STAMP(7);
        if (ST) {                                                        // the draws lay over the radix bins: clear those again
            __syncthreads();
            hist[tid] = 0;
            hist[IHTP_THREADS + tid] = 0;
            if (tid < 8 * IHTP8_HSTRIDE - 2048) hist[2048 + tid] = 0;
            __syncthreads();
        }
        if (A.threshold && A.K < A.x_len)
            ihtp8_threshold(qx, xs, tid, A.x_len, A.K, hist, hsum, wtot, A.dbg && it < 16 ? A.dbg + ((size_t)g * 16 + it) * 32 : nullptr);
        IHTP_STAMP(8);
        if (tid < n / 8) {
            xw[0] = pack4_i8(qx);
            xw[1] = pack4_i8(qx + 4);
            ihtp8_store_images(X, XS1, gl1, tid >> 3, tid & 7, xw[0], xw[1]);
            if ((tid & 7) == 0) c1[tid >> 3] = p1[tid >> 3] * (xs * (1.0f / 127.0f));
            if (last && g == 0) {
                A.t3[2 * tid] = pack4_i8(q3); A.t3[2 * tid + 1] = pack4_i8(q3 + 4);
                A.x[2 * tid] = xw[0]; A.x[2 * tid + 1] = xw[1];
                if ((tid & 7) == 0) { A.st3[tid >> 3] = t3s; A.sx[tid >> 3] = xs; }
            }
        }
        __syncthreads();
        IHTP_STAMP(9);
    }
    if (ST && g == 0) {                                                  // stamp the slot written in the last iteration (rng_device.h: rng_commit)
        if (tid0 == IHTP_THREADS - 4 * segs1) {
            __threadfence();
            A.rng[(rng_slot ^ 1) * RNG_SLOT_WORDS + RNG_STAMP_WORD] = rng_effective_seq(A.rng, A.seq);
        }
    }
}

// ---- host ---------------------------------------------------------------------------------------------------------------------------
static std::atomic<uint64_t> g_persist_launches{0};
extern "C" uint64_t clv_iht_persistent_launches(void) { return g_persist_launches.load(std::memory_order_relaxed); }

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

// the lock is held from the cross-stream wait to the launch: an event recorded for a later launch on another stream must lie BEHIND this one
struct PersistChain {
    std::unique_lock<std::mutex> lock;
    int rc;
    explicit PersistChain(hipStream_t st) : lock(g_persist_mutex), rc(begin(st)) {}
    static int begin(hipStream_t st)
    {
        int dev = 0;
        CLV_HIP(hipGetDevice(&dev));
        if (dev < 0 || dev >= 64) return CLV_OK;
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(st, &cs) != hipSuccess) { (void)hipGetLastError(); cs = hipStreamCaptureStatusNone; }
        if (cs != hipStreamCaptureStatusNone) return CLV_OK;              // inside a capture the graph's own edges order the launches
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
};

// stochastic rounding: an iteration draws D values per generator lane; a lane that has produced its segment of 16 jumps on by T^(D - 16)
// (T^(D - 17) towards the state the call leaves behind).  Column form, by square and multiply on the host (64 x 64 bits: microseconds)
static int ihtp_jump_matrices(uint64_t D, const u64 **seg_rows, u64 *jump, u64 *jump1)
{
    RngTables T;
    if (clv_rng_tables(&T)) return -1;
    *seg_rows = T.seg_rows;
    auto power = [](uint64_t e, u64 *out) {
        u64 sq[64], tmp[64];
        for (int i = 0; i < 64; i++) { sq[i] = xs_T(1ull << i); out[i] = 1ull << i; }
        for (; e; e >>= 1) {
            if (e & 1) { for (int i = 0; i < 64; i++) tmp[i] = gf2_matvec(sq, out[i]); for (int i = 0; i < 64; i++) out[i] = tmp[i]; }
            for (int i = 0; i < 64; i++) tmp[i] = gf2_matvec(sq, sq[i]);
            for (int i = 0; i < 64; i++) sq[i] = tmp[i];
        }
    };
    static std::mutex jump_mutex;                                         // the last D's matrices are kept: ~40 us of host time per call otherwise
    static uint64_t jump_D = 0;
    static u64 jump_kept[2][64];
    std::lock_guard<std::mutex> lock(jump_mutex);
    if (jump_D != D) {
        power(D - 16, jump_kept[0]);
        power(D - 17, jump_kept[1]);
        jump_D = D;
    }
    memcpy(jump, jump_kept[0], 64 * sizeof(u64));
    memcpy(jump1, jump_kept[1], 64 * sizeof(u64));
    return 0;
}

int clv_internal_persist_enter(hipStream_t stream)
{
    g_persist_mutex.lock();
    const int rc = PersistChain::begin(stream);
    if (rc) g_persist_mutex.unlock();
    return rc;
}
void clv_internal_persist_leave(void) { g_persist_mutex.unlock(); }

// returns 1 if the persistent kernel was launched, 0 if the problem does not qualify (the caller runs the launch-per-step loop), < 0 on error
int clm4_iht_persistent(const int8_t *Phi, const float *sPhi, const int8_t *PhiT, const float *sPhiT, uint64_t m, uint64_t n, int8_t *x,
                        float *sx, uint64_t x_len, const int8_t *y, const float *sy, int8_t *t1, float *st1, int8_t *t2, float *st2, int8_t *t3,
                        float *st3, uint64_t iterations, uint64_t K, float mu, int threshold, uint64_t *rng, hipStream_t st)
{
    const int mode = [] { const char *e = getenv("CLV_IHT_PERSISTENT"); return e ? atoi(e) : 1; }();      // read per call: A/B runs flip it
    if (!mode || threshold < 0 || threshold > 1 || !iterations || iterations >= 0x7FFFFFFFull) return 0;
    if (rng && (m + n) / 64 * 4 < 32) return 0;                             // stochastic: the per-iteration jumps T^(D - 16), T^(D - 17) want D >= 32
    if (m > IHTP_MAXLEN || n > IHTP_MAXLEN || m % 128 || n % 128 || !m || !n) return 0;
    const int cus = clv_cu_count();
    // rows per workgroup: 16 (one unit = one 128-byte line of granules), 32 or 64
    uint32_t R1 = pow2_ceil((uint32_t)((m + cus - 1) / cus)), R2 = pow2_ceil((uint32_t)((n + cus - 1) / cus));
    if (R1 < 16) R1 = 16;
    if (R2 < 16) R2 = 16;
    if (R1 > 64 || R2 > 64) return 0;
    const IhtpLayout L = ihtp_layout((uint32_t)m, (uint32_t)n, R1, R2, rng != nullptr);
    if (L.total > 160u * 1024u) return 0;
    const uint32_t grid = (uint32_t)((m / R1 > n / R2) ? m / R1 : n / R2);
    if ((int)grid > cus) return 0;

    static std::mutex attr_mutex;
    static uint32_t attr_set[64];
    int dev = 0;
    CLV_HIP(hipGetDevice(&dev));
    {
        std::lock_guard<std::mutex> lock(attr_mutex);
        if (dev >= 0 && dev < 64 && attr_set[dev] < L.total) {
            // a device that does not grant a workgroup 160 KiB of LDS does not run this kernel: the launch-per-step loop takes over
            if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_iht4_persist<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
                hipFuncSetAttribute(reinterpret_cast<const void *>(k_iht4_persist<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
                (void)hipGetLastError();
                return 0;
            }
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
    IhtpArgs a;
    a.Phi = (const uint8_t *)Phi; a.sPhi = sPhi; a.PhiT = (const uint8_t *)PhiT; a.sPhiT = sPhiT;
    a.m = (uint32_t)m; a.n = (uint32_t)n; a.x_len = (uint32_t)x_len; a.R1 = R1; a.R2 = R2;
    a.x = (uint32_t *)x; a.sx = sx; a.y = (const uint32_t *)y; a.sy = sy;
    a.t1 = (uint32_t *)t1; a.st1 = st1; a.t2 = (uint32_t *)t2; a.st2 = st2; a.t3 = (uint32_t *)t3; a.st3 = st3;
    a.iterations = (uint32_t)iterations; a.K = (uint32_t)(K > 0xFFFFFFFFull ? 0xFFFFFFFFull : K); a.mu = mu; a.threshold = threshold;
    a.g1 = (u64 *)ws; a.g2 = (u64 *)ws + m;
    a.rng = rng;
    a.seq = 0;
    a.seg_rows = nullptr;
    if (rng) {
        if (ihtp_jump_matrices((m + n) / 64 * 4, &a.seg_rows, a.jump, a.jump1)) return -1;
    }
    a.nap0 = 14;         // ~0.4 us: measured best of 0 / 6 / 10 / 14 / 18 / 22 / 26 ... 40 at N = 8192 (profiles/r06_iht_persist_notes.txt)
    a.nap = 2;
    if (const char *e = getenv("CLV_IHT_NAP0")) a.nap0 = (uint32_t)atoi(e);                             // probe only
    if (const char *e = getenv("CLV_IHT_NAP")) a.nap = (uint32_t)atoi(e);                               // probe only
    a.dbg = nullptr;
    if (const char *e = getenv("CLV_IHT_DEBUG_STAMPS")) a.dbg = (u64 *)strtoull(e, nullptr, 0);      // probe only: a device buffer of grid * 16 * 32 words
    PersistChain chain(st);
    if (chain.rc) return -1;
    if (rng) {
        a.seq = clv_rng_seq_for(rng, st);                                  // right in front of the launch, as every stochastic call
        hipLaunchKernelGGL(k_iht4_persist<true>, dim3(grid), dim3(IHTP_THREADS), L.total, st, a);
    } else {
        hipLaunchKernelGGL(k_iht4_persist<false>, dim3(grid), dim3(IHTP_THREADS), L.total, st, a);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        clv_set_error("clm4_iht: persistent launch failed: %s", hipGetErrorString(e));
        return -1;
    }
    g_persist_launches.fetch_add(1, std::memory_order_relaxed);
    return 1;
}

// the CloverVector8 loop (clm4_iht_v8): 1 = launched, 0 = does not qualify, < 0 = error
int clm4_iht_v8_persistent(const int8_t *Phi, const float *sPhi, const int8_t *PhiT, const float *sPhiT, uint64_t m, uint64_t n, int8_t *x, float *sx,
                           uint64_t x_len, const int8_t *y, const float *sy, int8_t *t1, float *st1, int8_t *t2, float *st2, int8_t *t3, float *st3,
                           uint64_t iterations, uint64_t K, float mu, int threshold, uint64_t *rng, hipStream_t st)
{
    const int mode = [] { const char *e = getenv("CLV_IHT_PERSISTENT"); return e ? atoi(e) : 1; }();
    if (!mode || threshold < 0 || threshold > 1 || !iterations || iterations >= 0x7FFFFFFFull) return 0;
    if (rng && (m + n) / 64 * 4 < 32) return 0;
    if (m > IHTP_MAXLEN || n > IHTP_MAXLEN || m % 128 || n % 128 || !m || !n) return 0;
    const int cus = clv_cu_count();
    uint32_t R1 = pow2_ceil((uint32_t)((m + cus - 1) / cus)), R2 = pow2_ceil((uint32_t)((n + cus - 1) / cus));
    if (R1 < 16) R1 = 16;
    if (R2 < 16) R2 = 16;
    if (R1 > 64 || R2 > 64) return 0;
    const Ihtp8Layout L = ihtp8_layout((uint32_t)m, (uint32_t)n, R1, R2);
    if (L.total > 160u * 1024u) return 0;
    if (L.GB1 * R1 > 512u || L.GB2 * R2 > 512u) return 0;               // the row dots deal a chain over 128 / R lanes, 4 groups each at most
    // stochastic: the raw draws overlay LDS regions that are dead while they are needed (ihtp8_layout); a shape whose draws do not fit there
    // (the first phase's where m > 0.75 n: gradient descent's 1.5 : 1 systems) runs the launch-per-step loop
    if (rng && (L.raw1_bytes > L.raw1_room || L.raw2_bytes > L.raw2_room)) return 0;
    const uint32_t grid = (uint32_t)((m / R1 > n / R2) ? m / R1 : n / R2);
    if ((int)grid > cus) return 0;
    static std::mutex attr_mutex;
    static bool attr_set[64];
    int dev = 0;
    CLV_HIP(hipGetDevice(&dev));
    {
        std::lock_guard<std::mutex> lock(attr_mutex);
        if (dev >= 0 && dev < 64 && !attr_set[dev]) {
            if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_iht8_persist<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
                hipFuncSetAttribute(reinterpret_cast<const void *>(k_iht8_persist<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
                (void)hipGetLastError();
                return 0;
            }
            attr_set[dev] = true;
        }
    }
    void *ws = nullptr;
    if (clv_internal_workspace(&ws, (m + n) * sizeof(u64), st)) return -1;
    if (hipMemsetAsync(ws, 0, (m + n) * sizeof(u64), st) != hipSuccess) {
        clv_set_error("clm4_iht_v8: hipMemsetAsync failed: %s", hipGetErrorString(hipGetLastError()));
        return -1;
    }
    Ihtp8Args a;
    a.Phi = (const uint8_t *)Phi; a.sPhi = sPhi; a.PhiT = (const uint8_t *)PhiT; a.sPhiT = sPhiT;
    a.m = (uint32_t)m; a.n = (uint32_t)n; a.x_len = (uint32_t)x_len; a.R1 = R1; a.R2 = R2;
    a.x = (uint32_t *)x; a.sx = sx; a.y = (const uint32_t *)y; a.sy = sy;
    a.t1 = (uint32_t *)t1; a.st1 = st1; a.t2 = (uint32_t *)t2; a.st2 = st2; a.t3 = (uint32_t *)t3; a.st3 = st3;
    a.iterations = (uint32_t)iterations; a.K = (uint32_t)(K > 0xFFFFFFFFull ? 0xFFFFFFFFull : K); a.mu = mu; a.threshold = threshold;
    a.g1 = (u64 *)ws; a.g2 = (u64 *)ws + m;
    a.nap0 = 16;         // best of 6 ... 26 for this kernel at N = 8192
    a.nap = 2;
    if (const char *e = getenv("CLV_IHT_NAP0")) a.nap0 = (uint32_t)atoi(e);                             // probe only
    a.dbg = nullptr;
    if (const char *e = getenv("CLV_IHT_DEBUG_STAMPS")) a.dbg = (u64 *)strtoull(e, nullptr, 0);      // probe only: a device buffer of grid * 16 * 32 words
    a.rng = rng;
    a.seq = 0;
    a.seg_rows = nullptr;
    if (rng && ihtp_jump_matrices((m + n) / 64 * 4, &a.seg_rows, a.jump, a.jump1)) return -1;
    PersistChain chain(st);
    if (chain.rc) return -1;
    if (rng) {
        a.seq = clv_rng_seq_for(rng, st);
        hipLaunchKernelGGL(k_iht8_persist<true>, dim3(grid), dim3(IHTP_THREADS), L.total, st, a);
    } else {
        hipLaunchKernelGGL(k_iht8_persist<false>, dim3(grid), dim3(IHTP_THREADS), L.total, st, a);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        clv_set_error("clm4_iht_v8: persistent launch failed: %s", hipGetErrorString(e));
        return -1;
    }
    g_persist_launches.fetch_add(1, std::memory_order_relaxed);
    return 1;
}
