// common.h -- shared host/device helpers of libclover_hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "clover_hip.h"

// ---- error plumbing ---------------------------------------------------------------------------
void clv_set_error(const char *fmt, ...);

#define CLV_HIP(call)                                                                       \
    do {                                                                                    \
        hipError_t e__ = (call);                                                            \
        if (e__ != hipSuccess) {                                                            \
            clv_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__, \
                          __LINE__);                                                        \
            return CLV_ERR_HIP;                                                             \
        }                                                                                   \
    } while (0)

#define CLV_REQUIRE(cond, ...)           \
    do {                                 \
        if (!(cond)) {                   \
            clv_set_error(__VA_ARGS__);  \
            return CLV_ERR_INVALID;      \
        }                                \
    } while (0)

#define CLV_LAUNCH_CHECK()                                                             \
    do {                                                                               \
        hipError_t e__ = hipGetLastError();                                            \
        if (e__ != hipSuccess) {                                                       \
            clv_set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(e__),  \
                          __FILE__, __LINE__);                                         \
            return CLV_ERR_HIP;                                                        \
        }                                                                              \
    } while (0)

static inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

// number of compute units of the current device (cached per device)
int clv_cu_count();
// grow-only per-device scratch buffer used when the caller passes workspace == NULL
int clv_internal_workspace(void **ptr, uint64_t bytes, hipStream_t stream);      // grow-only scratch per (device, stream)
void clv_internal_workspace_forget(hipStream_t stream);
// the zero-initialised hand-over memory of a (device, stream), clv_internal_sync_slots: the single-launch reductions (dot_common.h) use the
// first 64 KiB, the large-vector threshold's control block (threshold4.hip) lies behind them; every user leaves its part all zero
#define CLV_SYNC_SLOT_BYTES_TOTAL (128u << 10)
#define CLV_SYNC_SLOT_THRESHOLD_OFFSET (64u << 10)
// persistent launches (workgroups that wait for each other) run one at a time per device: enter chains the stream behind the previous such
// launch and holds the chain's lock until leave (iht_persist.hip)
int clv_internal_persist_enter(hipStream_t stream);
void clv_internal_persist_leave(void);
void clv_internal_persist_forget(hipStream_t stream);     // iht_persist.hip: the chain of persistent launches forgets a destroyed stream
// the exact-order chain kernel of vector4.hip for other sources (mixed8.hip: CloverVector8::dot): see there for the layout of X
uint64_t clv_internal_dot_chain_blocks_padded(uint64_t steps);
uint64_t clv_internal_dot_chain_bytes(uint64_t steps);
int clv_internal_dot_chain(const void *X, uint64_t blocks_padded, float *out_dev, hipStream_t st);
// zero-initialised hand-over slots per (device, stream), <= 64 KiB; every user leaves them zero again (runtime.hip)
int clv_internal_sync_slots(void **ptr, uint64_t bytes, hipStream_t stream);

// ---- device helpers ---------------------------------------------------------------------------
#define CLV_RCP49 (1.0f / 49.0f)   // 0x3CA72F05, the reference's clover_mm256_rcp_49_ps (CloverBase.h:88)

// v_dot8_i32_i4: sum of the 8 signed-nibble products of two dwords, plus c.  Both operands use the
// same nibble order inside the word, so this IS the reference's per-32-bit-word integer.
__device__ __forceinline__ int sdot8(uint32_t a, uint32_t b, int c)
{
    return __builtin_amdgcn_sdot8((int)a, (int)b, c, false);
}

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int dot32(const u32x4 &a, const u32x4 &b)
{
    int s = sdot8(a.x, b.x, 0);
    s = sdot8(a.y, b.y, s);
    s = sdot8(a.z, b.z, s);
    return sdot8(a.w, b.w, s);
}

// x / 7.0f, correctly rounded, in 4 VALU instead of the ~12 of the IEEE division sequence (v_div_scale x2, v_rcp, 4 fma,
// v_div_fmas, v_div_fixup): q = x * c; r = fma(-7, q, x) (exact residual); q += r * c -- the classic division by a constant
// known in advance (Brisebarre, Muller, Raina 2004).  Checked EXHAUSTIVELY against x / 7.0f for all 2^32 bit patterns
// (tools/check_div7.c): identical for every finite x except x = -0 (gives +0), which the sign copy at the end repairs.
// NOT identical for x = +/-inf: the residual fma(-7, inf, inf) is NaN, so div7(inf) = NaN where inf / 7.0f = inf.  Callers pass block
// scales (absolute maxima of the data); a block with an infinite element is outside the reference's contract already (its quantiser
// feeds 7/inf = 0 and inf * 0 = NaN to cvttps, CloverVector4.h:668-685) and DESIGN.md 4 lists non-finite inputs as out of contract,
// so the callers do not guard (a guard would cost the streaming kernels one VALU instruction per block word).
__device__ __forceinline__ float div7(float x)
{
    const float c = 1.0f / 7.0f;
    const float q = x * c;
    const float r = __builtin_fmaf(-7.0f, q, x);
    return __builtin_copysignf(__builtin_fmaf(r, c, q), x);
}
// the same for CloverVector8's x / 127.0f (checked the same way: tools/check_div7.c 127)
__device__ __forceinline__ float div127(float x)
{
    const float c = 1.0f / 127.0f;
    const float q = x * c;
    const float r = __builtin_fmaf(-127.0f, q, x);
    return __builtin_copysignf(__builtin_fmaf(r, c, q), x);
}

// 0 -> 1.0 on the bit pattern (CloverVector4.h:661-663)
__device__ __forceinline__ float fix_zero_max(float m)
{
    return __float_as_uint(m) == 0u ? m + 1.0f : m;
}

// One element of the quantiser.  The reference computes trunc(fma(|x|, k, noise)) and re-applies the sign of x
// (_mm256_sign_epi32, CloverVector4.h:741-772).  Rounding is sign-symmetric, so the same integer comes out of
// the SIGNED product: RN(x*k + copysign(noise, x)) == copysign(RN(|x|*k + noise), x), and the float->int
// conversion truncates toward zero for either sign.  That is one multiply + one convert per element
// (v_cvt_i32_f32 truncates like cvttps).  The only input for which cvttps' out-of-range answer (0x80000000,
// i.e. nibble 0) differs from v_cvt's saturation is k == inf (block maximum below 2.06e-38, 7/max overflows):
// there every element becomes 0 in the reference, which the callers mirror per block with `k < INFINITY`.
// copysign(noise, x) for noise >= +0: noise | (x & 0x80000000) as ONE v_bitop3_b32 ((s0 & s1) | s2 = table 0xEA).  hipcc emits
// v_bfi_b32 for copysign, which occupies the VALU 4.3 cycles per wave; v_bitop3_b32 takes 2.9 like v_fma_f32 (tools/valu_rate.hip:
// v_xor 2.4, v_fma / v_bitop3 2.9, conversions / v_bfi / v_max3 / 64-bit shifts 4.3-4.9, v_pk_fma_f32 5.1).
__device__ __forceinline__ float sign_onto_nonneg(float noise, float x)
{
    float r;
    asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0xea" : "=v"(r) : "v"(x), "s"(0x80000000u), "v"(noise));
    return r;
}
__device__ __forceinline__ int quant1_det(float x, float k) { return (int)(x * k); }
__device__ __forceinline__ int quant1_st(float x, float k, float noise)
{
    return (int)__builtin_fmaf(x, k, sign_onto_nonneg(noise, x));
}
// single-element form with the overflow guard folded in (mvm epilogue: one value per lane)
__device__ __forceinline__ int quant1(float x, float k, float noise)
{
    const int t = quant1_st(x, k, noise);
    return k < __builtin_inff() ? t : 0;
}

// bit position of element e (0..7) of a little-endian 32-bit word: even elements sit in the HIGH nibble
__device__ __forceinline__ int nib_shift(int e) { return 8 * (e >> 1) + ((e & 1) ? 0 : 4); }

__device__ __forceinline__ uint32_t pack8(const int q[8])
{
    uint32_t w = 0;
#pragma unroll
    for (int e = 0; e < 8; e++) w |= ((uint32_t)q[e] & 0xFu) << nib_shift(e);
    return w;
}

// 8 quantised integers in [-7,7] -> one word: byte i = (q[2i] << 4) | (q[2i+1] & 0xF), the bytes gathered with v_perm_b32
// (11 VALU instead of 16).  The upper 24 bits of the per-byte intermediates hold sign-extension garbage, which v_perm drops.
__device__ __forceinline__ uint32_t pack8_perm(const int q[8])
{
    uint32_t b[4];
#pragma unroll
    for (int i = 0; i < 4; i++) b[i] = ((uint32_t)q[2 * i] << 4) | ((uint32_t)q[2 * i + 1] & 0xFu);
    const uint32_t lo = __builtin_amdgcn_perm(b[1], b[0], 0x0C0C0400u);      // [0, 0, b1.byte0, b0.byte0]
    const uint32_t hi = __builtin_amdgcn_perm(b[3], b[2], 0x04000C0Cu);      // [b3.byte0, b2.byte0, 0, 0]
    return lo | hi;
}

// float -> int (v_cvt_i32_f32: truncation toward zero, like the (int) cast) with the LOW BYTE of the result deposited straight into
// byte K of an accumulating register (SDWA destination select): the quantised integers of a word never exist as separate
// registers, and packing them costs 2 VALU per 8 elements instead of 11.  CVT_BYTE0_FIRST zeroes the other bytes (no read of acc).
#define CLV_CVT_INTO_BYTE(K)                                                                                                          \
    __device__ __forceinline__ void cvt_i32_into_byte##K(uint32_t &acc, float t)                                                       \
    {                                                                                                                                 \
        asm("v_cvt_i32_f32_sdwa %0, %1 dst_sel:BYTE_" #K " dst_unused:UNUSED_PRESERVE src0_sel:DWORD" : "+v"(acc) : "v"(t));         \
    }
CLV_CVT_INTO_BYTE(1)
CLV_CVT_INTO_BYTE(2)
CLV_CVT_INTO_BYTE(3)
#undef CLV_CVT_INTO_BYTE
__device__ __forceinline__ uint32_t cvt_i32_byte0_first(float t)
{
    uint32_t acc;
    asm("v_cvt_i32_f32_sdwa %0, %1 dst_sel:BYTE_0 dst_unused:UNUSED_PAD src0_sel:DWORD" : "=v"(acc) : "v"(t));
    return acc;
}
// even = bytes q[0], q[2], q[4], q[6]; odd = bytes q[1], q[3], q[5], q[7] (two's complement): byte i of the word is
// (q[2i] << 4) | (q[2i+1] & 0xF).  The shift drags each byte's top nibble into its neighbour's low nibble; v_bfi takes the low
// nibbles from `odd` instead.
__device__ __forceinline__ uint32_t nibbles_from_bytes(uint32_t even, uint32_t odd)
{
    return ((even << 4) & 0xF0F0F0F0u) | (odd & 0x0F0F0F0Fu);      // v_lshlrev_b32 + v_bfi_b32
}

typedef float f32x2 __attribute__((ext_vector_type(2)));

// quantise + pack 8 consecutive elements (one output dword); noise == nullptr <=> rounding disabled
__device__ __forceinline__ uint32_t quant_pack8(const float v[8], float k, const float *noise)
{
    float t[8];
    if (noise) {
        // the stochastic products as four v_pk_fma_f32 (hipcc packs the deterministic multiplies itself, but not the fmas whose addend
        // comes out of the v_bitop3 asm): the same fused, singly rounded fma per lane, half the issue slots
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
            const f32x2 r = __builtin_elementwise_fma(f32x2{v[e], v[e + 1]}, f32x2{k, k},
                                                      f32x2{sign_onto_nonneg(noise[e], v[e]), sign_onto_nonneg(noise[e + 1], v[e + 1])});
            t[e] = r.x;
            t[e + 1] = r.y;
        }
    } else {
#pragma unroll
        for (int e = 0; e < 8; e++) t[e] = v[e] * k;                                                                      // quant1_det
    }
    uint32_t even = cvt_i32_byte0_first(t[0]), odd = cvt_i32_byte0_first(t[1]);
    cvt_i32_into_byte1(even, t[2]);
    cvt_i32_into_byte1(odd, t[3]);
    cvt_i32_into_byte2(even, t[4]);
    cvt_i32_into_byte2(odd, t[5]);
    cvt_i32_into_byte3(even, t[6]);
    cvt_i32_into_byte3(odd, t[7]);
    return k < __builtin_inff() ? nibbles_from_bytes(even, odd) : 0u;
}

// 4 products (already multiplied by k, noise added) -> the 16 bits of half an output dword (elements 4h..4h+3 -> bytes 0, 1)
__device__ __forceinline__ uint32_t pack4_of_products(float t0, float t1, float t2, float t3)
{
    uint32_t even = cvt_i32_byte0_first(t0), odd = cvt_i32_byte0_first(t1);
    cvt_i32_into_byte1(even, t2);
    cvt_i32_into_byte1(odd, t3);
    return nibbles_from_bytes(even, odd);
}

// maximum over the 16 lanes of a DPP row (rotations inside the row), result in every lane
__device__ __forceinline__ float row16_max(float m)
{
#define ROR_MAX(n) m = fmaxf(m, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(m), 0x120 + n, 0xF, 0xF, false)))
    ROR_MAX(8);
    ROR_MAX(4);
    ROR_MAX(2);
    ROR_MAX(1);
#undef ROR_MAX
    return m;
}

// the 4 nibbles of half h of an output dword (elements 4h..4h+3 -> bytes 2h, 2h+1, even elements in the high nibble)
__device__ __forceinline__ uint32_t quant_pack4(const f32x4 v, float k)
{
    const uint32_t h = pack4_of_products(v.x * k, v.y * k, v.z * k, v.w * k);
    return k < __builtin_inff() ? h : 0u;       // see quant_pack8
}

// 4x4 transpose of dwords inside every quad of lanes (m = lane & 3): lane m ends with component m of lanes 0..3, i.e. after four
// lanes loaded 64 contiguous bytes as one dwordx4 each, lane m holds words m, 4+m, 8+m, 12+m.  Two DPP exchanges, 16 VALU.
__device__ __forceinline__ void quad_transpose4(uint32_t &v0, uint32_t &v1, uint32_t &v2, uint32_t &v3, int m)
{
    {   // bit 0: (lane j, comp i) <-> (lane j^1, comp i^1) where the low bits differ
        const bool b = m & 1;
        const uint32_t s0 = b ? v0 : v1, s1 = b ? v2 : v3;
        const uint32_t r0 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)s0, 0xB1, 0xF, 0xF, false);    // quad_perm [1,0,3,2]
        const uint32_t r1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)s1, 0xB1, 0xF, 0xF, false);
        v0 = b ? r0 : v0; v1 = b ? v1 : r0; v2 = b ? r1 : v2; v3 = b ? v3 : r1;
    }
    {   // bit 1
        const bool b = m & 2;
        const uint32_t s0 = b ? v0 : v2, s1 = b ? v1 : v3;
        const uint32_t r0 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)s0, 0x4E, 0xF, 0xF, false);    // quad_perm [2,3,0,1]
        const uint32_t r1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)s1, 0x4E, 0xF, 0xF, false);
        v0 = b ? r0 : v0; v1 = b ? r1 : v1; v2 = b ? v2 : r0; v3 = b ? v3 : r1;
    }
}

// ---- cheaper nibble <-> float conversions for the VALU-bound kernels ------------------------------------------------------
// The 8 nibbles of a word as floats, SIXTEEN TIMES their value: f[e] = 16 * q_e (exact).  `w & 0xF0F0F0F0` leaves the even
// elements as signed bytes 16 q, `(w << 4) & 0xF0F0F0F0` the odd ones, and v_cvt_f32_i32 with an SDWA byte select + sign
// extension converts a byte in ONE instruction: 11 VALU per word instead of 16 (v_bfe_i32 + v_cvt_f32_i32 per nibble).  The
// caller folds the 1/16 into its scale (exact: a power of two) -- see scaled_by_16th().  hipcc does not form these SDWA
// conversions itself (it splits them into v_and_b32_sdwa + v_cvt), hence the asm; it is not volatile, hipcc may schedule it.
#define CLV_SBYTE_TO_F32(K)                                                                                                     \
    __device__ __forceinline__ float sbyte##K##_to_f32(uint32_t w)                                                              \
    {                                                                                                                           \
        float f;                                                                                                                \
        asm("v_cvt_f32_i32_sdwa %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_" #K : "=v"(f) : "v"(w));     \
        return f;                                                                                                               \
    }
CLV_SBYTE_TO_F32(0)
CLV_SBYTE_TO_F32(1)
CLV_SBYTE_TO_F32(2)
CLV_SBYTE_TO_F32(3)
#undef CLV_SBYTE_TO_F32

__device__ __forceinline__ void unpack8_x16(uint32_t w, float f[8])
{
    const uint32_t even = w & 0xF0F0F0F0u, odd = (w << 4) & 0xF0F0F0F0u;
    f[0] = sbyte0_to_f32(even); f[1] = sbyte0_to_f32(odd);
    f[2] = sbyte1_to_f32(even); f[3] = sbyte1_to_f32(odd);
    f[4] = sbyte2_to_f32(even); f[5] = sbyte2_to_f32(odd);
    f[6] = sbyte3_to_f32(even); f[7] = sbyte3_to_f32(odd);
}

// The same 8 nibbles as floats of ONE SIXTEENTH their value, f[e] = q_e / 16 (exact), one instruction per element and no masks:
// v_cvt_off_f32_i4 converts the signed 4-bit integer in bits [3:0] of its source (the hardware's interpolation-offset table: 1000 ->
// -0.5 ... 0111 -> 0.4375) and the SDWA source select hands it byte K -- the low nibble of byte K is element 2K+1, and after ONE shift
// of the word by 4 the same select reaches the high nibbles (elements 2K): 9 VALU per word instead of 11.  The caller folds the 16
// into its scale: (q / 16) * (16 s) is the same real number as q * s and 16 s is exact unless it overflows (times16_is_finite), so the
// product rounds identically -- also for denormal scales, where the s / 16 form above has to take the plain path.
#define CLV_LOWNIB_16TH(K)                                                                                                       \
    __device__ __forceinline__ float lownib##K##_16th(uint32_t w)                                                                \
    {                                                                                                                           \
        float f;                                                                                                                \
        asm("v_cvt_off_f32_i4_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_" #K : "=v"(f) : "v"(w));          \
        return f;                                                                                                               \
    }
CLV_LOWNIB_16TH(0)
CLV_LOWNIB_16TH(1)
CLV_LOWNIB_16TH(2)
CLV_LOWNIB_16TH(3)
#undef CLV_LOWNIB_16TH

__device__ __forceinline__ void unpack8_16th(uint32_t w, float f[8])
{
    const uint32_t hi = w >> 4;
    f[0] = lownib0_16th(hi); f[1] = lownib0_16th(w);
    f[2] = lownib1_16th(hi); f[3] = lownib1_16th(w);
    f[4] = lownib2_16th(hi); f[5] = lownib2_16th(w);
    f[6] = lownib3_16th(hi); f[7] = lownib3_16th(w);
}
__device__ __forceinline__ bool times16_is_finite(float s) { return __builtin_fabsf(s * 16.0f) < __builtin_inff(); }

// s / 16 is exact unless it falls into the denormals: true when (16 q) * (s / 16) rounds exactly like q * s
__device__ __forceinline__ bool sixteenth_is_exact(float s)
{
    const float a = __builtin_fabsf(s);
    return a >= 0x1p-120f || a == 0.0f;
}

// signed nibble e of word w
__device__ __forceinline__ int unpack1(uint32_t w, int e)
{
    return ((int)(w << (28 - nib_shift(e)))) >> 28;
}

__device__ __forceinline__ float wave_max(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// splitmix64 finaliser: counter-based generator for synthetic data
__host__ __device__ __forceinline__ uint64_t splitmix64(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
