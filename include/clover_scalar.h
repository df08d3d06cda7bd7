/*
 * clover_scalar.h -- the `_scalar` validation partners of the reference, as scalar HOST code.
 *
 * In the reference every hot method has a plain-C twin (`quantize_scalar`, `restore_scalar`, `dot_scalar`, `scaleAndAdd_scalar`,
 * `mvm_scalar`, `transpose_scalar`; CloverVector4.h:336-595, CloverMatrix4.h:178-502) whose only job is to be compared with the SIMD
 * method by the validation harness (test/validate/02_vector.cpp, 03_matrix.cpp).  They keep that job here: the containers' `_scalar`
 * methods run the loops below on the host copy, so "device kernel == scalar twin" is a comparison of two independent
 * implementations again (round 1 aliased them to the kernels, which made the relation vacuous).  NO hot method calls anything in
 * this file, and nothing here is a fallback: without the HIP library the containers still fail loudly.
 *
 * Own code following the reference's scalar arithmetic:
 *   quantise: m = max |x| over the block; k = 7 / m; q = sign(x) * floor(fma(|x|, k, noise))          (CloverVector4.h:452-517)
 *             (an all-zero block divides by zero in the reference's scalar code (:478-479); the SIMD contract -- scale 1.0,
 *              nibbles 0 -- is used instead, as the drop-in containers document)
 *   noise   : 0 when CLOVER_STOCHASTIC_ROUNDING_DISABLED, else uniform [0,1) from a per-thread generator seeded by
 *             std::random_device (the reference draws from RDRAND here: unseedable, never comparable with the SIMD stream)
 *   restore : (scale / 7) * q                                                                          (:519-553)
 *   scaleAndAdd: val = fma((float)qv, f32(f32(sv * a) / 7), (float)qu * f32(su / 7)), then the quantiser (:336-449)
 */
#ifndef CLOVER_SCALAR_H
#define CLOVER_SCALAR_H

#include <cmath>
#include <cstdint>
#include <cstring>
#include <random>

namespace clover_hip {
namespace scalar {

inline float noise()
{
#ifdef CLOVER_STOCHASTIC_ROUNDING_DISABLED
    return 0.0f;
#else
    static thread_local std::mt19937 gen{std::random_device{}()};
    return (float)(gen() >> 8) * (1.0f / 16777216.0f);
#endif
}

inline int8_t nibble_hi(int8_t b) { return (int8_t)(b >> 4); }
inline int8_t nibble_lo(int8_t b) { return (int8_t)((int8_t)(b << 4) >> 4); }

/* one element: floor(fma(|x|, k, noise)) with the sign of x's bit pattern re-applied */
inline int8_t quant1(float x, float k)
{
    uint32_t bits;
    std::memcpy(&bits, &x, 4);
    const int8_t sgn = (bits >> 31) ? -1 : 1;
    const float mag = std::floor(std::fma(std::fabs(x), k, noise()));
    return (int8_t)((int8_t)mag * sgn);
}

/* 64 values (any stride between them) -> 32 packed bytes at dst; returns the stored scale */
inline float quantize_block4(const float *x, int8_t *dst)
{
    float m = 0.0f;
    for (int i = 0; i < 64; i++) { const float a = std::fabs(x[i]); if (a > m) m = a; }
    if (m == 0.0f) m = 1.0f;
    const float k = 7.0f / m;
    for (int i = 0; i < 64; i += 2) dst[i >> 1] = (int8_t)((quant1(x[i], k) << 4) | (quant1(x[i + 1], k) & 0xF));
    return m;
}

inline void quantize4(const float *x, uint64_t n_pad, int8_t *q, float *s)
{
    for (uint64_t b = 0; b < n_pad / 64; b++) s[b] = quantize_block4(x + 64 * b, q + 32 * b);
}

inline void restore4(const int8_t *q, const float *s, uint64_t n_pad, float *x)
{
    for (uint64_t b = 0; b < n_pad / 64; b++) {
        const float r = s[b] / 7.0f;
        for (uint64_t i = 0; i < 32; i++) {
            x[64 * b + 2 * i] = r * (float)nibble_hi(q[32 * b + i]);
            x[64 * b + 2 * i + 1] = r * (float)nibble_lo(q[32 * b + i]);
        }
    }
}

/* r = quantize(u + a * v), block by block; r may alias u */
inline void scale_and_add4(const int8_t *u, const float *su, const int8_t *v, const float *sv, float a, uint64_t n_pad, int8_t *r, float *sr)
{
    float block[64];
    for (uint64_t b = 0; b < n_pad / 64; b++) {
        const float ur = su[b] / 7.0f, vr = sv[b] * a / 7.0f;
        for (uint64_t i = 0; i < 32; i++) {
            const int8_t bu = u[32 * b + i], bv = v[32 * b + i];
            block[2 * i] = std::fma((float)nibble_hi(bv), vr, ur * (float)nibble_hi(bu));
            block[2 * i + 1] = std::fma((float)nibble_lo(bv), vr, ur * (float)nibble_lo(bu));
        }
        sr[b] = quantize_block4(block, r + 32 * b);
    }
}

/* CloverVector8 twins (CloverVector8.h:205-253 quantize, :255-290 restore, :292-340 scaleAndAdd): scale 127, one int8 per element */
inline float quantize_block8(const float *x, int8_t *dst)
{
    float m = 0.0f;
    for (int i = 0; i < 64; i++) { const float a = std::fabs(x[i]); if (a > m) m = a; }
    if (m == 0.0f) m = 1.0f;
    const float k = 127.0f / m;
    for (int i = 0; i < 64; i++) {
        uint32_t bits;
        std::memcpy(&bits, &x[i], 4);
        const float mag = std::floor(std::fma(std::fabs(x[i]), k, noise()));
        dst[i] = (int8_t)((bits >> 31) ? -(int)mag : (int)mag);
    }
    return m;
}
inline void quantize8(const float *x, uint64_t n_pad, int8_t *q, float *s)
{
    for (uint64_t b = 0; b < n_pad / 64; b++) s[b] = quantize_block8(x + 64 * b, q + 64 * b);
}
inline void restore8(const int8_t *q, const float *s, uint64_t n_pad, float *x)
{
    for (uint64_t b = 0; b < n_pad / 64; b++) {
        const float r = s[b] / 127.0f;
        for (uint64_t i = 0; i < 64; i++) x[64 * b + i] = r * (float)q[64 * b + i];
    }
}
inline void scale_and_add8(const int8_t *u, const float *su, const int8_t *v, const float *sv, float a, uint64_t n_pad, int8_t *r, float *sr)
{
    float block[64];
    for (uint64_t b = 0; b < n_pad / 64; b++) {
        const float ur = su[b] / 127.0f, vr = sv[b] * a / 127.0f;
        for (uint64_t i = 0; i < 64; i++) block[i] = std::fma((float)v[64 * b + i], vr, ur * (float)u[64 * b + i]);
        sr[b] = quantize_block8(block, r + 64 * b);
    }
}

}  // namespace scalar
}  // namespace clover_hip

#endif
