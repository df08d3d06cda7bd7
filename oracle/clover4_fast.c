/*
 * clover4_fast.c -- AVX2 + OpenMP restatement of the 4-bit hot path: the timed CPU baseline ("port").
 * TEST / BENCH INFRASTRUCTURE ONLY -- never linked into libclover_hip.so.
 *
 * Same results, bit for bit, as the scalar oracle (clover4_oracle.c) and therefore as the reference's
 * AVX2 path (rounding disabled); tests/test_oracle_properties.py asserts that.  It plays the role of
 * "Clover's own AVX2 path on the host cores" next to the GPU numbers, because the reference sources
 * can neither be built in this image (IPP/MKL headers) nor travel to the GPU box.
 *
 * Parallel decomposition mirrors the reference's OpenMP variants: mvm splits 64-row blocks contiguously
 * over threads (CloverMatrix4.h:1700-1705), quantize splits 64-element blocks (CloverVector4.h:818-828);
 * dot stays sequential because its fp32 order is part of the result (dot_parallel is not reproducible).
 *
 * Two interchangeable kernels for the per-word integer sums (orcf_set_kernel), both own code:
 *   1 "maddubs" (default, the timed one): the reference's INSTRUCTION MIX (CloverVector4.h:1136-1180) -- nibbles kept as
 *     16x-scaled bytes (and 0xF0 / shift-left-4), |u| and sign(v, u) so that vpmaddubsw can multiply unsigned x signed,
 *     arithmetic shift right 8 (the products are multiples of 256), add the two planes, vpmaddwd with ones -> int32 per word;
 *   0 "planes": nibbles sign-extended into four int16 planes with shifts, multiplied with vpmaddwd.
 * Same integers either way (tests/test_oracle_properties.py), so the fp32 results are bit-identical.
 */
#include <immintrin.h>
#include <math.h>
#include <stdint.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static const float RCP49 = 1.0f / 49.0f;

int orcf_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void orcf_set_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* exact int32 sums of the 8 nibble products of each of the 8 words of a 32-byte block */
static inline __m256i block_word_isums(__m256i u, __m256i v)
{
    /* a 16-bit lane holds 4 nibbles: bits 12-15, 8-11, 4-7, 0-3; plane k = sign-extended nibble k */
    const __m256i u3 = _mm256_srai_epi16(u, 12);
    const __m256i v3 = _mm256_srai_epi16(v, 12);
    const __m256i u2 = _mm256_srai_epi16(_mm256_slli_epi16(u, 4), 12);
    const __m256i v2 = _mm256_srai_epi16(_mm256_slli_epi16(v, 4), 12);
    const __m256i u1 = _mm256_srai_epi16(_mm256_slli_epi16(u, 8), 12);
    const __m256i v1 = _mm256_srai_epi16(_mm256_slli_epi16(v, 8), 12);
    const __m256i u0 = _mm256_srai_epi16(_mm256_slli_epi16(u, 12), 12);
    const __m256i v0 = _mm256_srai_epi16(_mm256_slli_epi16(v, 12), 12);
    /* vpmaddwd adds the two 16-bit products of every 32-bit word */
    const __m256i p32 = _mm256_add_epi32(_mm256_madd_epi16(u3, v3), _mm256_madd_epi16(u2, v2));
    const __m256i p10 = _mm256_add_epi32(_mm256_madd_epi16(u1, v1), _mm256_madd_epi16(u0, v0));
    return _mm256_add_epi32(p32, p10);
}

/* the same integers with the reference's instruction mix: 16x-scaled nibble bytes through vpmaddubsw */
static inline __m256i block_word_isums_maddubs(__m256i u, __m256i v)
{
    const __m256i hi_mask = _mm256_set1_epi8((char)0xF0);
    const __m256i u_hi = _mm256_and_si256(u, hi_mask), v_hi = _mm256_and_si256(v, hi_mask);                   /* 16 * q(2i)   */
    const __m256i u_lo = _mm256_and_si256(_mm256_slli_epi16(u, 4), hi_mask);                                  /* 16 * q(2i+1) */
    const __m256i v_lo = _mm256_and_si256(_mm256_slli_epi16(v, 4), hi_mask);
    /* |16 q| <= 112 fits the unsigned operand; the sign moves to the signed one.  2 * 112^2 < 32767: no saturation */
    const __m256i p_hi = _mm256_maddubs_epi16(_mm256_abs_epi8(u_hi), _mm256_sign_epi8(v_hi, u_hi));
    const __m256i p_lo = _mm256_maddubs_epi16(_mm256_abs_epi8(u_lo), _mm256_sign_epi8(v_lo, u_lo));
    const __m256i s16 = _mm256_add_epi16(_mm256_srai_epi16(p_hi, 8), _mm256_srai_epi16(p_lo, 8));            /* 4 products each */
    return _mm256_madd_epi16(s16, _mm256_set1_epi16(1));                                                      /* 8 per 32-bit word */
}

static int g_kernel = 1;
void orcf_set_kernel(int k) { g_kernel = k ? 1 : 0; }
int orcf_get_kernel(void) { return g_kernel; }

/* the reference's final tree (CloverBase.h:149-157) on acc0 + acc1 */
static inline float reduce_tree(__m256 acc0, __m256 acc1)
{
    float v[8];
    _mm256_storeu_ps(v, _mm256_add_ps(acc0, acc1));
    const float x0 = v[4] + v[0], x1 = v[5] + v[1], x2 = v[6] + v[2], x3 = v[7] + v[3];
    const float y0 = x0 + x2, y1 = x1 + x3;
    return y0 + y1;
}

#define DOT_ROW_BODY(ISUMS)                                                                     \
    __m256 acc0 = _mm256_setzero_ps(), acc1 = _mm256_setzero_ps();                             \
    for (uint64_t b = 0; b < nb; b += 2) {                                                     \
        const __m256i ua = _mm256_loadu_si256((const __m256i *)(qu + 32 * b));                 \
        const __m256i va = _mm256_loadu_si256((const __m256i *)(qv + 32 * b));                 \
        const __m256i ub = _mm256_loadu_si256((const __m256i *)(qu + 32 * b + 32));            \
        const __m256i vb = _mm256_loadu_si256((const __m256i *)(qv + 32 * b + 32));            \
        const float ca = (su[b] * RCP49) * sv[b];                                              \
        const float cb = (su[b + 1] * RCP49) * sv[b + 1];                                      \
        const __m256 fa = _mm256_cvtepi32_ps(ISUMS(ua, va));                                   \
        const __m256 fb = _mm256_cvtepi32_ps(ISUMS(ub, vb));                                   \
        acc0 = _mm256_fmadd_ps(_mm256_set1_ps(ca), fa, acc0);                                  \
        acc1 = _mm256_fmadd_ps(_mm256_set1_ps(cb), fb, acc1);                                  \
    }                                                                                          \
    return reduce_tree(acc0, acc1);

static float dot_row_planes(const uint8_t *qu, const float *su, const uint8_t *qv, const float *sv, uint64_t nb) { DOT_ROW_BODY(block_word_isums) }
static float dot_row_maddubs(const uint8_t *qu, const float *su, const uint8_t *qv, const float *sv, uint64_t nb) { DOT_ROW_BODY(block_word_isums_maddubs) }

static inline float dot_row(const uint8_t *qu, const float *su, const uint8_t *qv, const float *sv, uint64_t nb)
{
    return g_kernel ? dot_row_maddubs(qu, su, qv, sv, nb) : dot_row_planes(qu, su, qv, sv, nb);
}

float orcf_v4_dot(const uint8_t *qu, const float *su, const uint8_t *qv, const float *sv, uint64_t n_pad)
{
    return dot_row(qu, su, qv, sv, n_pad / 64);
}

/* quantise 64 floats with multiplier k (rounding disabled) into 32 bytes */
static inline void quant64(const float *x, float k, uint8_t *dst)
{
    const __m256 vk = _mm256_set1_ps(k);
    const __m256 absmask = _mm256_castsi256_ps(_mm256_set1_epi32(0x7FFFFFFF));
    /* element e of a word goes to bit 8*(e/2) + (e even ? 4 : 0) */
    const __m256i sh = _mm256_setr_epi32(4, 0, 12, 8, 20, 16, 28, 24);
    uint32_t words[8];
    for (int g = 0; g < 8; g++) {
        const __m256 xv = _mm256_loadu_ps(x + 8 * g);
        const __m256 p = _mm256_fmadd_ps(_mm256_and_ps(xv, absmask), vk, _mm256_setzero_ps());
        const __m256i t = _mm256_cvttps_epi32(p);
        const __m256i q = _mm256_sign_epi32(t, _mm256_castps_si256(xv));
        __m256i n = _mm256_sllv_epi32(_mm256_and_si256(q, _mm256_set1_epi32(0xF)), sh);
        /* OR-reduce the 8 lanes into one dword */
        __m128i r = _mm_or_si128(_mm256_castsi256_si128(n), _mm256_extracti128_si256(n, 1));
        r = _mm_or_si128(r, _mm_shuffle_epi32(r, 0x4E));
        r = _mm_or_si128(r, _mm_shuffle_epi32(r, 0xB1));
        words[g] = (uint32_t)_mm_cvtsi128_si32(r);
    }
    memcpy(dst, words, 32);
}

static inline float absmax64(const float *x)
{
    const __m256 absmask = _mm256_castsi256_ps(_mm256_set1_epi32(0x7FFFFFFF));
    __m256 m = _mm256_setzero_ps();
    for (int g = 0; g < 8; g++) m = _mm256_max_ps(m, _mm256_and_ps(_mm256_loadu_ps(x + 8 * g), absmask));
    float v[8];
    _mm256_storeu_ps(v, m);
    float r = 0.0f;
    for (int i = 0; i < 8; i++) if (v[i] > r) r = v[i];
    return r;
}

static inline float fix_zero(float m)
{
    uint32_t u;
    memcpy(&u, &m, 4);
    return u == 0 ? m + 1.0f : m;
}

void orcf_v4_quantize(const float *x, uint64_t n_pad, uint8_t *q, float *s)
{
    const int64_t nb = (int64_t)(n_pad / 64);
#pragma omp parallel for schedule(static)
    for (int64_t b = 0; b < nb; b++) {
        const float m = fix_zero(absmax64(x + 64 * b));
        s[b] = m;
        quant64(x + 64 * b, 7.0f / m, q + 32 * b);
    }
}

/* copy a matrix into `dst` with the SAME static partition orcf_m4_mvm uses, so that with bound threads every thread's row
 * groups land in its own NUMA node's memory (first touch) -- what a tuned CPU run of the reference would arrange */
void orcf_first_touch_copy(uint8_t *dst, const uint8_t *src, uint64_t rows, uint64_t row_bytes)
{
    const int64_t groups = (int64_t)(rows / 64);
#pragma omp parallel for schedule(static)
    for (int64_t g = 0; g < groups; g++) memcpy(dst + (uint64_t)g * 64 * row_bytes, src + (uint64_t)g * 64 * row_bytes, 64 * row_bytes);
}

void orcf_m4_mvm(const uint8_t *A, const float *sA, uint64_t rows, uint64_t cols,
                 const uint8_t *x, const float *sx, uint8_t *r, float *sr)
{
    const uint64_t hb = cols / 64;
    const int64_t groups = (int64_t)(rows / 64);
#pragma omp parallel for schedule(static)
    for (int64_t g = 0; g < groups; g++) {
        float d[64];
        for (int k = 0; k < 64; k++)
            d[k] = dot_row(A + ((uint64_t)g * 64 + k) * (cols / 2), sA + (uint64_t)g * hb, x, sx, hb);
        const float m = fix_zero(absmax64(d));
        sr[g] = m;
        quant64(d, 7.0f / m, r + 32 * g);
    }
}

/* CloverVector4::dot_parallel (CloverVector4.h:1793-1907): block PAIRS split contiguously over the threads
 * (iter_per_thread = ceil(pairs / nt), :1807-1810), every thread runs dot's two 8-lane fma chains over its own range and folds them with
 * the same tree, and the per-thread partials meet in an OpenMP `reduction(+:sum)` whose order is unspecified -- so this result is
 * tolerance-only, like the reference's (never a parity target; timed as "dot, all cores"). */
float orcf_v4_dot_parallel(const uint8_t *qu, const float *su, const uint8_t *qv, const float *sv, uint64_t n_pad)
{
    const uint64_t pairs = n_pad / 128;
    float sum = 0.0f;
#pragma omp parallel reduction(+ : sum)
    {
#ifdef _OPENMP
        const uint64_t nt = (uint64_t)omp_get_num_threads(), tid = (uint64_t)omp_get_thread_num();
#else
        const uint64_t nt = 1, tid = 0;
#endif
        const uint64_t per = (pairs + nt - 1) / nt;
        const uint64_t start = per * tid, end = start + per < pairs ? start + per : pairs;
        if (start < end) sum = dot_row(qu + 64 * start, su + 2 * start, qv + 64 * start, sv + 2 * start, 2 * (end - start));
    }
    return sum;
}

/* ---- GEMM (build-defined, clover4_oracle.c orc_m4_gemm): C = A * B^T, A is M x K, B is N x K, one fma chain over the K-blocks per element:
 *     S_b = sum of the block's 64 nibble products (exact),  c_b = f32(f32(sA * 1/49) * sB),  C = fmaf(c_b, (float)S_b, C)  for b = 0, 1, ...
 * AVX2: eight columns j at a time -- their eight per-word sum vectors (the maddubs pipeline above) are folded to one vector of eight block
 * sums with a vphaddd tree, and the eight chains advance with one vfmadd.  Rows of C are split over the threads.  Same bits as the scalar
 * orc_m4_gemm (tests/test_oracle_properties.py); exists so that WHOLE results of the timed GEMM sizes can be compared. */
static inline __m256i hsum8x8(const __m256i v[8])
{
    /* vphaddd works inside 128-bit halves: after three levels lane k of each half holds the half-sum of v[k]'s (k < 4: from the first level's
     * ordering), then the halves are added crosswise */
    const __m256i a01 = _mm256_hadd_epi32(v[0], v[1]), a23 = _mm256_hadd_epi32(v[2], v[3]);
    const __m256i a45 = _mm256_hadd_epi32(v[4], v[5]), a67 = _mm256_hadd_epi32(v[6], v[7]);
    const __m256i b0123 = _mm256_hadd_epi32(a01, a23), b4567 = _mm256_hadd_epi32(a45, a67);
    /* b0123 = [s0 s1 s2 s3 | s0' s1' s2' s3'] (low / high half partial sums), b4567 likewise for 4..7 */
    const __m256i lo = _mm256_permute2x128_si256(b0123, b4567, 0x20);      /* low halves:  [s0..s3 | s4..s7]   */
    const __m256i hi = _mm256_permute2x128_si256(b0123, b4567, 0x31);      /* high halves: [s0'..s3' | s4'..s7'] */
    return _mm256_add_epi32(lo, hi);
}

void orcf_m4_gemm(const uint8_t *A, const float *sA, uint64_t M, uint64_t K, const uint8_t *B, const float *sB, uint64_t N, float *C)
{
    const uint64_t kb = K / 64;
    const int64_t rows = (int64_t)M;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < rows; i++) {
        const uint8_t *a = A + (uint64_t)i * (K / 2);
        const float *sa = sA + ((uint64_t)i >> 6) * kb;
        for (uint64_t j = 0; j < N; j += 8) {
            const float *sb = sB + (j >> 6) * kb;
            __m256 acc = _mm256_setzero_ps();
            for (uint64_t b = 0; b < kb; b++) {
                const __m256i u = _mm256_loadu_si256((const __m256i *)(a + 32 * b));
                __m256i w[8];
                for (int t = 0; t < 8; t++) {
                    const __m256i v = _mm256_loadu_si256((const __m256i *)(B + (j + (uint64_t)t) * (K / 2) + 32 * b));
                    w[t] = g_kernel ? block_word_isums_maddubs(u, v) : block_word_isums(u, v);
                }
                const float c = (sa[b] * RCP49) * sb[b];
                acc = _mm256_fmadd_ps(_mm256_set1_ps(c), _mm256_cvtepi32_ps(hsum8x8(w)), acc);
            }
            _mm256_storeu_ps(C + (uint64_t)i * N + j, acc);
        }
    }
}
