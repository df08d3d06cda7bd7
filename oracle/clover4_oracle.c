/*
 * clover4_oracle.c -- scalar CPU restatement of Clover's 4-bit hot path.  TEST INFRASTRUCTURE ONLY.
 * See clover4_oracle.h for scope, pinning status and the reference file:line each function follows.
 *
 * Build: gcc -O2 -std=c11 -ffp-contract=off -mfma (see oracle/Makefile).  -ffp-contract=off matters:
 * the reference's arithmetic is a fixed sequence of separately rounded fp32 operations plus explicit
 * fused multiply-adds; the compiler must not fuse or split anything.
 */
#include "clover4_oracle.h"

#include <math.h>
#include <string.h>

/* 1.0f/49.0f, correctly rounded = 0x3CA72F05 (CloverBase.h:88). */
static const float RCP49 = 1.0f / 49.0f;

static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float    u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

/* signed value of the high / low nibble of a byte */
static inline int nib_hi(uint8_t b) { return (int)(int8_t)b >> 4; }
static inline int nib_lo(uint8_t b) { return (int)(int8_t)(uint8_t)(b << 4) >> 4; }

/* ------------------------------------------------------------------------------------------------ */
/* XORShift128+ as the reference runs it                                                              */
/* ------------------------------------------------------------------------------------------------ */

/* canonical xorshift128+ state step, used only while deriving lanes 1..3 (simdxorshift128plus.h:38-44) */
static void canon_step(uint64_t *a, uint64_t *b)
{
    uint64_t s1 = *a;
    const uint64_t s0 = *b;
    *a = s0;
    s1 ^= s1 << 23;
    *b = s1 ^ s0 ^ (s1 >> 18) ^ (s0 >> 5);
}

/* 2^64-step jump polynomial (simdxorshift128plus.h:47-62) */
static void canon_jump(uint64_t in0, uint64_t in1, uint64_t *out0, uint64_t *out1)
{
    static const uint64_t poly[2] = { 0x8a5cd789635d2dffULL, 0x121fd2155c472f96ULL };
    uint64_t a = 0, b = 0;
    for (int i = 0; i < 2; i++) {
        for (int bit = 0; bit < 64; bit++) {
            if (poly[i] & (1ULL << bit)) { a ^= in0; b ^= in1; }
            canon_step(&in0, &in1);
        }
    }
    *out0 = a; *out1 = b;
}

void orc_rng_init(orc_rng *r, uint64_t key1, uint64_t key2)
{
    r->s0[0] = key1; r->s1[0] = key2;
    for (int l = 1; l < 4; l++) canon_jump(r->s0[l - 1], r->s1[l - 1], &r->s0[l], &r->s1[l]);
}

/*
 * One draw.  As written in the reference (simdxorshift128plus.h:97-109) `part1 = part2` runs first and
 * both temporaries derive from part2, so the effective per-lane state is the 64-bit s1 alone.
 */
void orc_rng_draw(orc_rng *r, uint32_t W[8])
{
    for (int l = 0; l < 4; l++) {
        const uint64_t a = r->s1[l];
        const uint64_t t = a ^ (a << 23);
        const uint64_t n = t ^ a ^ (t >> 18) ^ (a >> 5);
        const uint64_t out = n + a;
        r->s0[l] = a;
        r->s1[l] = n;
        W[2 * l]     = (uint32_t)out;
        W[2 * l + 1] = (uint32_t)(out >> 32);
    }
}

/* avx_xorshift128plus_jump (simdxorshift128plus.h:115-127): new lane 0 = 2^64-step jump of OLD lane 3,
 * lanes 1..3 chained from it -- a fresh, non-overlapping key for another thread. */
void orc_rng_jump(orc_rng *r)
{
    canon_jump(r->s0[3], r->s1[3], &r->s0[0], &r->s1[0]);
    for (int l = 1; l < 4; l++) canon_jump(r->s0[l - 1], r->s1[l - 1], &r->s0[l], &r->s1[l]);
}

/* `count` draws folded into two words (xor and wrapping sum of the four 64-bit lane outputs): long-stream checks */
void orc_rng_digest(orc_rng *r, uint64_t count, uint64_t *xor_fold, uint64_t *sum)
{
    uint64_t x = 0, a = 0;
    for (uint64_t i = 0; i < count; i++) {
        uint32_t W[8];
        orc_rng_draw(r, W);
        for (int l = 0; l < 4; l++) {
            const uint64_t o = (uint64_t)W[2 * l] | ((uint64_t)W[2 * l + 1] << 32);
            x ^= o; a += o;
        }
    }
    *xor_fold = x; *sum = a;
}

/* CloverVector4.h:690-734: mask 0x7F7F7F7F, four byte-shifts per draw, int->float, times 2^-31. */
void orc_rng_block_noise(orc_rng *r, float noise[8][8])
{
    const float rcp_2pow31 = 1.0f / 2147483648.0f;
    for (int d = 0; d < 2; d++) {
        uint32_t W[8];
        orc_rng_draw(r, W);
        for (int sh = 0; sh < 4; sh++)
            for (int j = 0; j < 8; j++) {
                const uint32_t v = (W[j] & 0x7F7F7F7Fu) << (8 * sh);
                noise[4 * d + sh][j] = (float)(int32_t)v * rcp_2pow31;
            }
    }
}

/* ------------------------------------------------------------------------------------------------ */
/* quantisation core                                                                                 */
/* ------------------------------------------------------------------------------------------------ */

/* 0 -> 1.0 fix-up on the bit pattern (CloverVector4.h:661-663) */
static inline float fix_zero_max(float m) { return f2u(m) == 0u ? m + 1.0f : m; }

/* one element: trunc(fma(|x|, k, noise)) with x's sign re-applied the way _mm256_sign_epi32 does
 * (CloverVector4.h:741-772): pattern > 0 keeps, pattern == 0 zeroes, sign bit negates. */
static inline int quant1(float x, float k, float noise)
{
    const float ax = u2f(f2u(x) & 0x7FFFFFFFu);
    const float p = fmaf(ax, k, noise);
    /* cvttps semantics; out-of-range (never reached for finite in-contract data) gives INT_MIN */
    int32_t t = (p >= 2147483648.0f || p < -2147483648.0f || p != p) ? INT32_MIN : (int32_t)p;
    const int32_t xb = (int32_t)f2u(x);
    if (xb == 0) return 0;
    return xb < 0 ? (int)(0u - (uint32_t)t) : t;
}

static inline uint8_t pack2(int q_even, int q_odd)
{
    return (uint8_t)(((q_even & 0xF) << 4) | (q_odd & 0xF));
}

/* quantise 64 consecutive values with a given scale multiplier; noise_of(e) supplies the noise */
static void quant_block64(const float *x, float k, const float *noise64 /* per element or NULL */, uint8_t *q)
{
    for (int i = 0; i < 64; i += 2) {
        const int a = quant1(x[i],     k, noise64 ? noise64[i]     : 0.0f);
        const int b = quant1(x[i + 1], k, noise64 ? noise64[i + 1] : 0.0f);
        q[i >> 1] = pack2(a, b);
    }
}

void orc_v4_quantize(const float *x, uint64_t n_pad, uint8_t *q, float *s, orc_rng *rng)
{
    const uint64_t nb = n_pad / 64;
    for (uint64_t b = 0; b < nb; b++) {
        const float *xb = x + 64 * b;
        float m = 0.0f;
        for (int i = 0; i < 64; i++) { const float a = fabsf(xb[i]); if (a > m) m = a; }
        m = fix_zero_max(m);
        s[b] = m;
        const float k = 7.0f / m;
        if (rng) {
            /* noise group g, lane j is added to element 8g + j (before the register transpose) */
            float nz[8][8], flat[64];
            orc_rng_block_noise(rng, nz);
            for (int g = 0; g < 8; g++) for (int j = 0; j < 8; j++) flat[8 * g + j] = nz[g][j];
            quant_block64(xb, k, flat, q + 32 * b);
        } else {
            quant_block64(xb, k, 0, q + 32 * b);
        }
    }
}

void orc_v4_restore(const uint8_t *q, const float *s, uint64_t n_pad, float *x)
{
    const uint64_t nb = n_pad / 64;
    for (uint64_t b = 0; b < nb; b++) {
        const float sc = s[b] / 7.0f;                         /* division first (CloverVector4.h:1050) */
        for (int i = 0; i < 32; i++) {
            const uint8_t v = q[32 * b + i];
            x[64 * b + 2 * i]     = (float)nib_hi(v) * sc;
            x[64 * b + 2 * i + 1] = (float)nib_lo(v) * sc;
        }
    }
}

float orc_v4_get(const uint8_t *q, const float *s, uint64_t pos)
{
    const float sc = s[pos >> 6] / 7.0f;
    const uint8_t v = q[pos >> 1];
    return sc * (float)((pos & 1) ? nib_lo(v) : nib_hi(v));
}

/* ------------------------------------------------------------------------------------------------ */
/* dot                                                                                               */
/* ------------------------------------------------------------------------------------------------ */

/* exact integer sum of the 8 nibble products of one little-endian 32-bit word (4 bytes) */
static inline int32_t word_isum(const uint8_t *u, const uint8_t *v)
{
    int32_t acc = 0;
    for (int i = 0; i < 4; i++)
        acc += nib_hi(u[i]) * nib_hi(v[i]) + nib_lo(u[i]) * nib_lo(v[i]);
    return acc;
}

void orc_v4_word_isums(const uint8_t *qu, const uint8_t *qv, uint64_t n_pad, int32_t *I)
{
    const uint64_t nw = n_pad / 8;
    for (uint64_t w = 0; w < nw; w++) I[w] = word_isum(qu + 4 * w, qv + 4 * w);
}

/* the final reduction of the 2 x 8 lane accumulators (CloverVector4.h:1190-1191, CloverBase.h:149-157) */
static inline float reduce16(const float acc[2][8])
{
    float v[8], x[4];
    for (int w = 0; w < 8; w++) v[w] = acc[0][w] + acc[1][w];
    for (int j = 0; j < 4; j++) x[j] = v[j + 4] + v[j];
    const float y0 = x[0] + x[2];
    const float y1 = x[1] + x[3];
    return y0 + y1;
}

/* one row/vector dot in the reference's SIMD order; su may be a tile-scale row of a matrix */
static float dot_simd_order(const uint8_t *qu, const float *su, const uint8_t *qv, const float *sv, uint64_t nb)
{
    float acc[2][8];
    memset(acc, 0, sizeof acc);
    for (uint64_t b = 0; b < nb; b++) {
        const float c = (su[b] * RCP49) * sv[b];              /* two separately rounded multiplies */
        float *a = acc[b & 1];
        for (int w = 0; w < 8; w++) {
            const int32_t I = word_isum(qu + 32 * b + 4 * w, qv + 32 * b + 4 * w);
            a[w] = fmaf(c, (float)I, a[w]);
        }
    }
    return reduce16(acc);
}

float orc_v4_dot(const uint8_t *qu, const float *su, const uint8_t *qv, const float *sv, uint64_t n_pad)
{
    return dot_simd_order(qu, su, qv, sv, n_pad / 64);
}

float orc_v4_dot_scalar(const uint8_t *qu, const float *su, const uint8_t *qv, const float *sv, uint64_t n_pad)
{
    const uint64_t nb = n_pad / 64;
    float result = 0.0f;
    for (uint64_t b = 0; b < nb; b++) {
        int16_t acc = 0;
        for (int i = 0; i < 32; i++) {
            const uint8_t u = qu[32 * b + i], v = qv[32 * b + i];
            acc = (int16_t)(acc + (int16_t)(nib_hi(u) * nib_hi(v)) + (int16_t)(nib_lo(u) * nib_lo(v)));
        }
        const float sc = (su[b] / 7.0f) * (sv[b] / 7.0f);
        const float term = sc * (float)acc;                   /* separate multiply, then add */
        result = result + term;
    }
    return result;
}

double orc_v4_dot_f64(const uint8_t *qu, const float *su, const uint8_t *qv, const float *sv, uint64_t n_pad)
{
    const uint64_t nb = n_pad / 64;
    double r = 0.0;
    for (uint64_t b = 0; b < nb; b++) {
        int32_t I = 0;
        for (int w = 0; w < 8; w++) I += word_isum(qu + 32 * b + 4 * w, qv + 32 * b + 4 * w);
        r += (double)su[b] * (double)sv[b] * (double)I / 49.0;
    }
    return r;
}

/* ------------------------------------------------------------------------------------------------ */
/* matrix                                                                                            */
/* ------------------------------------------------------------------------------------------------ */

void orc_m4_quantize(const float *A, uint64_t rows, uint64_t cols, uint8_t *q, float *s, orc_rng *rng)
{
    const uint64_t hb = cols >> 6, vb = rows >> 6;
    for (uint64_t bj = 0; bj < hb; bj++) {                    /* column-block outer (CloverMatrix4.h:524) */
        for (uint64_t bi = 0; bi < vb; bi++) {
            const float *t = A + (bi << 6) * cols + (bj << 6);
            float m = 0.0f;
            for (int i = 0; i < 64; i++)
                for (int j = 0; j < 64; j++) { const float a = fabsf(t[(uint64_t)i * cols + j]); if (a > m) m = a; }
            m = fix_zero_max(m);
            s[bi * hb + bj] = m;
            const float k = 7.0f / m;
            for (int i = 0; i < 64; i++) {
                uint8_t *dst = q + (((bi << 6) + i) * cols + (bj << 6)) / 2;
                if (rng) {
                    float nz[8][8], flat[64];
                    orc_rng_block_noise(rng, nz);             /* two draws per tile row (:650-700) */
                    for (int g = 0; g < 8; g++) for (int j = 0; j < 8; j++) flat[8 * g + j] = nz[g][j];
                    quant_block64(t + (uint64_t)i * cols, k, flat, dst);
                } else {
                    quant_block64(t + (uint64_t)i * cols, k, 0, dst);
                }
            }
        }
    }
}

void orc_m4_restore(const uint8_t *q, const float *s, uint64_t rows, uint64_t cols, float *A)
{
    const uint64_t hb = cols >> 6;
    for (uint64_t i = 0; i < rows; i++)
        for (uint64_t b = 0; b < hb; b++) {
            const float sc = s[(i >> 6) * hb + b] / 7.0f;                 /* :284 */
            for (uint64_t k = 0; k < 32; k++) {
                const uint8_t v = q[(i * cols + 64 * b) / 2 + k];
                A[i * cols + 64 * b + 2 * k]     = sc * (float)nib_hi(v);
                A[i * cols + 64 * b + 2 * k + 1] = sc * (float)nib_lo(v);
            }
        }
}

float orc_m4_get(const uint8_t *q, const float *s, uint64_t rows, uint64_t cols, uint64_t i, uint64_t j)
{
    (void)rows;
    const float sc = s[(i >> 6) * (cols >> 6) + (j >> 6)] / 7.0f;
    const uint64_t pos = i * cols + j;
    const uint8_t v = q[pos >> 1];
    return sc * (float)((pos & 1) ? nib_lo(v) : nib_hi(v));
}

void orc_m4_rowdots(const uint8_t *A, const float *sA, uint64_t rows, uint64_t cols,
                    const uint8_t *x, const float *sx, float *d)
{
    const uint64_t hb = cols >> 6;
    for (uint64_t r = 0; r < rows; r++)
        d[r] = dot_simd_order(A + r * (cols / 2), sA + (r >> 6) * hb, x, sx, hb);
}

void orc_m4_requantize64(const float d[64], uint8_t r[32], float *sr, orc_rng *rng)
{
    float m = 0.0f;
    for (int i = 0; i < 64; i++) { const float a = fabsf(d[i]); if (a > m) m = a; }
    m = fix_zero_max(m);
    *sr = m;
    const float k = 7.0f / m;
    if (rng) {
        /* results sit pre-transposed in block_values (CloverMatrix4.h:806-808, 909): noise group g,
         * lane j lands on output row 8j + g */
        float nz[8][8], flat[64];
        orc_rng_block_noise(rng, nz);
        for (int g = 0; g < 8; g++) for (int j = 0; j < 8; j++) flat[8 * j + g] = nz[g][j];
        quant_block64(d, k, flat, r);
    } else {
        quant_block64(d, k, 0, r);
    }
}

void orc_m4_mvm(const uint8_t *A, const float *sA, uint64_t rows, uint64_t cols,
                const uint8_t *x, const float *sx, uint8_t *r, float *sr, orc_rng *rng)
{
    const uint64_t hb = cols >> 6;
    for (uint64_t i = 0; i < rows; i += 64) {
        float d[64];
        for (int k = 0; k < 64; k++)
            d[k] = dot_simd_order(A + (i + k) * (cols / 2), sA + (i >> 6) * hb, x, sx, hb);
        orc_m4_requantize64(d, r + i / 2, sr + (i >> 6), rng);
    }
}

/* ------------------------------------------------------------------------------------------------ */
/* section 8(f): scaleAndAdd, transpose, threshold                                                    */
/* ------------------------------------------------------------------------------------------------ */

void orc_v4_scale_and_add(const uint8_t *qu, const float *su, const uint8_t *qv, const float *sv, float a,
                          uint64_t n_pad, uint8_t *r, float *sr, orc_rng *rng)
{
    const uint64_t nb = n_pad / 64;
    for (uint64_t b = 0; b < nb; b++) {
        const float su7 = su[b] / 7.0f;
        const float sva = sv[b] * a;                 /* rounded product first (CloverVector4.h:1226) */
        const float sv7 = sva / 7.0f;
        float val[64];
        for (int i = 0; i < 32; i++) {
            const uint8_t bu = qu[32 * b + i], bv = qv[32 * b + i];
            const float du_hi = (float)nib_hi(bu) * su7;
            const float du_lo = (float)nib_lo(bu) * su7;
            val[2 * i]     = fmaf((float)nib_hi(bv), sv7, du_hi);
            val[2 * i + 1] = fmaf((float)nib_lo(bv), sv7, du_lo);
        }
        float m = 0.0f;
        for (int i = 0; i < 64; i++) { const float x = fabsf(val[i]); if (x > m) m = x; }
        m = fix_zero_max(m);
        sr[b] = m;
        const float k = 7.0f / m;
        if (rng) {
            /* nibbles are unpacked by bit position (lowest nibble first), so noise group g of AVX lane j
             * meets the element at bit-nibble g of word j, which is element 8j + (g ^ 1) */
            float nz[8][8], flat[64];
            orc_rng_block_noise(rng, nz);
            for (int g = 0; g < 8; g++) for (int j = 0; j < 8; j++) flat[8 * j + (g ^ 1)] = nz[g][j];
            quant_block64(val, k, flat, r + 32 * b);
        } else {
            quant_block64(val, k, 0, r + 32 * b);
        }
    }
}

static inline int nib_at(const uint8_t *q, uint64_t pos)
{
    const uint8_t v = q[pos >> 1];
    return (pos & 1) ? nib_lo(v) : nib_hi(v);
}

static inline void nib_set(uint8_t *q, uint64_t pos, int val)
{
    uint8_t *p = q + (pos >> 1);
    if (pos & 1) *p = (uint8_t)((*p & 0xF0) | (val & 0xF));
    else         *p = (uint8_t)((*p & 0x0F) | ((val & 0xF) << 4));
}

void orc_m4_transpose(const uint8_t *q, const float *s, uint64_t rows, uint64_t cols, uint8_t *qt, float *st)
{
    for (uint64_t i = 0; i < rows; i++)
        for (uint64_t j = 0; j < cols; j++)
            nib_set(qt, j * rows + i, nib_at(q, i * cols + j));
    const uint64_t vb = rows >> 6, hb = cols >> 6;
    for (uint64_t bi = 0; bi < vb; bi++)
        for (uint64_t bj = 0; bj < hb; bj++) st[bj * vb + bi] = s[bi * hb + bj];
}

/* the reference's heap element and helpers (CloverBase.h:208-249) */
typedef struct { float value; int bits; uint64_t idx; } heap_item;

static void sift_min(heap_item *h, uint64_t pos, uint64_t k)      /* min_heapify */
{
    uint64_t smallest = pos;
    for (;;) {
        const uint64_t l = pos * 2 + 1, r = pos * 2 + 2;
        if (l < k && h[l].value < h[smallest].value) smallest = l;
        if (r < k && h[r].value < h[smallest].value) smallest = r;
        if (smallest == pos) break;
        const heap_item t = h[pos]; h[pos] = h[smallest]; h[smallest] = t;
        pos = smallest;
    }
}

/* gt_idx_t (CloverBase.h:216-218): (a.value > b.value) || isnan(a.value) -- the NaN clause only matters for non-finite block scales */
static int gt_item(const heap_item *a, const heap_item *b) { return (a->value > b->value) || isnan(a->value); }

/* std::make_heap(first, last, gt_idx_t): libstdc++'s bottom-up construction (min-heap) */
static void push_down_gt(heap_item *h, uint64_t hole, uint64_t len, heap_item v)
{
    /* __adjust_heap: move the hole down to a leaf choosing the child that is NOT "less" under comp,
     * then __push_heap back up */
    const uint64_t top = hole;
    uint64_t child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (gt_item(&h[child], &h[child - 1])) child--;               /* comp(first+child, first+child-1) */
        h[hole] = h[child];
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        h[hole] = h[child - 1];
        hole = child - 1;
    }
    uint64_t parent = (hole - 1) / 2;
    while (hole > top && gt_item(&h[parent], &v)) {                   /* comp(first+parent, value) */
        h[hole] = h[parent];
        hole = parent;
        parent = (hole - 1) / 2;
    }
    h[hole] = v;
}

static void make_heap_gt(heap_item *h, uint64_t len)
{
    if (len < 2) return;
    uint64_t parent = (len - 2) / 2;
    for (;;) {
        const heap_item v = h[parent];
        push_down_gt(h, parent, len, v);
        if (parent == 0) return;
        parent--;
    }
}

#include <stdlib.h>

void orc_v4_threshold(uint8_t *q, const float *s, uint64_t n, uint64_t k)
{
    if (k == 0) { for (uint64_t i = 0; i < n; i++) nib_set(q, i, 0); return; }
    if (k >= n) return;
    heap_item *h = (heap_item *)malloc(k * sizeof(heap_item));
    for (uint64_t i = 0; i < k; i++) {
        h[i].value = fabsf(orc_v4_get(q, s, i));
        h[i].bits = nib_at(q, i);
        h[i].idx = i;
        nib_set(q, i, 0);
    }
    make_heap_gt(h, k);
    for (uint64_t i = k; i < n; i++) {
        const float v = fabsf(orc_v4_get(q, s, i));
        if (v > h[0].value) {
            h[0].value = v; h[0].idx = i; h[0].bits = nib_at(q, i);
            sift_min(h, 0, k);
        }
        nib_set(q, i, 0);
    }
    for (uint64_t i = 0; i < k; i++) nib_set(q, h[i].idx, h[i].bits);
    free(h);
}

void orc_m4_mvm_f32(const uint8_t *A, const float *sA, uint64_t rows, uint64_t cols, const float *x, float *r)
{
    const uint64_t hb = cols >> 6;
    for (uint64_t i = 0; i < rows; i++) {
        const uint8_t *u = A + i * (cols / 2);
        const float *su = sA + (i >> 6) * hb;
        float acc[4][8];
        memset(acc, 0, sizeof acc);
        for (uint64_t b = 0; b < hb; b++) {
            const float sc = su[b] / 7.0f;
            for (int g = 0; g < 8; g++)                      /* acc_{g mod 4} gets group g; g and g+4 in this order */
                for (int j = 0; j < 8; j++) {
                    const uint64_t e = 64 * b + 8 * g + j;
                    const float f = (float)nib_at(u, e) * sc;
                    acc[g & 3][j] = fmaf(x[e], f, acc[g & 3][j]);
                }
        }
        float s3[8], t[4];
        for (int j = 0; j < 8; j++) {
            const float s1 = acc[0][j] + acc[1][j];
            const float s2 = acc[2][j] + acc[3][j];
            s3[j] = s1 + s2;
        }
        for (int j = 0; j < 4; j++) t[j] = s3[j + 4] + s3[j];
        const float y0 = t[0] + t[2], y1 = t[1] + t[3];
        r[i] = y0 + y1;
    }
}

/* ------------------------------------------------------------------------------------------------ */
/* GEMM (build-defined; see header)                                                                  */
/* ================================================================================================
 * mixed precision: CloverMatrix4 x CloverVector8
 * ================================================================================================ */
static void quant8_block64(const float *x, float k, const float *noise64, int8_t *q)
{
    for (int i = 0; i < 64; i++) q[i] = (int8_t)quant1(x[i], k, noise64 ? noise64[i] : 0.0f);     /* low byte, as the packing keeps it */
}

void orc_v8_quantize(const float *x, uint64_t n_pad, int8_t *q, float *s, orc_rng *rng)
{
    const uint64_t nb = n_pad / 64;
    for (uint64_t b = 0; b < nb; b++) {
        const float *xb = x + 64 * b;
        float m = 0.0f;
        for (int i = 0; i < 64; i++) { const float a = fabsf(xb[i]); if (a > m) m = a; }
        m = fix_zero_max(m);
        s[b] = m;
        const float k = 127.0f / m;                              /* CloverVector8.h:455 */
        if (rng) {
            float nz[8][8], flat[64];
            orc_rng_block_noise(rng, nz);
            for (int g = 0; g < 8; g++) for (int j = 0; j < 8; j++) flat[8 * g + j] = nz[g][j];     /* rnd_(g+1) meets u_(g+1) (:546-553) */
            quant8_block64(xb, k, flat, q + 64 * b);
        } else {
            quant8_block64(xb, k, 0, q + 64 * b);
        }
    }
}

/* CloverVector8::dot (CloverVector8.h:911-977): per block the 64 byte products are summed per 32-bit lane of the two 32-byte halves
 * (sign / maddubs / madd / add_epi32: exact, |q| <= 127 so maddubs never saturates), scale = f32(f32(su * 1/127) * f32(sv * 1/127)), ONE
 * accumulator register: acc[w] = fma(scale, (float)I[w], acc[w]) for every block in order -- 8 sequential chains -- then the
 * _mm256_haddf32_ps tree (CloverBase.h:149-157). */
static const float RCP127 = 1.0f / 127.0f;
float orc_v8_dot(const int8_t *qu, const float *su, const int8_t *qv, const float *sv, uint64_t n_pad)
{
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (uint64_t b = 0; b < n_pad / 64; b++) {
        const float scale = (su[b] * RCP127) * (sv[b] * RCP127);
        for (int w = 0; w < 8; w++) {
            int32_t I = 0;
            for (int i = 0; i < 4; i++)
                I += (int32_t)qu[64 * b + 4 * w + i] * qv[64 * b + 4 * w + i] + (int32_t)qu[64 * b + 32 + 4 * w + i] * qv[64 * b + 32 + 4 * w + i];
            acc[w] = fmaf(scale, (float)I, acc[w]);
        }
    }
    float x[4];
    for (int j = 0; j < 4; j++) x[j] = acc[j + 4] + acc[j];
    const float y0 = x[0] + x[2];
    const float y1 = x[1] + x[3];
    return y0 + y1;
}
/* CloverVector8::dot_scalar (CloverVector8.h:268-310): the reference's tolerance partner */
float orc_v8_dot_scalar(const int8_t *qu, const float *su, const int8_t *qv, const float *sv, uint64_t n_pad)
{
    float result = 0.0f;
    for (uint64_t b = 0; b < n_pad / 64; b++) {
        const float scale = (su[b] / 127.0f) * (sv[b] / 127.0f);
        int32_t I = 0;
        for (int i = 0; i < 64; i++) I += (int32_t)qu[64 * b + i] * qv[64 * b + i];
        const float term = (float)I * scale;                  /* separate multiply, then add */
        result = result + term;
    }
    return result;
}
double orc_v8_dot_f64(const int8_t *qu, const float *su, const int8_t *qv, const float *sv, uint64_t n_pad)
{
    double r = 0.0;
    for (uint64_t b = 0; b < n_pad / 64; b++) {
        const float scale = (su[b] * RCP127) * (sv[b] * RCP127);
        int64_t I = 0;
        for (int i = 0; i < 64; i++) I += (int32_t)qu[64 * b + i] * qv[64 * b + i];
        r += (double)scale * (double)I;
    }
    return r;
}

void orc_v8_restore(const int8_t *q, const float *s, uint64_t n_pad, float *x)
{
    for (uint64_t i = 0; i < n_pad; i++) x[i] = (float)q[i] * (s[i >> 6] / 127.0f);
}

static float dot_v8_simd_order(const uint8_t *arow, const float *sa, const int8_t *x, const float *sx, uint64_t hb)
{
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (uint64_t b = 0; b < hb; b++) {
        const float su_scaled = sa[b] * (1.0f / 7.0f);           /* CloverMatrix4.h:1147-1149 */
        const float sv_scaled = sx[b] * (1.0f / 127.0f);
        const float c = su_scaled * sv_scaled;
        const uint8_t *ab = arow + 32 * b;
        const int8_t *xb = x + 64 * b;
        for (int L = 0; L < 8; L++) {
            int32_t I = 0;
            for (int e = 4 * L; e < 4 * L + 4; e++) {
                const int lo = (e & 1) ? nib_lo(ab[e >> 1]) : nib_hi(ab[e >> 1]);
                const int e2 = 32 + e;
                const int hi = (e2 & 1) ? nib_lo(ab[e2 >> 1]) : nib_hi(ab[e2 >> 1]);
                I += lo * (int)xb[e] + hi * (int)xb[e2];
            }
            acc[L] = fmaf(c, (float)I, acc[L]);                  /* :1224 */
        }
    }
    /* :1229-1234: (hi128 + lo128), then + movehl, then lanes 0 + 1 */
    const float h0 = acc[4] + acc[0], h1 = acc[5] + acc[1], h2 = acc[6] + acc[2], h3 = acc[7] + acc[3];
    return (h0 + h2) + (h1 + h3);
}

void orc_m4_rowdots_v8(const uint8_t *A, const float *sA, uint64_t rows, uint64_t cols, const int8_t *x, const float *sx, float *d)
{
    const uint64_t hb = cols >> 6;
    for (uint64_t i = 0; i < rows; i++) d[i] = dot_v8_simd_order(A + i * (cols / 2), sA + (i >> 6) * hb, x, sx, hb);
}

void orc_m4_mvm_v8(const uint8_t *A, const float *sA, uint64_t rows, uint64_t cols, const int8_t *x, const float *sx,
                   int8_t *r, float *sr, orc_rng *rng)
{
    const uint64_t hb = cols >> 6;
    for (uint64_t i = 0; i < rows; i += 64) {
        float d[64];
        for (int k = 0; k < 64; k++) d[k] = dot_v8_simd_order(A + (i + k) * (cols / 2), sA + (i >> 6) * hb, x, sx, hb);
        float m = 0.0f;
        for (int k = 0; k < 64; k++) { const float a = fabsf(d[k]); if (a > m) m = a; }
        m = fix_zero_max(m);
        sr[i >> 6] = m;
        const float kq = 127.0f / m;
        if (rng) {
            float nz[8][8], flat[64];
            orc_rng_block_noise(rng, nz);
            for (int g = 0; g < 8; g++) for (int j = 0; j < 8; j++) flat[8 * g + j] = nz[g][j];     /* rnd_(g+1) meets u_(g+1) = rows 8g.. (:1381-1388) */
            quant8_block64(d, kq, flat, r + i);
        } else {
            quant8_block64(d, kq, 0, r + i);
        }
    }
}

void orc_v8_scale_and_add(const int8_t *qu, const float *su, const int8_t *qv, const float *sv, float a, uint64_t n_pad,
                          int8_t *r, float *sr, orc_rng *rng)
{
    const uint64_t nb = n_pad / 64;
    for (uint64_t b = 0; b < nb; b++) {
        const float su_ps = su[b] / 127.0f;                     /* CloverVector8.h:1145-1146 */
        const float sva = sv[b] * a;                            /* :1095 */
        const float sv_ps = sva / 127.0f;
        float val[64];
        for (int i = 0; i < 64; i++) {
            const float du = (float)qu[64 * b + i] * su_ps;
            val[i] = fmaf((float)qv[64 * b + i], sv_ps, du);
        }
        float m = 0.0f;
        for (int i = 0; i < 64; i++) { const float x = fabsf(val[i]); if (x > m) m = x; }
        m = fix_zero_max(m);
        sr[b] = m;
        const float k = 127.0f / m;
        if (rng) {
            /* register k-1 = g holds byte (g&3) of the dwords of half g>>2: lane j of it is element 32 (g>>2) + 4 j + (g&3) (:1104-1126) */
            float nz[8][8], flat[64];
            orc_rng_block_noise(rng, nz);
            for (int g = 0; g < 8; g++) for (int j = 0; j < 8; j++) flat[32 * (g >> 2) + 4 * j + (g & 3)] = nz[g][j];
            quant8_block64(val, k, flat, r + 64 * b);
        } else {
            quant8_block64(val, k, 0, r + 64 * b);
        }
    }
}

static inline float v8_abs(const int8_t *q, const float *s, uint64_t i)
{
    return fabsf((float)q[i] * s[i >> 6] / 127.0f);            /* getAbs -> get (:137-140) */
}

void orc_v8_threshold(int8_t *q, const float *s, uint64_t n, uint64_t k)
{
    if (k == 0) { for (uint64_t i = 0; i < n; i++) q[i] = 0; return; }
    if (k >= n) return;
    heap_item *h = (heap_item *)malloc(k * sizeof(heap_item));
    for (uint64_t i = 0; i < k; i++) {
        h[i].value = v8_abs(q, s, i);
        h[i].bits = q[i];
        h[i].idx = i;
        q[i] = 0;
    }
    make_heap_gt(h, k);
    for (uint64_t i = k; i < n; i++) {
        const float v = v8_abs(q, s, i);
        if (v > h[0].value) {
            h[0].value = v; h[0].idx = i; h[0].bits = q[i];
            sift_min(h, 0, k);
        }
        q[i] = 0;
    }
    for (uint64_t i = 0; i < k; i++) q[h[i].idx] = (int8_t)h[i].bits;
    free(h);
}

void orc_m4_rowdots_v8_f64(const uint8_t *A, const float *sA, uint64_t rows, uint64_t cols, const int8_t *x, const float *sx, float *d)
{
    for (uint64_t i = 0; i < rows; i++) {
        double sum = 0;
        for (uint64_t j = 0; j < cols; j++) {
            const float xv = (float)x[j] * sx[j >> 6] / 127.0f;           /* CloverVector8::get (:137-140): q * scale / 127 */
            sum += (double)orc_m4_get(A, sA, rows, cols, i, j) * (double)xv;
        }
        d[i] = (float)sum;
    }
}

/* ------------------------------------------------------------------------------------------------ */

static inline int32_t block_isum(const uint8_t *u, const uint8_t *v)
{
    int32_t acc = 0;
    for (int w = 0; w < 8; w++) acc += word_isum(u + 4 * w, v + 4 * w);
    return acc;
}

void orc_m4_gemm_isums(const uint8_t *A, uint64_t M, uint64_t K, const uint8_t *B, uint64_t N, int32_t *S)
{
    const uint64_t kb = K >> 6;
    for (uint64_t i = 0; i < M; i++)
        for (uint64_t j = 0; j < N; j++)
            for (uint64_t b = 0; b < kb; b++)
                S[(i * N + j) * kb + b] = block_isum(A + i * (K / 2) + 32 * b, B + j * (K / 2) + 32 * b);
}

void orc_m4_gemm(const uint8_t *A, const float *sA, uint64_t M, uint64_t K,
                 const uint8_t *B, const float *sB, uint64_t N, float *C)
{
    const uint64_t kb = K >> 6;
    for (uint64_t i = 0; i < M; i++) {
        const float *sa = sA + (i >> 6) * kb;
        for (uint64_t j = 0; j < N; j++) {
            const float *sb = sB + (j >> 6) * kb;
            float acc = 0.0f;
            for (uint64_t b = 0; b < kb; b++) {
                const float c = (sa[b] * RCP49) * sb[b];
                const int32_t S = block_isum(A + i * (K / 2) + 32 * b, B + j * (K / 2) + 32 * b);
                acc = fmaf(c, (float)S, acc);
            }
            C[i * N + j] = acc;
        }
    }
}
