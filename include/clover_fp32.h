/*
 * clover_fp32.h -- the fp32 side of the reference's class family on the HOST: what CloverVector32 / CloverMatrix32 compute
 * (CloverVector32.h:160-684, CloverMatrix32.h:90-215), so that code written against the reference -- its validation tests compare
 * every 4-bit result with the 32-bit one, its experiments run Q_IHT / Q_GD on <CloverMatrix32, CloverVector32> as the baseline --
 * compiles and runs unchanged against these headers.
 *
 * This is NOT the hot path and nothing here runs on the GPU: the 32-bit classes are the callers' data format on either side of
 * the 4-bit path, their arithmetic is the comparison baseline.  Plain loops, written from the definitions:
 *
 *   dot          32 sequential fma chains (chain = element index mod 32), then (a1 + a2) + (a3 + a4) per lane and the tree of
 *                CloverBase.h:149-157 -- the order of the reference's AVX2 dot (CloverVector32.h:406-451), so the float is the same;
 *   dot_scalar   one chain, product rounded, then added (CloverVector32.h:191-205);
 *   scaleAndAdd  r = fma(v, s, u) per element (:291-323, its FMA branch); _scalar: u + v * s with both roundings (:181-189);
 *   threshold    the reference's walk (:549-600): std::make_heap over the first k under gt_idx_t, every later element against the root
 *                with a strict >, min_heapify with left-first ties (CloverBase.h:208-249) -- so the same elements survive;
 *   mvm          row dots in the order of `dot` above.  The reference hands this to MKL's sgemv (CloverMatrix32.h:90-128), whose
 *                summation order is not specified: there is nothing to be bit-equal to;
 *   transpose    out(j, i) = in(i, j) (:169-179).
 * Compile with -fopenmp and the row / element loops of the _parallel methods use the team; without it they are the same loops.
 */
#ifndef CLOVER_FP32_H
#define CLOVER_FP32_H

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace clover_fp32 {

/* a + b without the compiler fusing or re-associating anything around it */
inline float add(float a, float b) { volatile float r = a + b; return r; }

inline float dot_chains32(const float *u, const float *v, uint64_t n_pad)      /* n_pad: a multiple of 32 (it is one of 128) */
{
    float acc[32];
    for (int j = 0; j < 32; j++) acc[j] = 0.0f;
    for (uint64_t i = 0; i < n_pad; i += 32)
        for (int j = 0; j < 32; j++) acc[j] = std::fma(v[i + j], u[i + j], acc[j]);
    float lane[8];
    for (int j = 0; j < 8; j++) lane[j] = add(add(acc[j], acc[8 + j]), add(acc[16 + j], acc[24 + j]));
    const float t0 = add(lane[4], lane[0]), t1 = add(lane[5], lane[1]), t2 = add(lane[6], lane[2]), t3 = add(lane[7], lane[3]);
    return add(add(t0, t2), add(t1, t3));
}

inline float dot_sequential(const float *u, const float *v, uint64_t n)
{
    float r = 0.0f;
    for (uint64_t i = 0; i < n; i++) {
        volatile float p = u[i] * v[i];
        r = add(r, p);
    }
    return r;
}

inline void axpy_fma(const float *u, const float *v, float s, float *r, uint64_t n, bool team)
{
    (void)team;
#if defined(_OPENMP)
#pragma omp parallel for schedule(static) if (team)
#endif
    for (int64_t i = 0; i < (int64_t)n; i++) r[i] = std::fma(v[i], s, u[i]);
}

inline void axpy_two_roundings(const float *u, const float *v, float s, float *r, uint64_t n)
{
    for (uint64_t i = 0; i < n; i++) {
        volatile float p = v[i] * s;
        r[i] = add(u[i], p);
    }
}

struct HeapItem {
    float value;      /* |x| */
    float bits;       /* x itself */
    uint64_t idx;
};

/* keep the k largest magnitudes among values[0 .. n), zero the rest -- the reference's survivor set (see the header comment) */
inline void keep_top_k(float *values, uint64_t n, uint64_t k)
{
    if (k >= n) return;
    if (k == 0) { std::memset(values, 0, n * sizeof(float)); return; }
    std::vector<HeapItem> h(k);
    for (uint64_t i = 0; i < k; i++) {
        h[i] = HeapItem{std::fabs(values[i]), values[i], i};
        values[i] = 0.0f;
    }
    std::make_heap(h.begin(), h.end(), [](const HeapItem &a, const HeapItem &b) { return (a.value > b.value) || std::isnan(a.value); });
    for (uint64_t i = k; i < n; i++) {
        const float m = std::fabs(values[i]);
        if (m > h[0].value) {
            h[0] = HeapItem{m, values[i], i};
            uint64_t pos = 0;
            for (;;) {                                    /* min_heapify: the smaller child, the left one on a tie */
                const uint64_t l = 2 * pos + 1, r = 2 * pos + 2;
                uint64_t smallest = pos;
                if (l < k && h[l].value < h[smallest].value) smallest = l;
                if (r < k && h[r].value < h[smallest].value) smallest = r;
                if (smallest == pos) break;
                std::swap(h[pos], h[smallest]);
                pos = smallest;
            }
        }
        values[i] = 0.0f;
    }
    for (uint64_t i = 0; i < k; i++) values[h[i].idx] = h[i].bits;
}

inline void mvm_rows(const float *A, uint64_t rows, uint64_t cols, const float *x, float *y, bool team)
{
    (void)team;
#if defined(_OPENMP)
#pragma omp parallel for schedule(static) if (team)
#endif
    for (int64_t i = 0; i < (int64_t)rows; i++) y[i] = dot_chains32(A + (uint64_t)i * cols, x, cols);
}

inline void transpose(const float *in, uint64_t rows, uint64_t cols, float *out, bool team)
{
    (void)team;
    const uint64_t T = 32;                                 /* square blocks: both sides stay inside a few cache lines */
#if defined(_OPENMP)
#pragma omp parallel for schedule(static) if (team)
#endif
    for (int64_t bi = 0; bi < (int64_t)rows; bi += T)
        for (uint64_t bj = 0; bj < cols; bj += T)
            for (uint64_t i = (uint64_t)bi; i < (uint64_t)bi + T && i < rows; i++)
                for (uint64_t j = bj; j < bj + T && j < cols; j++) out[j * rows + i] = in[i * cols + j];
}

/* splitmix64: the stream behind setRandomInteger / setRandomFloats of these headers (test data; the reference draws from its XORShift keys) */
inline uint64_t splitmix_next(uint64_t &z)
{
    z += 0x9E3779B97F4A7C15ull;
    uint64_t r = z;
    r = (r ^ (r >> 30)) * 0xBF58476D1CE4E5B9ull;
    r = (r ^ (r >> 27)) * 0x94D049BB133111EBull;
    return r ^ (r >> 31);
}

/* uniform floats in [lo, hi): |31 random bits| * (hi - lo) / 2^31 + lo with one fma, as CloverVector32.h:751-770 forms them */
inline void fill_uniform(float *v, uint64_t n, float lo, float hi, uint64_t seed)
{
    const float step = (hi - lo) / 2147483648.0f;
    uint64_t z = seed;
    for (uint64_t i = 0; i < n; i++) v[i] = std::fma((float)(uint32_t)(splitmix_next(z) >> 33), step, lo);
}

}  // namespace clover_fp32

#endif
