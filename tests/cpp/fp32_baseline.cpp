// fp32_baseline.cpp -- the fp32 side of the class family (include/clover_fp32.h, CloverVector32 / CloverMatrix32 methods), host only:
//   * dot in the order of the reference's AVX2 dot (CloverVector32.h:406-451): compared bit for bit with an AVX2 + FMA evaluation written
//     here with intrinsics (4 accumulators, fmadd, (a1+a2)+(a3+a4), the horizontal tree of CloverBase.h:149-157);
//   * scaleAndAdd (2- and 3-argument, in place, _parallel, _scalar), quantize / restore as copies, mvm == row dots, transpose round trip;
//   * threshold: distinct magnitudes -> exactly the k largest survive; tie-heavy data -> the survivors are printed ("survivors ...") and the
//     Python side compares them with the oracle's heap walk on a 4-bit vector holding the same magnitudes;
//   * Q_IHT<CloverMatrix32, CloverVector32> (the generic template of CloverIHT.h, 01_measure.h:923-946) recovers a sparse vector: the fp32
//     baseline loop of the reference's experiments runs on these headers.
// g++ -std=c++11 -O2 -mavx2 -mfma -Iinclude tests/cpp/fp32_baseline.cpp fake_clv.o   (CPU only: tests/cpp/fake_clv.c stands in for the C ABI)
#include <immintrin.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "CloverIHT.h"
#include "CloverMatrix32.h"
#include "CloverVector32.h"

static int failures = 0;
#define CHECK(c)                                                               \
    do {                                                                       \
        if (!(c)) { std::printf("FAIL line %d: %s\n", __LINE__, #c); failures++; } \
    } while (0)

static bool same_bits(float a, float b) { return std::memcmp(&a, &b, 4) == 0; }

static float dot_avx2(const float *u, const float *v, uint64_t n_pad)
{
    __m256 a1 = _mm256_setzero_ps(), a2 = a1, a3 = a1, a4 = a1;
    for (uint64_t i = 0; i < n_pad; i += 32) {
        a1 = _mm256_fmadd_ps(_mm256_loadu_ps(v + i), _mm256_loadu_ps(u + i), a1);
        a2 = _mm256_fmadd_ps(_mm256_loadu_ps(v + i + 8), _mm256_loadu_ps(u + i + 8), a2);
        a3 = _mm256_fmadd_ps(_mm256_loadu_ps(v + i + 16), _mm256_loadu_ps(u + i + 16), a3);
        a4 = _mm256_fmadd_ps(_mm256_loadu_ps(v + i + 24), _mm256_loadu_ps(u + i + 24), a4);
    }
    const __m256 t = _mm256_add_ps(_mm256_add_ps(a1, a2), _mm256_add_ps(a3, a4));
    // horizontal sum: high half + low half, then [0]+[2], [1]+[3], then those two
    const __m128 x = _mm_add_ps(_mm256_extractf128_ps(t, 1), _mm256_castps256_ps128(t));
    float w[4];
    _mm_storeu_ps(w, x);
    volatile float y0 = w[0] + w[2], y1 = w[1] + w[3];
    volatile float r = y0 + y1;
    return r;
}

int main(int argc, char **argv)
{
    const uint64_t tie_n = argc > 1 ? std::strtoull(argv[1], nullptr, 10) : 0, tie_k = argc > 2 ? std::strtoull(argv[2], nullptr, 10) : 0;
    // ---- dot, three sizes incl. one that needs padding
    for (uint64_t n : {128ull, 1000ull, 40000ull}) {
        CloverVector32 a(n), b(n);
        a.setRandomFloats(-3.0f, 5.0f, 11 + n);
        b.setRandomFloats(-1.0f, 1.0f, 12 + n);
        CHECK(same_bits(a.dot(b), dot_avx2(a.getData(), b.getData(), a.size_pad())));
        CHECK(same_bits(a.dot_parallel(b), a.dot(b)));
        double ref = 0;
        for (uint64_t i = 0; i < n; i++) ref += (double)a.get(i) * b.get(i);
        CHECK(std::fabs(a.dot_scalar(b) - ref) <= 1e-4 * (1 + std::fabs(ref)) && std::fabs(a.dot(b) - ref) <= 1e-4 * (1 + std::fabs(ref)));
        for (uint64_t i = n; i < a.size_pad(); i++) CHECK(a.get(i) == 0.0f);      // padding stays zero
        for (uint64_t i = 0; i < n; i++) CHECK(a.get(i) >= -3.0f && a.get(i) < 5.0f);
    }
    // ---- scaleAndAdd, quantize / restore
    {
        const uint64_t n = 777;
        CloverVector32 u(n), v(n), r(n), keep(n);
        u.setRandomFloats(-2, 2, 1);
        v.setRandomFloats(-2, 2, 2);
        keep.quantize(u);
        for (uint64_t i = 0; i < n; i++) CHECK(same_bits(keep.get(i), u.get(i)));
        u.scaleAndAdd(v, 0.37f, r);
        for (uint64_t i = 0; i < n; i++) CHECK(same_bits(r.get(i), std::fma(v.get(i), 0.37f, u.get(i))));
        CloverVector32 r2(n);
        u.scaleAndAdd_parallel(v, 0.37f, r2);
        for (uint64_t i = 0; i < n; i++) CHECK(same_bits(r.get(i), r2.get(i)));
        u.scaleAndAdd(v, -1.5f);                                                  // in place
        for (uint64_t i = 0; i < n; i++) CHECK(same_bits(u.get(i), std::fma(v.get(i), -1.5f, keep.get(i))));
        keep.restore(u);
        u.scaleAndAdd_scalar(v, 0.1f);
        for (uint64_t i = 0; i < n; i++) {
            volatile float p = v.get(i) * 0.1f;
            volatile float s = keep.get(i) + p;
            CHECK(same_bits(u.get(i), s));
        }
        float *mine = (float *)std::malloc(keep.size_pad() * sizeof(float));
        for (uint64_t i = 0; i < keep.size_pad(); i++) mine[i] = (float)i;
        r.setData(mine);                                                          // a view from now on
        CHECK(r.getData() == mine && r.get(5) == 5.0f);
        r.set(5, -1.0f);
        CHECK(mine[5] == -1.0f);
        r.setData(nullptr);                                                       // the view lets go before the caller frees
        std::free(mine);
    }
    // ---- threshold: distinct magnitudes
    {
        const uint64_t n = 5000, k = 1234;
        CloverVector32 x(n), before(n);
        x.setRandomFloats(-1, 1, 99);
        before.quantize(x);
        std::vector<float> mags(n);
        for (uint64_t i = 0; i < n; i++) mags[i] = std::fabs(before.get(i));
        std::vector<float> sorted = mags;
        std::sort(sorted.begin(), sorted.end());
        const float tau = sorted[n - k];
        x.threshold(k);
        uint64_t kept = 0;
        for (uint64_t i = 0; i < n; i++) {
            if (mags[i] >= tau) { CHECK(same_bits(x.get(i), before.get(i))); kept++; }
            else CHECK(x.get(i) == 0.0f);
        }
        CHECK(kept == k);                                                          // (31-bit uniform draws: no ties at tau in this sample)
        CloverVector32 all(n), none(n);
        all.quantize(before); none.quantize(before);
        all.threshold(n);
        none.threshold_parallel(0);
        for (uint64_t i = 0; i < n; i++) CHECK(same_bits(all.get(i), before.get(i)) && none.get(i) == 0.0f);
    }
    // ---- threshold: ties everywhere (values in -7 .. 7), survivors handed to the Python side
    if (tie_n) {
        CloverVector32 x(tie_n);
        x.setRandomInteger(7, 4242);
        std::printf("values");
        for (uint64_t i = 0; i < tie_n; i++) std::printf(" %d", (int)x.get(i));
        std::printf("\n");
        x.threshold(tie_k);
        std::printf("survivors");
        for (uint64_t i = 0; i < tie_n; i++) std::printf(" %d", (int)x.get(i));
        std::printf("\n");
    }
    // ---- matrix: mvm == row dots, transpose
    {
        const uint64_t m = 256, n = 384;
        CloverMatrix32 A(m, n), AT(n, m), back(m, n);
        CloverVector32 x(n), y(m), y2(m);
        A.setRandomFloats(-1, 1, 5);
        x.setRandomFloats(-1, 1, 6);
        A.mvm(x, y);
        A.mvm_parallel(x, y2);
        for (uint64_t i = 0; i < m; i++) {
            CHECK(same_bits(y.get(i), dot_avx2(A.getData() + i * n, x.getData(), n)));
            CHECK(same_bits(y.get(i), y2.get(i)));
        }
        A.transpose(AT);
        AT.transpose_parallel(back);
        for (uint64_t i = 0; i < m; i += 7)
            for (uint64_t j = 0; j < n; j += 5) CHECK(same_bits(AT.get(j, i), A.get(i, j)) && same_bits(back.get(i, j), A.get(i, j)));
    }
    // ---- the fp32 baseline loop: Q_IHT on the 32-bit classes recovers a K-sparse vector (03_iht_gd_util.cpp:449-495 shape: Phi m x 2m)
    {
        const uint64_t m = 256, n = 512, K = 16;
        CloverMatrix32 Phi(m, n), PhiT(n, m);
        Phi.setRandomFloats(-1, 1, 77);
        Phi.transpose(PhiT);
        CloverVector32 xs(n), y(m), x(n), t1(m), t2(m), t3(n);
        xs.clear();
        for (uint64_t i = 0; i < K; i++) xs.set((i * 131 + 7) % n, 1.0f);
        Phi.mvm(xs, y);
        Q_IHT(Phi, PhiT, x, y, t1, t2, t3, 600, K, 0.001f);
        double err = 0;
        for (uint64_t i = 0; i < n; i++) err += (double)(x.get(i) - xs.get(i)) * (x.get(i) - xs.get(i));
        std::printf("iht_fp32_rel_err %.3e\n", std::sqrt(err / K));
        CHECK(std::sqrt(err / K) < 1e-3);
    }
    std::printf(failures ? "FAILED %d\n" : "ok\n", failures);
    return failures ? 1 : 0;
}
