// validate_relations.cpp -- the reference's validation harness relations (test/validate/02_vector.cpp, 03_matrix.cpp) against the
// drop-in containers: the device method on one side, its scalar host twin (include/clover_scalar.h) on the other, at the same
// strictness the reference uses (exact where it is exact there).  Built with -DCLOVER_STOCHASTIC_ROUNDING_DISABLED=1, as the
// reference's own exact checks require ("skip" otherwise).
#include <cmath>
#include <cstdio>

#include "CloverMatrix4.h"
#include "CloverVector4.h"
#include "CloverVector8.h"

static int failures = 0;
static void expect(bool ok, const char *what, uint64_t a, uint64_t b)
{
    if (!ok) { std::printf("FAILED %s (%llu, %llu)\n", what, (unsigned long long)a, (unsigned long long)b); failures++; }
}

int main()
{
    int ndev = 0;
    if (clv_device_count(&ndev) != CLV_OK || ndev == 0) { std::printf("no_device\n"); return 0; }
    // ---- vectors (02_vector.cpp:111-447) ---------------------------------------------------------------------------
    for (uint64_t n = 128; n < 2048; n += 97) {
        CloverVector32 x(n), y(n), r1(n), r2(n);
        x.setRandomInteger(10, 1000 + n);
        y.setRandomInteger(7, 2000 + n);
        CloverVector4 q(n), qs(n), qy(n);
        q.quantize(x);
        qs.quantize_scalar(x);
        for (uint64_t i = 0; i < n; i++) expect(q.get(i) == qs.get(i), "quantize vs quantize_scalar", n, i);                      // :111-144
        q.restore(r1);
        q.restore_scalar(r2);
        for (uint64_t i = 0; i < n; i++) expect(r1.get(i) == r2.get(i), "restore vs restore_scalar", n, i);                      // :223-256
        qy.quantize(y);
        qy.restore(r1);
        for (uint64_t i = 0; i < n; i++) expect(std::fabs(y.get(i) - r1.get(i)) <= 1.0f, "quantize -> restore consistency", n, i);   // :181-221
        CloverVector4 qa(n), qb(n);
        qa.quantize(y);
        y.setRandomInteger(7, 3000 + n);
        qb.quantize(y);
        expect(std::fabs(qa.dot(qb) - qa.dot_scalar(qb)) <= 0.02f, "dot vs dot_scalar", n, 0);                                   // :258-295
        expect(std::fabs(qa.dot_parallel(qb) - qa.dot_scalar(qb)) <= 0.02f, "dot_parallel vs dot_scalar", n, 0);                 // :298-339
        CloverVector4 s1(qa), s2(qa), s3(n), s4(n);
        s1.scaleAndAdd(qb, 0.5f);
        s2.scaleAndAdd_scalar(qb, 0.5f);
        for (uint64_t i = 0; i < n; i++) expect(s1.get(i) == s2.get(i), "scaleAndAdd vs scalar (in place)", n, i);                // :341-393
        qa.scaleAndAdd(qb, 0.5f, s3);
        qa.scaleAndAdd_scalar(qb, 0.5f, s4);
        for (uint64_t i = 0; i < n; i++) expect(s3.get(i) == s4.get(i), "scaleAndAdd vs scalar (3 operands)", n, i);
        // 8-bit container, same relations
        CloverVector8 e(n), es(n), e1(n), e2(n);
        e.quantize(x);
        es.quantize_scalar(x);
        for (uint64_t i = 0; i < n; i++) expect(e.getBits(i) == es.getBits(i) && e.get(i) == es.get(i), "8-bit quantize vs scalar", n, i);
        e.restore(r1);
        e.restore_scalar(r2);
        for (uint64_t i = 0; i < n; i++) expect(r1.get(i) == r2.get(i), "8-bit restore vs scalar", n, i);
        CloverVector8 f(n);
        f.quantize(y);
        e.scaleAndAdd(f, 0.5f, e1);
        e.scaleAndAdd_scalar(f, 0.5f, e2);
        for (uint64_t i = 0; i < n; i++) expect(e1.getBits(i) == e2.getBits(i), "8-bit scaleAndAdd vs scalar", n, i);
    }
    // ---- matrices (03_matrix.cpp:38-573) -----------------------------------------------------------------------------
    for (uint64_t bi = 1; bi <= 3; bi++)
        for (uint64_t bj = 1; bj <= 3; bj++) {
            const uint64_t M = 128 * bi, N = 128 * bj;
            CloverMatrix32 A(M, N), R1(M, N), R2(M, N);
            A.setRandomInteger(10, 77 * bi + bj);
            CloverMatrix4 qA(M, N), qS(M, N);
            qA.quantize(A);
            qS.quantize_scalar(A);
            for (uint64_t i = 0; i < M; i++)
                for (uint64_t j = 0; j < N; j++) expect(qA.get(i, j) == qS.get(i, j), "matrix quantize vs scalar", i, j);         // :38-96
            qA.restore(R1);
            qA.restore_scalar(R2);
            for (uint64_t i = 0; i < M; i += 7)
                for (uint64_t j = 0; j < N; j++) expect(R1.get(i, j) == R2.get(i, j), "matrix restore vs scalar", i, j);
            CloverVector32 x(N);
            x.setRandomInteger(10, 5 * bi + bj);
            CloverVector4 qx(N), r(M), rp(M), rs(M);
            qx.quantize(x);
            qA.mvm(qx, r);
            qA.mvm_parallel(qx, rp);
            qA.mvm_scalar(qx, rs);
            for (uint64_t k = 0; k < M; k++) {
                expect(r.get(k) == rs.get(k), "mvm vs mvm_scalar", M, k);                                                    // :248-326
                expect(r.get(k) == rp.get(k), "mvm vs mvm_parallel", M, k);                                                  // :495-573
            }
            // mixed precision (:328-491): relative 1.6 % or one 8-bit step; |delta| <= 0.01 against the double-accumulated scalar
            CloverVector8 x8(N), y8(M), y8s(M);
            x8.quantize(x);
            qA.mvm(x8, y8);
            qA.mvm_scalar(x8, y8s);
            for (uint64_t k = 0; k < M; k++) {
                const float a = y8.get(k), b = y8s.get(k), step = y8s.getScales()[k >> 6] / 127.0f;
                expect(std::fabs(a - b) <= 0.016f * std::fabs(b) + step, "4b x 8b mvm vs mvm_scalar", M, k);
            }
            CloverVector32 xs(N), y32(M), y32s(M);
            for (uint64_t j = 0; j < N; j++) xs.set(j, x.get(j) * 0.001f);
            qA.mvm(xs, y32);
            qA.mvm_scalar(xs, y32s);
            for (uint64_t k = 0; k < M; k++) expect(std::fabs(y32.get(k) - y32s.get(k)) <= 0.01f, "4b x fp32 mvm vs mvm_scalar", M, k);
            CloverMatrix4 T(N, M), Ts(N, M);
            qA.transpose(T);
            qA.transpose_scalar(Ts);
            for (uint64_t i = 0; i < M; i++)
                for (uint64_t j = 0; j < N; j++) {
                    expect(qA.get(i, j) == T.get(j, i), "transpose", i, j);                                                  // :153-197
                    expect(T.get(j, i) == Ts.get(j, i), "transpose vs transpose_scalar", i, j);
                }
        }
    std::printf(failures ? "validate FAILED (%d)\n" : "validate ok\n", failures);
    return failures ? 1 : 0;
}
