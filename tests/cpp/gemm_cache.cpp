// gemm_cache.cpp -- CloverMatrix4::cacheGemmOperand(): the cached FP6 operand image must follow every change of the matrix
// (quantize, a write through a kept getData() pointer) and never change a result.  Prints "gemm_cache ok" or the first mismatch.
#include <cstdio>
#include <cstring>
#include <vector>

#include <CloverMatrix4.h>

static void fill(CloverMatrix32 &m, unsigned seed)
{
    float *p = m.getData();
    unsigned s = seed;
    for (uint64_t i = 0; i < m.getRows() * m.getCols(); i++) {
        s = s * 1664525u + 1013904223u;
        p[i] = (float)((int)(s >> 20) % 201 - 100) * 0.013f;
    }
}

static bool same(const CloverMatrix32 &a, const CloverMatrix32 &b)
{
    return memcmp(a.getData(), b.getData(), a.getRows() * a.getCols() * sizeof(float)) == 0;
}

int main()
{
    int ndev = 0;
    if (clv_device_count(&ndev) != CLV_OK || ndev == 0) { std::printf("no_device\n"); return 0; }
    const uint64_t M = 256, N = 384, K = 640;
    CloverMatrix32 a32(M, K), b32(N, K), a32b(M, K);
    fill(a32, 1); fill(b32, 2); fill(a32b, 3);
    CloverMatrix4 A(M, K), B(N, K), Aref(M, K), Bref(N, K);
    A.quantize(a32); B.quantize(b32); Aref.quantize(a32); Bref.quantize(b32);
    CloverMatrix32 C0(M, N), C1(M, N);

    Aref.gemm(Bref, C0);                                   // uncached reference
    A.cacheGemmOperand(); B.cacheGemmOperand();
    if (A.gemmOperandCached()) { std::printf("cached before the first gemm\n"); return 1; }
    A.gemm(B, C1);
    if (!same(C0, C1)) { std::printf("cached gemm differs from the uncached one\n"); return 1; }
    if (!A.gemmOperandCached() || !B.gemmOperandCached()) { std::printf("images not kept\n"); return 1; }
    C1.clear();
    A.gemm(B, C1);                                         // both images re-used
    if (!same(C0, C1) || !A.gemmOperandCached()) { std::printf("second cached gemm differs\n"); return 1; }
    A.gemm(Bref, C1);                                      // one cached operand, one raw
    if (!same(C0, C1)) { std::printf("mixed cached / raw gemm differs\n"); return 1; }

    // 1. the matrix is re-quantised: the image must be rebuilt
    A.quantize(a32b); Aref.quantize(a32b);
    if (A.gemmOperandCached()) { std::printf("image survived quantize()\n"); return 1; }
    Aref.gemm(Bref, C0); A.gemm(B, C1);
    if (!same(C0, C1)) { std::printf("stale image after quantize()\n"); return 1; }

    // 2. a write through a pointer taken earlier (the reference's raw-pointer style)
    CloverMatrix32 D0(N, M), D1(N, M);
#ifndef CLOVER_HIP_NO_PAGE_TRACKING
    int8_t *pb = B.getData();                              // taken BEFORE the device operation below and kept across it
    int8_t *pr = Bref.getData();
    B.gemm(A, D1);                                         // device copies current, images cached (roles swapped: N x M result)
#else
    B.gemm(A, D1);
    int8_t *pb = B.getData();                              // untracked build: a pointer is valid until the next device operation
    int8_t *pr = Bref.getData();
#endif
    pb[7] = (int8_t)(pb[7] ^ 0x35); pr[7] = (int8_t)(pr[7] ^ 0x35);
    B.getScales()[1] *= 1.5f; Bref.getScales()[1] *= 1.5f;
    Bref.gemm(Aref, D0); B.gemm(A, D1);
    if (!same(D0, D1)) { std::printf("stale image after a write through getData()\n"); return 1; }

    // 3. switching the cache off releases the image and changes nothing
    A.cacheGemmOperand(false);
    if (A.gemmOperandCached()) { std::printf("image kept after cacheGemmOperand(false)\n"); return 1; }
    B.gemm(A, D1);
    if (!same(D0, D1)) { std::printf("result changed after cacheGemmOperand(false)\n"); return 1; }
    // 4. a large and a small cached operand: their images are in different staging layouts (256- and 128-row tiles; the layout follows
    //    an operand's own row count), and the call re-codes the smaller one -- same bits as the uncached product, either way round
    {
        const uint64_t ML = 4096, NS = 256, KL = 256;
        CloverMatrix32 l32(ML, KL), s32(NS, KL);
        fill(l32, 5); fill(s32, 6);
        CloverMatrix4 L(ML, KL), S(NS, KL), Lref(ML, KL), Sref(NS, KL);
        L.quantize(l32); S.quantize(s32); Lref.quantize(l32); Sref.quantize(s32);
        CloverMatrix32 E0(ML, NS), E1(ML, NS), F0(NS, ML), F1(NS, ML);
        Lref.gemm(Sref, E0); Sref.gemm(Lref, F0);
        L.cacheGemmOperand(); S.cacheGemmOperand();
        L.gemm(S, E1);
        if (!same(E0, E1)) { std::printf("large x small cached gemm differs from the uncached one\n"); return 1; }
        S.gemm(L, F1);
        if (!same(F0, F1)) { std::printf("small x large cached gemm differs from the uncached one\n"); return 1; }
        if (!L.gemmOperandCached() || !S.gemmOperandCached()) { std::printf("images of the mixed pair not kept\n"); return 1; }
    }
    std::printf("gemm_cache ok\n");
    return 0;
}
