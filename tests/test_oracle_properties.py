"""CPU tests of the oracle itself: the relations the reference's own `clover -v` harness asserts
(test/validate/02_vector.cpp, 03_matrix.cpp), edge cases, and fast == scalar."""
import numpy as np
import pytest

from conftest import bits, random_packed


def ints(rng, n, lim):
    return rng.integers(-lim, lim + 1, size=n).astype(np.float32)


@pytest.mark.parametrize("n", [128, 256, 640, 1024, 2048])
def test_quantize_restore_consistency(oracle, n):
    # 02_vector.cpp:181-221: ints in [-7,7], |x - restore(quantize(x))| <= 1
    rng = np.random.default_rng(n)
    x = ints(rng, n, 7)
    q, s = oracle.v4_quantize(x)
    xr = oracle.v4_restore(q, s)
    assert np.max(np.abs(x - xr)) <= 1.0
    for pos in (0, 1, 63, 64, n - 1):
        assert bits(oracle.v4_get(q, s, pos)) == bits(xr[pos])


@pytest.mark.parametrize("n", [128, 384, 1152, 2048])
def test_dot_simd_vs_scalar_tolerance(oracle, n):
    # 02_vector.cpp:258-295: |dot - dot_scalar| <= 0.02 on ints in [-7,7]
    rng = np.random.default_rng(n + 1)
    a, b = oracle.v4_quantize(ints(rng, n, 7)), oracle.v4_quantize(ints(rng, n, 7))
    d, ds = oracle.v4_dot(*a, *b), oracle.v4_dot_scalar(*a, *b)
    assert abs(float(d) - float(ds)) <= 0.02
    assert abs(float(d) - oracle.v4_dot_f64(*a, *b)) <= 0.02


def test_zero_block_and_negative_zero(oracle):
    x = np.zeros(128, np.float32)
    x[64:] = np.linspace(-3, 3, 64, dtype=np.float32)
    x[70] = -0.0
    q, s = oracle.v4_quantize(x)
    assert s[0] == 1.0 and not q[:32].any()          # all-zero block: scale 1.0, bytes 0 (CloverVector4.h:661-663)
    assert s[1] == 3.0
    assert (q[32 + 3] & 0xF0) == 0                    # element 70 (-0.0) -> nibble 0


def test_quantized_range_and_truncation(oracle):
    rng = np.random.default_rng(7)
    x = (rng.normal(size=4096) * 10).astype(np.float32)
    q, s = oracle.v4_quantize(x)
    hi = (q.astype(np.int8) >> 4).astype(np.int32)
    lo = ((q << 4).astype(np.int8) >> 4).astype(np.int32)
    assert hi.min() >= -7 and hi.max() <= 7 and lo.min() >= -7 and lo.max() <= 7
    # truncation, not rounding: |q| == floor(|x| * (7/max))
    k = np.float32(7.0) / np.repeat(s, 64)
    expect = np.floor(np.abs(x).astype(np.float32) * k).astype(np.int32)
    got = np.abs(np.stack([hi, lo], 1).reshape(-1))
    assert np.array_equal(got, np.minimum(expect, 7))


def test_word_isums_match_bruteforce(oracle):
    rng = np.random.default_rng(3)
    (qu, _), (qv, _) = random_packed(rng, 512), random_packed(rng, 512)
    I = oracle.v4_word_isums(qu, qv)

    def nibbles(b):
        hi = (b.astype(np.int8) >> 4).astype(np.int32)
        lo = ((b << 4).astype(np.int8) >> 4).astype(np.int32)
        return np.stack([hi, lo], 1).reshape(-1)
    ref = (nibbles(qu) * nibbles(qv)).reshape(-1, 8).sum(1)
    assert np.array_equal(I, ref)


@pytest.mark.parametrize("shape", [(128, 128), (256, 384), (384, 128)])
def test_mvm_is_rowdots_plus_requantize(oracle, shape):
    M, N = shape
    rng = np.random.default_rng(M * 7 + N)
    A = ints(rng, M * N, 10).reshape(M, N)
    qA, sA = oracle.m4_quantize(A)
    qx = oracle.v4_quantize(ints(rng, N, 10))
    d = oracle.m4_rowdots(qA, sA, M, N, *qx)
    r, sr = oracle.m4_mvm(qA, sA, M, N, *qx)
    r2, sr2 = oracle.v4_quantize(d)                    # epilogue == vector quantize of the 64 dots
    assert np.array_equal(r, r2) and np.array_equal(bits(sr), bits(sr2))
    # each row dot is the vector dot of the row view (what mvm_scalar does, CloverMatrix4.h:338-342)
    hb = N // 64
    for row in (0, 63, 64, M - 1):
        dv = oracle.v4_dot(qA[row * N // 2:(row + 1) * N // 2], sA[(row >> 6) * hb:(row >> 6) * hb + hb], *qx)
        assert bits(dv) == bits(d[row])


def test_matrix_quantize_consistency(oracle):
    # 03_matrix.cpp:99-149
    rng = np.random.default_rng(11)
    M, N = 256, 128
    A = ints(rng, M * N, 7).reshape(M, N)
    q, s = oracle.m4_quantize(A)
    for (i, j) in [(0, 0), (63, 127), (64, 0), (255, 127), (100, 65)]:
        assert abs(float(oracle.m4_get(q, s, M, N, i, j)) - A[i, j]) <= 1.0


def test_gemm_definition(oracle):
    rng = np.random.default_rng(5)
    M, N, K = 128, 128, 256
    qA, sA = oracle.m4_quantize(ints(rng, M * K, 10).reshape(M, K))
    qB, sB = oracle.m4_quantize(ints(rng, N * K, 10).reshape(N, K))
    C = oracle.m4_gemm(qA, sA, M, K, qB, sB, N)
    S = oracle.m4_gemm_isums(qA, M, K, qB, N)
    kb = K // 64
    for (i, j) in [(0, 0), (5, 77), (127, 127), (64, 63)]:
        acc = np.float32(0)
        for b in range(kb):
            c = np.float32(np.float32(sA[(i >> 6) * kb + b] * np.float32(1.0 / 49.0)) * sB[(j >> 6) * kb + b])
            acc = np.float32(np.float64(c) * np.float64(S[i, j, b]) + np.float64(acc))   # exact product + 1 rounding
        assert bits(acc) == bits(C[i, j])
        # and it is the same number, up to fp32 order, as the 16-chain dot of the two rows
        d = oracle.v4_dot(qA[i * K // 2:(i + 1) * K // 2], sA[(i >> 6) * kb:(i >> 6) * kb + kb],
                          qB[j * K // 2:(j + 1) * K // 2], sB[(j >> 6) * kb:(j >> 6) * kb + kb])
        assert abs(float(d) - float(C[i, j])) <= 1e-5 * max(1.0, abs(float(d)))


def test_stochastic_stream(oracle):
    r = oracle.rng(12345, 67890)
    k1, k2 = oracle.rng_keys(r)
    assert k1[0] == 12345 and k2[0] == 67890 and len(set(k1.tolist())) == 4
    # the draw only depends on part2 (simdxorshift128plus.h:97-109)
    a = int(k2[0])
    m = (1 << 64) - 1
    t = (a ^ (a << 23)) & m
    n = t ^ a ^ (t >> 18) ^ (a >> 5)
    w = oracle.rng_draw(r)
    out = (n + a) & m
    assert int(w[0]) == (out & 0xFFFFFFFF) and int(w[1]) == (out >> 32)
    # stochastic quantize: in range, unbiased-ish, never below truncation and at most +1
    rng = np.random.default_rng(0)
    x = rng.uniform(-1, 1, size=1 << 14).astype(np.float32)
    q0, s0 = oracle.v4_quantize(x)
    q1, s1 = oracle.v4_quantize(x, oracle.rng(1, 2))
    assert np.array_equal(s0, s1)

    def nib(b):
        hi = (b.astype(np.int8) >> 4).astype(np.int32)
        lo = ((b << 4).astype(np.int8) >> 4).astype(np.int32)
        return np.stack([hi, lo], 1).reshape(-1)
    d = np.abs(nib(q1)) - np.abs(nib(q0))
    assert d.min() >= 0 and d.max() <= 1 and np.abs(nib(q1)).max() <= 7
    xr = oracle.v4_restore(q1, s1)
    assert abs(float(np.mean(xr - x))) < 5e-3                 # stochastic rounding is (nearly) unbiased


@pytest.mark.parametrize("kernel", ["maddubs", "planes"])
@pytest.mark.parametrize("n", [128, 1024, 4096 + 128])
def test_fast_oracle_equals_scalar(oracle, fast_oracle, n, kernel):
    fast_oracle.set_kernel(kernel)
    rng = np.random.default_rng(n)
    x, y = (rng.normal(size=n) * 3).astype(np.float32), ints(rng, n, 10)
    a, b = oracle.v4_quantize(x), oracle.v4_quantize(y)
    fa, fb = fast_oracle.v4_quantize(x), fast_oracle.v4_quantize(y)
    assert np.array_equal(a[0], fa[0]) and np.array_equal(bits(a[1]), bits(fa[1])) and np.array_equal(b[0], fb[0])
    assert bits(oracle.v4_dot(*a, *b)) == bits(fast_oracle.v4_dot(*a, *b))
    # every nibble pair incl. the extremes (+-7 x +-7): the 16x-scaled bytes of the maddubs kernel must not saturate
    q = np.array([((a_ & 0xF) << 4) | (b_ & 0xF) for a_ in range(-7, 8) for b_ in range(-7, 8)] * 2, np.uint8)[: 128 * 3 // 2 * 2 // 2 * 2]
    q = np.resize(q, 64 * 4)
    s = np.ones(q.size * 2 // 64, np.float32)
    for other in (q, q[::-1].copy(), np.full_like(q, 0x77), np.full_like(q, 0x99)):
        assert bits(oracle.v4_dot(q, s, other, s)) == bits(fast_oracle.v4_dot(q, s, other, s))
    fast_oracle.set_kernel("maddubs")


def test_fast_oracle_mvm_equals_scalar(oracle, fast_oracle):
    rng = np.random.default_rng(2)
    for (M, N) in ((128, 128), (256, 640)):
        qA, sA = oracle.m4_quantize(rng.normal(size=(M, N)).astype(np.float32))
        qx = oracle.v4_quantize(rng.normal(size=N).astype(np.float32))
        r0, r1 = oracle.m4_mvm(qA, sA, M, N, *qx), fast_oracle.m4_mvm(qA, sA, M, N, *qx)
        assert np.array_equal(r0[0], r1[0]) and np.array_equal(bits(r0[1]), bits(r1[1]))


@pytest.mark.parametrize("kernel", ["maddubs", "planes"])
def test_fast_oracle_gemm_equals_scalar(oracle, fast_oracle, kernel):
    """orcf_m4_gemm (AVX2 + OpenMP, 8 columns per fma) is the checker of the WHOLE GEMM results at 4096^3 / 8192^3: it must itself be
    the scalar definition bit for bit, incl. all-+-7 operands (block sums +-3136) and scales 30 decades apart"""
    fast_oracle.set_kernel(kernel)
    rng = np.random.default_rng(8)
    for (M, N, K) in ((128, 128, 128), (256, 384, 640), (128, 256, 1024)):
        A = (rng.normal(size=(M, K)) * 10.0 ** rng.integers(-15, 15, size=(M, 1))).astype(np.float32)
        B = rng.normal(size=(N, K)).astype(np.float32)
        qA, sA = oracle.m4_quantize(A)
        qB, sB = oracle.m4_quantize(B)
        assert np.array_equal(bits(oracle.m4_gemm(qA, sA, M, K, qB, sB, N)), bits(fast_oracle.m4_gemm(qA, sA, M, K, qB, sB, N)))
    q7, q9 = np.full(128 * 128 // 2, 0x77, np.uint8), np.full(128 * 128 // 2, 0x99, np.uint8)
    s = np.full(4, 3.0, np.float32)
    for a, b in ((q7, q7), (q7, q9), (q9, q9)):
        assert np.array_equal(bits(oracle.m4_gemm(a, s, 128, 128, b, s, 128)), bits(fast_oracle.m4_gemm(a, s, 128, 128, b, s, 128)))
    fast_oracle.set_kernel("maddubs")


def test_fast_oracle_dot_parallel_is_tolerance_only(oracle, fast_oracle):
    """dot_parallel's decomposition (CloverVector4.h:1793-1907): thread partials summed in unspecified order -- the reference's own
    check allows 0.02 absolute on small vectors (test/validate/02_vector.cpp); here relative 1e-5 of the sum of magnitudes"""
    rng = np.random.default_rng(9)
    n = 1 << 16
    a, b = oracle.v4_quantize(ints(rng, n, 10)), oracle.v4_quantize(ints(rng, n, 10))
    d, dp = float(fast_oracle.v4_dot(*a, *b)), float(fast_oracle.v4_dot_parallel(*a, *b))
    mag = float(np.abs(oracle.v4_restore(*a) * oracle.v4_restore(*b)).sum())
    assert abs(d - dp) <= 1e-5 * mag
    before = fast_oracle.max_threads()
    fast_oracle.set_threads(1)
    try:
        assert bits(fast_oracle.v4_dot_parallel(*a, *b)) == bits(fast_oracle.v4_dot(*a, *b))   # one thread: dot's own order
    finally:
        fast_oracle.set_threads(before)
