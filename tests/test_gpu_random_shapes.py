"""Randomised sweep: GPU == oracle, bit for bit, over shapes and data patterns drawn from fixed seeds -- ragged tails against every
internal chunk size (quantize 2048-element chunks, restore 1024, mvm 65536 / 32768-column LDS chunks are covered elsewhere),
blocks of zeros, tiny and huge magnitudes, both rounding modes."""
import numpy as np
import pytest

from conftest import random_packed

pytestmark = pytest.mark.gpu
same = lambda a, b: a.tobytes() == b.tobytes()      # noqa: E731


def _data(rng, n, kind):
    if kind == 0:
        x = rng.integers(-10, 11, n).astype(np.float32)
    elif kind == 1:
        x = (rng.normal(size=n) * 10.0 ** rng.integers(-6, 7)).astype(np.float32)
    elif kind == 2:
        x = rng.uniform(-1, 1, n).astype(np.float32)
        x[rng.random(n) < 0.7] = 0.0
    else:
        x = (rng.normal(size=n)).astype(np.float32)
        for b in rng.integers(0, n // 64, max(1, n // 640)):          # whole blocks of zeros and of one repeated value
            x[64 * b:64 * b + 64] = 0.0 if b % 2 else np.float32(3.25)
    return x


@pytest.mark.parametrize("seed", range(12))
def test_vector_ops_random(hip, oracle, seed):
    rng = np.random.default_rng(1000 + seed)
    n = 128 * int(rng.integers(1, 400))
    x, y = _data(rng, n, seed % 4), _data(rng, n, (seed + 1) % 4)
    qx, sx = hip.v4_quantize(x)
    qo, so = oracle.v4_quantize(x)
    assert same(qx, qo) and same(sx, so)
    qy, sy = oracle.v4_quantize(y)
    assert same(hip.v4_restore(qx, sx), oracle.v4_restore(qo, so))
    assert np.float32(hip.v4_dot(qx, sx, qy, sy)).tobytes() == np.float32(oracle.v4_dot(qx, sx, qy, sy)).tobytes()
    a = float(rng.uniform(-2, 2))
    r, sr = hip.v4_scale_and_add(qx, sx, qy, sy, a)
    ro, sro = oracle.v4_scale_and_add(qx, sx, qy, sy, a)
    assert same(r, ro) and same(sr, sro)
    q8, s8 = hip.v8_quantize(x)
    q8o, s8o = oracle.v8_quantize(x)
    assert same(q8, q8o) and same(s8, s8o) and same(hip.v8_restore(q8, s8), oracle.v8_restore(q8o, s8o))
    st, o = hip.new_rng(seed + 1, 99 - seed), oracle.rng(seed + 1, 99 - seed)
    for _ in range(2):
        qs, ss = hip.v4_quantize(x, rng=st)
        qso, sso = oracle.v4_quantize(x, o)
        assert same(qs, qso) and same(ss, sso)
        r, sr = hip.v4_scale_and_add(qx, sx, qy, sy, a, rng=st)
        ro, sro = oracle.v4_scale_and_add(qx, sx, qy, sy, a, o)
        assert same(r, ro) and same(sr, sro)
    assert np.array_equal(hip.rng_get(st)[1], oracle.rng_keys(o)[1])


@pytest.mark.parametrize("seed", range(10))
def test_matrix_ops_random(hip, oracle, seed):
    rng = np.random.default_rng(2000 + seed)
    M, N = 128 * int(rng.integers(1, 6)), 128 * int(rng.integers(1, 12))
    A = _data(rng, M * N, seed % 4).reshape(M, N)
    qA, sA = hip.m4_quantize(A)
    qAo, sAo = oracle.m4_quantize(A)
    assert same(qA, qAo) and same(sA, sAo)
    assert same(hip.m4_restore(qA, sA, M, N), oracle.m4_restore(qA, sA, M, N))
    qt, st_ = hip.m4_transpose(qA, sA, M, N)
    qto, sto = oracle.m4_transpose(qA, sA, M, N)
    assert same(qt, qto) and same(st_, sto)
    x = _data(rng, N, (seed + 2) % 4)
    qx, sx = oracle.v4_quantize(x)
    r, sr = hip.m4_mvm(qA, sA, M, N, qx, sx)
    ro, sro = oracle.m4_mvm(qA, sA, M, N, qx, sx)
    assert same(r, ro) and same(sr, sro)
    q8, s8 = oracle.v8_quantize(x)
    r8, sr8 = hip.m4_mvm_v8(qA, sA, M, N, q8, s8)
    r8o, sr8o = oracle.m4_mvm_v8(qA, sA, M, N, q8, s8)
    assert same(r8, r8o) and same(sr8, sr8o)
    assert same(hip.m4_mvm_f32(qA, sA, M, N, x), oracle.m4_mvm_f32(qA, sA, M, N, x))
    g, o = hip.new_rng(5 + seed, 7), oracle.rng(5 + seed, 7)
    qs, ss = hip.m4_quantize(A, rng=g)
    qso, sso = oracle.m4_quantize(A, o)
    assert same(qs, qso) and same(ss, sso)
    r, sr = hip.m4_mvm(qA, sA, M, N, qx, sx, rng=g)
    ro, sro = oracle.m4_mvm(qA, sA, M, N, qx, sx, o)
    assert same(r, ro) and same(sr, sro)
    r8, sr8 = hip.m4_mvm_v8(qA, sA, M, N, q8, s8, rng=g)
    r8o, sr8o = oracle.m4_mvm_v8(qA, sA, M, N, q8, s8, o)
    assert same(r8, r8o) and same(sr8, sr8o)
    assert np.array_equal(hip.rng_get(g)[1], oracle.rng_keys(o)[1])


@pytest.mark.parametrize("seed", range(8))
def test_gemm_random(hip, oracle, seed):
    """GEMM through quantized REAL data (zeros, tiny / huge blocks): every shape class of the FP6 path -- K-block counts that are
    not multiples of the 8 the re-coding pass walks, single-tile and multi-tile M and N."""
    rng = np.random.default_rng(7000 + seed)
    M, N, K = (128 * int(rng.integers(1, 5)) for _ in range(3))
    K = 128 * int(rng.integers(1, 11))
    A = _data(rng, M * K, seed % 4).reshape(M, K)
    B = _data(rng, N * K, (seed + 2) % 4).reshape(N, K)
    qA, sA = oracle.m4_quantize(A)
    qB, sB = oracle.m4_quantize(B)
    C = hip.m4_gemm(qA, sA, M, K, qB, sB, N)
    assert same(C, oracle.m4_gemm(qA, sA, M, K, qB, sB, N))
