"""mixed precision, CloverMatrix4 x CloverVector8 (CloverMatrix4.h:1093-1441) and the CloverVector8 quantize /
restore it needs (CloverVector8.h:393-606, 835-909).

The reference holds no golden vectors for this path; its own check is relational -- SIMD mvm against mvm_scalar (double
accumulation) within 1.6 % relative or one quantisation step (test/validate/03_matrix.cpp).  The CPU tests below
assert exactly that relation for the oracle's SIMD-order restatement; the GPU tests assert GPU == oracle bit for bit."""
import numpy as np
import pytest

from conftest import random_packed

same = lambda a, b: a.tobytes() == b.tobytes()      # noqa: E731


def _inputs(rng, M, N):
    qA, _ = random_packed(rng, M * N)
    sA = rng.uniform(0.5, 2, size=(M // 64) * (N // 64)).astype(np.float32)
    x = (rng.normal(size=N) * 3).astype(np.float32)
    return qA, sA, x


# ---------------------------------------------------------------- oracle (CPU)
def test_oracle_v8_quantize_restore_roundtrip(oracle):
    rng = np.random.default_rng(1)
    x = (rng.normal(size=1024) * 5).astype(np.float32)
    x[64:128] = 0.0                                  # all-zero block -> scale 1.0
    x[130] = -0.0
    q, s = oracle.v8_quantize(x)
    assert s[1] == 1.0 and not q[64:128].any()
    assert np.abs(q.astype(np.int32)).max() == 127 and q.min() >= -127
    for b in range(x.size // 64):                    # block maximum maps to +-127, truncation toward zero elsewhere
        blk = x[64 * b:64 * b + 64]
        if np.abs(blk).max() > 0:
            assert s[b] == np.abs(blk).max()
            assert abs(int(q[64 * b + np.abs(blk).argmax()])) in (126, 127)       # trunc(max * f32(127/max)) may land just below 127
    xr = oracle.v8_restore(q, s)
    assert np.all(np.abs(x - xr) <= np.repeat(s, 64) / np.float32(127.0) + 1e-6)
    assert np.all(np.abs(xr) <= np.abs(x) + 1e-6)                      # truncation never grows a magnitude
    # reference definition of one element (CloverVector8::get, :137-140)
    assert xr[5] == np.float32(np.float32(q[5]) * (s[0] / np.float32(127.0)))


@pytest.mark.parametrize("shape", [(128, 128), (128, 512), (256, 1152)])
def test_oracle_mixed_mvm_simd_order_vs_scalar_double(oracle, shape):
    M, N = shape
    rng = np.random.default_rng(M + N)
    qA, sA, x = _inputs(rng, M, N)
    qx, sx = oracle.v8_quantize(x)
    d = oracle.m4_rowdots_v8(qA, sA, M, N, qx, sx)
    d64 = oracle.m4_rowdots_v8(qA, sA, M, N, qx, sx, f64=True)
    assert np.allclose(d, d64, rtol=2e-5, atol=2e-5 * np.abs(d64).max())
    r, sr = oracle.m4_mvm_v8(qA, sA, M, N, qx, sx)
    # the reference's own acceptance test for this path: restored result vs scalar result, 1.6 % or one step
    rr = oracle.v8_restore(r, sr)
    step = np.repeat(sr, 64) / np.float32(127.0)
    assert np.all((np.abs(rr - d64) <= 0.016 * np.abs(d64)) | (np.abs(rr - d64) <= step + 1e-6))
    # block scale = block maximum of the row dots
    assert same(sr, np.abs(d.reshape(-1, 64)).max(axis=1).astype(np.float32))


def test_oracle_mixed_mvm_stochastic_stays_within_one_step(oracle):
    rng = np.random.default_rng(9)
    M, N = 128, 256
    qA, sA, x = _inputs(rng, M, N)
    qx, sx = oracle.v8_quantize(x)
    r0, sr0 = oracle.m4_mvm_v8(qA, sA, M, N, qx, sx)
    r1, sr1 = oracle.m4_mvm_v8(qA, sA, M, N, qx, sx, oracle.rng(3, 4))
    assert same(sr0, sr1)
    diff = r1.astype(np.int32) - r0.astype(np.int32)
    assert np.all(np.abs(diff) <= 1) and np.any(diff != 0)
    assert np.all(np.abs(r1.astype(np.int32)) >= np.abs(r0.astype(np.int32)))       # noise only rounds magnitudes up


# ---------------------------------------------------------------- GPU == oracle
@pytest.mark.gpu
@pytest.mark.parametrize("n", [128, 1024, 8192 + 128, (1 << 17) + 384, (1 << 21) + 256])
def test_gpu_v8_quantize_restore_exact(hip, oracle, n):
    rng = np.random.default_rng(n)
    for x in ((rng.normal(size=n) * 4).astype(np.float32), rng.integers(-300, 301, n).astype(np.float32)):
        x[:64] = 0.0
        q, s = hip.v8_quantize(x)
        qo, so = oracle.v8_quantize(x)
        assert same(q, qo) and same(s, so)
        assert same(hip.v8_restore(q, s), oracle.v8_restore(qo, so))


@pytest.mark.gpu
@pytest.mark.parametrize("segments", [0, 1, 4, 16])
def test_gpu_v8_quantize_stochastic_same_stream(hip, oracle, segments):
    n = 64 * (32 * max(segments, 1) * 5 + 7 * max(segments, 1) + 3)
    n += (-n) % 128
    rng = np.random.default_rng(n)
    x = (rng.normal(size=n) * 2).astype(np.float32)
    assert hip.lib.clvx_set_st_segments(segments) == 0
    try:
        st, o = hip.new_rng(21, 43), oracle.rng(21, 43)
        for _ in range(2):
            q, s = hip.v8_quantize(x, rng=st)
            qo, so = oracle.v8_quantize(x, o)
            assert same(q, qo) and same(s, so)
        assert np.array_equal(hip.rng_get(st)[1], oracle.rng_keys(o)[1])
    finally:
        hip.lib.clvx_set_st_segments(0)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(128, 128), (128, 256), (256, 1152), (1024, 32768 + 128), (192 * 2, 65536 + 384)])
def test_gpu_mixed_mvm_exact(hip, oracle, shape):
    M, N = shape
    rng = np.random.default_rng(M * 7 + N)
    qA, sA, x = _inputs(rng, M, N)
    qx, sx = oracle.v8_quantize(x)
    assert same(hip.m4_rowdots_v8(qA, sA, M, N, qx, sx), oracle.m4_rowdots_v8(qA, sA, M, N, qx, sx))
    r, sr = hip.m4_mvm_v8(qA, sA, M, N, qx, sx)
    ro, sro = oracle.m4_mvm_v8(qA, sA, M, N, qx, sx)
    assert same(r, ro) and same(sr, sro)


@pytest.mark.gpu
def test_gpu_mixed_mvm_stochastic_same_stream(hip, oracle):
    rng = np.random.default_rng(77)
    M, N = 384, 640
    qA, sA, x = _inputs(rng, M, N)
    st, o = hip.new_rng(5, 9), oracle.rng(5, 9)
    qx, sx = hip.v8_quantize(x, rng=st)
    qxo, sxo = oracle.v8_quantize(x, o)
    assert same(qx, qxo) and same(sx, sxo)
    for _ in range(2):
        r, sr = hip.m4_mvm_v8(qA, sA, M, N, qx, sx, rng=st)
        ro, sro = oracle.m4_mvm_v8(qA, sA, M, N, qx, sx, o)
        assert same(r, ro) and same(sr, sro)
    assert np.array_equal(hip.rng_get(st)[1], oracle.rng_keys(o)[1])


@pytest.mark.gpu
def test_gpu_mixed_mvm_extreme_values(hip, oracle):
    """every nibble -7 / +7 against every byte -127 / +127: the largest block integers, and the sign handling"""
    M, N = 128, 256
    for a, b in ((7, 127), (-7, 127), (7, -127), (-7, -127)):
        qA = np.full(M * N // 2, ((a & 0xF) << 4) | (a & 0xF), np.uint8)
        sA = np.full((M // 64) * (N // 64), 1.5, np.float32)
        qx = np.full(N, b, np.int8)
        sx = np.full(N // 64, 0.75, np.float32)
        r, sr = hip.m4_mvm_v8(qA, sA, M, N, qx, sx)
        ro, sro = oracle.m4_mvm_v8(qA, sA, M, N, qx, sx)
        assert same(r, ro) and same(sr, sro)
        assert abs(int(r[0])) == 127 and (int(r[0]) > 0) == ((a > 0) == (b > 0))
