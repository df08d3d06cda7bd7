"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on identical inputs.

Bar: bit-exact for every byte, scale and fp32 result produced in the reference's order (quantize, restore,
word sums, dot EXACT, mvm, rowdots, GEMM per-element chain); dot FAST is the only tolerance-based check
(tolerance written at the assert).  Sizes follow the reference's validation grid plus
ragged / edge cases; big configurations are covered by properties in test_gpu_large.py.
"""
import json
from pathlib import Path

import ctypes as C

import numpy as np
import pytest

from conftest import bits, kat2_inputs, kat3_inputs, random_packed

pytestmark = pytest.mark.gpu
KAT = json.loads((Path(__file__).parent / "golden" / "kat_reference.json").read_text())


def ints(rng, n, lim):
    return rng.integers(-lim, lim + 1, size=n).astype(np.float32)


def same(a, b):
    return np.array_equal(np.asarray(a).view(np.uint8), np.asarray(b).view(np.uint8))


# ---------------------------------------------------------------- known answers straight on the GPU
def test_gpu_kat1_kat2(hip):
    from clover_amd.lib_binding import DOT_EXACT
    a = hip.v4_quantize(np.full(128, 1.0, np.float32))
    b = hip.v4_quantize(np.full(128, 2.0, np.float32))
    assert set(a[0].tolist()) == {0x77} and a[1].tolist() == [1.0, 1.0] and b[1].tolist() == [2.0, 2.0]
    assert hex(bits(hip.v4_dot(*a, *b, mode=DOT_EXACT))) == KAT["KAT1"]["dot_bits"]
    x, y = kat2_inputs()
    qx, qy = hip.v4_quantize(x), hip.v4_quantize(y)
    assert qx[0][:32].tobytes().hex() == KAT["KAT2"]["qx_bytes_0_31"]
    assert qy[0][:32].tobytes().hex() == KAT["KAT2"]["qy_bytes_0_31"]
    assert hex(bits(hip.v4_dot(*qx, *qy, mode=DOT_EXACT))) == KAT["KAT2"]["dot_bits"]
    assert [hex(v) for v in bits(hip.v4_restore(*qx)[:4])] == KAT["KAT2"]["restore_qx_0_3_bits"]


def test_gpu_kat3(hip):
    A, x = kat3_inputs()
    M, N = A.shape
    qA, sA = hip.m4_quantize(A)
    qx = hip.v4_quantize(x)
    r, sr = hip.m4_mvm(qA, sA, M, N, *qx)
    assert r.tobytes().hex() == KAT["KAT3"]["r_bytes_0_63"]
    assert [hex(v) for v in bits(sr)] == KAT["KAT3"]["r_scale_bits"]


# ---------------------------------------------------------------- vector ops vs oracle
@pytest.mark.parametrize("n", [128, 256, 384, 1024, 2048 + 128, 65536, (1 << 20) + 128])
def test_quantize_restore_bit_exact(hip, oracle, n):
    rng = np.random.default_rng(n)
    for x in (ints(rng, n, 10), (rng.normal(size=n) * 4).astype(np.float32), rng.uniform(-1e-3, 1e-3, n).astype(np.float32)):
        x[n // 3] = 0.0
        x[n // 2] = -0.0
        q, s = hip.v4_quantize(x)
        qo, so = oracle.v4_quantize(x)
        assert same(q, qo) and same(s, so)
        assert same(hip.v4_restore(q, s), oracle.v4_restore(q, s))


def test_quantize_edge_blocks(hip, oracle):
    x = np.zeros(512, np.float32)
    x[64:128] = 1e-30                      # tiny but non-zero block
    x[128:192] = np.float32(3.4e38)        # huge block
    x[192:256] = -np.arange(64, dtype=np.float32)
    x[300] = np.float32(1e-45)             # a denormal max
    q, s = hip.v4_quantize(x)
    qo, so = oracle.v4_quantize(x)
    assert same(q, qo) and same(s, so)
    assert s[0] == 1.0 and not q[:32].any()


@pytest.mark.parametrize("n", [128, 256, 640, 2048, 4096 + 128, 1 << 16])
def test_word_isums_and_dot_exact(hip, oracle, n):
    from clover_amd.lib_binding import DOT_EXACT, DOT_FAST
    rng = np.random.default_rng(n + 17)
    (qu, su), (qv, sv) = random_packed(rng, n), random_packed(rng, n)
    assert np.array_equal(hip.v4_word_isums(qu, qv), oracle.v4_word_isums(qu, qv))
    d = hip.v4_dot(qu, su, qv, sv, mode=DOT_EXACT)
    assert bits(d) == bits(oracle.v4_dot(qu, su, qv, sv))
    # FAST: same exact block integers, different fp32 summation order.  Tolerance: 2e-6 * sum|terms|
    # (the reference's own SIMD-vs-scalar check allows 0.02 absolute, 02_vector.cpp:284)
    f = hip.v4_dot(qu, su, qv, sv, mode=DOT_FAST)
    I = oracle.v4_word_isums(qu, qv).reshape(-1, 8).sum(1)
    mag = float(np.sum(np.abs(I) * su.astype(np.float64) * sv.astype(np.float64) / 49.0))
    assert abs(float(f) - oracle.v4_dot_f64(qu, su, qv, sv)) <= 2e-6 * mag + 1e-6


def test_dot_on_quantized_ints(hip, oracle):
    # the reference's dot test data: ints in [-7,7] (02_vector.cpp:258-295)
    from clover_amd.lib_binding import DOT_EXACT, DOT_FAST
    rng = np.random.default_rng(99)
    for n in (128, 1152, 2048):
        a, b = hip.v4_quantize(ints(rng, n, 7)), hip.v4_quantize(ints(rng, n, 7))
        d = hip.v4_dot(*a, *b, mode=DOT_EXACT)
        assert bits(d) == bits(oracle.v4_dot(*a, *b))
        assert abs(float(d) - float(oracle.v4_dot_scalar(*a, *b))) <= 0.02
        assert abs(float(hip.v4_dot(*a, *b, mode=DOT_FAST)) - float(d)) <= 0.02


# ---------------------------------------------------------------- matrix ops vs oracle
GRID = [(128, 128), (128, 384), (256, 256), (384, 128), (512, 1280), (1280, 640)]


@pytest.mark.parametrize("shape", GRID)
def test_matrix_quantize_bit_exact(hip, oracle, shape):
    M, N = shape
    rng = np.random.default_rng(M * 31 + N)
    for A in (ints(rng, M * N, 10).reshape(M, N), rng.normal(size=(M, N)).astype(np.float32)):
        A[0, 0] = -0.0
        q, s = hip.m4_quantize(A)
        qo, so = oracle.m4_quantize(A)
        assert same(q, qo) and same(s, so)


def test_matrix_quantize_zero_tile(hip, oracle):
    A = np.zeros((128, 256), np.float32)
    A[64:, 128:] = 5.0
    q, s = hip.m4_quantize(A)
    qo, so = oracle.m4_quantize(A)
    assert same(q, qo) and same(s, so) and s[0] == 1.0


@pytest.mark.parametrize("shape", GRID + [(128, 65536 + 128), (256, 131072)])
def test_mvm_bit_exact(hip, oracle, shape):
    M, N = shape
    rng = np.random.default_rng(M * 13 + N)
    qA, _ = random_packed(rng, M * N)
    sA = rng.uniform(0.5, 2.0, size=(M // 64) * (N // 64)).astype(np.float32)
    qx, sx = random_packed(rng, N)
    d = hip.m4_rowdots(qA, sA, M, N, qx, sx)
    assert same(d, oracle.m4_rowdots(qA, sA, M, N, qx, sx))
    r, sr = hip.m4_mvm(qA, sA, M, N, qx, sx)
    ro, sro = oracle.m4_mvm(qA, sA, M, N, qx, sx)
    assert same(r, ro) and same(sr, sro)


def test_mvm_on_quantized_floats(hip, oracle):
    # end to end like the reference's mvm test (03_matrix.cpp:248-326): quantize both on the device, multiply
    rng = np.random.default_rng(4)
    M, N = 384, 512
    A, x = ints(rng, M * N, 10).reshape(M, N), ints(rng, N, 10)
    qA, sA = hip.m4_quantize(A)
    qx = hip.v4_quantize(x)
    r, sr = hip.m4_mvm(qA, sA, M, N, *qx)
    ro, sro = oracle.m4_mvm(*oracle.m4_quantize(A), M, N, *oracle.v4_quantize(x))
    assert same(r, ro) and same(sr, sro)


def test_mvm_zero_matrix(hip, oracle):
    M, N = 128, 256
    qA, sA = np.zeros(M * N // 2, np.uint8), np.ones((M // 64) * (N // 64), np.float32)
    qx, sx = random_packed(np.random.default_rng(1), N)
    r, sr = hip.m4_mvm(qA, sA, M, N, qx, sx)
    assert not r.any() and sr.tolist() == [1.0, 1.0]


@pytest.mark.parametrize("shape", [(128, 128, 128), (128, 256, 384), (256, 128, 1024)])
def test_gemm_bit_exact(hip, oracle, shape):
    M, N, K = shape
    rng = np.random.default_rng(M + N + K)
    qA, _ = random_packed(rng, M * K)
    qB, _ = random_packed(rng, N * K)
    sA = rng.uniform(0.5, 2.0, size=(M // 64) * (K // 64)).astype(np.float32)
    sB = rng.uniform(0.5, 2.0, size=(N // 64) * (K // 64)).astype(np.float32)
    C = hip.m4_gemm(qA, sA, M, K, qB, sB, N)
    assert same(C, oracle.m4_gemm(qA, sA, M, K, qB, sB, N))


@pytest.mark.parametrize("shape", [(128, 128, 128), (256, 128, 512), (128, 384, 768)])
def test_gemm_integer_sums_exact(hip, oracle, shape):
    """SURVEY 8(a8) output (1): the exact int32 K-block sums, through the C ABI.  Ranges with even begin and count run on the MFMA
    kernel (accumulating across K-blocks inside the matrix pipe), the others on the VALU kernel; both must equal the oracle's
    per-block sums added up, and the whole range is the unscaled int4 x int4 -> int32 GEMM."""
    M, N, K = shape
    rng = np.random.default_rng(7 * M + N + K)
    qA, _ = random_packed(rng, M * K)
    qB, _ = random_packed(rng, N * K)
    S = oracle.m4_gemm_isums(qA, M, K, qB, N).astype(np.int64)            # [M][N][K / 64]
    kb = K // 64
    assert np.array_equal(hip.m4_gemm_i32(qA, M, K, qB, N), S.sum(2))
    for b0, cnt in [(0, 2), (kb - 2, 2), (0, 1), (kb - 1, 1), (1, kb - 1), (2, kb - 2) if kb > 2 else (0, 2)]:
        got = hip.m4_gemm_i32(qA, M, K, qB, N, b0, cnt)
        assert np.array_equal(got, S[:, :, b0:b0 + cnt].sum(2)), (b0, cnt)
    # all nibbles at +-7: the largest sums the format allows
    q7 = np.full(M * K // 2, 0x77, np.uint8)
    qm = np.full(N * K // 2, 0x99, np.uint8)
    assert (hip.m4_gemm_i32(q7, M, K, qm, N) == -49 * K).all()
    # the same sums from operands prepared once (either or both), whole range and an even sub-range
    for prepare in (("A",), ("B",), ("A", "B")):
        assert np.array_equal(hip.m4_gemm_i32_prepared(qA, M, K, qB, N, prepare=prepare), S.sum(2)), prepare
    assert np.array_equal(hip.m4_gemm_i32_prepared(qA, M, K, qB, N, kb - 2, 2), S[:, :, kb - 2:].sum(2))


@pytest.mark.parametrize("prepare", [("A",), ("B",), ("A", "B")])
def test_gemm_with_prepared_operands_equals_gemm(hip, oracle, prepare):
    M, N, K = 256, 384, 640
    rng = np.random.default_rng(len(prepare) * 11 + ord(prepare[0]))
    qA, _ = random_packed(rng, M * K)
    qB, _ = random_packed(rng, N * K)
    sA = rng.uniform(0.5, 2.0, size=(M // 64) * (K // 64)).astype(np.float32)
    sB = rng.uniform(0.5, 2.0, size=(N // 64) * (K // 64)).astype(np.float32)
    C = hip.m4_gemm_prepared(qA, sA, M, K, qB, sB, N, prepare=prepare)
    assert same(C, hip.m4_gemm(qA, sA, M, K, qB, sB, N)) and same(C, oracle.m4_gemm(qA, sA, M, K, qB, sB, N))


def test_gemm_calls_on_two_streams_do_not_share_scratch(hip, oracle):
    """clm4_gemm keeps its FP6 images in scratch that belongs to (device, stream): two different products enqueued back to back
    on two streams must both come out right (with one shared buffer the second re-code would overwrite the first one's operands)"""
    lib = hip.lib
    M = N = K = 1024
    rng = np.random.default_rng(99)
    data = []
    for _ in range(2):
        qA, _ = random_packed(rng, M * K)
        qB, _ = random_packed(rng, N * K)
        sA = rng.uniform(0.5, 2.0, size=(M // 64) * (K // 64)).astype(np.float32)
        sB = rng.uniform(0.5, 2.0, size=(N // 64) * (K // 64)).astype(np.float32)
        data.append((qA, sA, qB, sB))
    streams = [C.c_void_p(), C.c_void_p()]
    for st in streams:
        hip.check(lib.clv_stream_create(C.byref(st)))
    dev = [[hip.to_device(a) for a in d] for d in data]
    out = [hip.alloc(M * N * 4), hip.alloc(M * N * 4)]
    hip.sync()
    for rep in range(3):
        for k in (0, 1):
            b = dev[k]
            hip.check(lib.clm4_gemm(b[0].ptr, b[1].ptr, M, K, b[2].ptr, b[3].ptr, N, out[k].ptr, streams[k]))
    for st in streams:
        hip.check(lib.clv_stream_sync(st))
    for k in (0, 1):
        assert same(out[k].download(np.float32, M * N).reshape(M, N), hip.m4_gemm(*data[k][:2], M, K, *data[k][2:], N))
    for st in streams:
        hip.check(lib.clv_stream_destroy(st))


def test_gemm_wide_scale_range(hip, oracle):
    """scales over 30 decades: the per-block factor c_b goes through the fold unscaled, whatever its magnitude"""
    M, N, K = 256, 384, 512
    rng = np.random.default_rng(99)
    qA, _ = random_packed(rng, M * K)
    qB, _ = random_packed(rng, N * K)
    sA = (10.0 ** rng.uniform(-15, 15, size=(M // 64) * (K // 64))).astype(np.float32)
    sB = (10.0 ** rng.uniform(-15, 15, size=(N // 64) * (K // 64))).astype(np.float32)
    C = hip.m4_gemm(qA, sA, M, K, qB, sB, N)
    assert np.isfinite(C).all() and same(C, oracle.m4_gemm(qA, sA, M, K, qB, sB, N))


def test_gemm_extreme_nibbles(hip, oracle):
    """every value +-7: the largest block sums (64 * 49) of either sign"""
    M, N, K = 128, 128, 256
    rng = np.random.default_rng(5)
    sign = rng.integers(0, 2, size=(M * K // 2, 2))
    qA = np.where(sign[:, 0], 0x70, 0x90).astype(np.uint8) | np.where(sign[:, 1], 0x07, 0x09).astype(np.uint8)
    sign = rng.integers(0, 2, size=(N * K // 2, 2))
    qB = np.where(sign[:, 0], 0x70, 0x90).astype(np.uint8) | np.where(sign[:, 1], 0x07, 0x09).astype(np.uint8)
    qB[: K // 2] = 0x77                                                 # row 0 of B all +7 ...
    qA[: K // 2] = 0x77                                                 # ... against row 0 of A all +7: S_b = 3136
    qA[K // 2: K] = 0x99                                                # row 1 of A all -7: S_b = -3136
    sA = np.ones((M // 64) * (K // 64), np.float32)
    sB = np.ones((N // 64) * (K // 64), np.float32)
    C = hip.m4_gemm(qA, sA, M, K, qB, sB, N).reshape(M, N)
    assert same(C, oracle.m4_gemm(qA, sA, M, K, qB, sB, N).reshape(M, N))
    assert C[0, 0] > 0 and C[1, 0] == -C[0, 0]


def test_gemm_int8_kernel_bit_exact():
    """the int8-MFMA kernel (gemm4.hip, CLV_GEMM_KERNEL=i8) is kept for A/B runs: same bits.  The switch is read once per
    process, hence the child process."""
    import subprocess
    import sys
    code = (
        "import numpy as np\n"
        "from clover_amd.lib_binding import CloverHip\n"
        "from oracle.binding import Oracle\n"
        "hip, o = CloverHip(), Oracle()\n"
        "rng = np.random.default_rng(3)\n"
        "M, N, K = 256, 128, 384\n"
        "qA = rng.integers(0, 256, M * K // 2).astype(np.uint8); qB = rng.integers(0, 256, N * K // 2).astype(np.uint8)\n"
        "qA[(qA >> 4) == 8] ^= 0x10; qA[(qA & 15) == 8] ^= 0x01; qB[(qB >> 4) == 8] ^= 0x10; qB[(qB & 15) == 8] ^= 0x01\n"
        "sA = rng.uniform(0.5, 2, (M // 64) * (K // 64)).astype(np.float32); sB = rng.uniform(0.5, 2, (N // 64) * (K // 64)).astype(np.float32)\n"
        "C = hip.m4_gemm(qA, sA, M, K, qB, sB, N); Co = o.m4_gemm(qA, sA, M, K, qB, sB, N)\n"
        "assert C.tobytes() == Co.tobytes()\n"
        "print('ok')\n")
    import os
    env = dict(os.environ, CLV_GEMM_KERNEL="i8")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr[-2000:]


@pytest.mark.parametrize("shape", [(128, 128), (256, 384), (1024, 640)])
def test_matrix_restore_exact(hip, oracle, shape):
    """CloverMatrix4::restore_scalar (CloverMatrix4.h:266-301)"""
    M, N = shape
    rng = np.random.default_rng(M + 3 * N)
    qA, _ = random_packed(rng, M * N)
    sA = rng.uniform(0.5, 2.0, size=(M // 64) * (N // 64)).astype(np.float32)
    A = hip.m4_restore(qA, sA, M, N)
    Ao = oracle.m4_restore(qA, sA, M, N)
    assert same(A, Ao)
    for (i, j) in ((0, 0), (63, 64), (64, 63), (M - 1, N - 1), (100, N - 127)):      # and the element accessor agrees (get: :123-139)
        assert A[i, j] == oracle.m4_get(qA, sA, M, N, i, j)


# ---------------------------------------------------------------- stochastic rounding: same XORShift stream
@pytest.mark.parametrize("n", [128, 1024, 8192 + 128, (1 << 17) + 384])
def test_stochastic_vector_quantize_same_stream(hip, oracle, n):
    rng = np.random.default_rng(n)
    x = (rng.normal(size=n) * 2).astype(np.float32)
    st = hip.new_rng(12345, 67890)
    o = oracle.rng(12345, 67890)
    for _ in range(2):                       # second call continues the stream
        q, s = hip.v4_quantize(x, rng=st)
        qo, so = oracle.v4_quantize(x, o)
        assert same(q, qo) and same(s, so)
    k1, k2 = hip.rng_get(st)
    o1, o2 = oracle.rng_keys(o)
    assert np.array_equal(k1, o1) and np.array_equal(k2, o2)


@pytest.mark.parametrize("segments", [1, 4, 16, 64])
def test_stochastic_vector_ops_every_kernel_shape(hip, oracle, segments):
    """The size-picked kernel shape (segments per wave) must not change a bit: force each one on sizes that span several
    workgroups of that shape (32 * segments blocks each) with a ragged tail, for quantize and scaleAndAdd, two calls each."""
    n = 64 * (32 * segments * 5 + 7 * segments + 3)
    n += (-n) % 128
    rng = np.random.default_rng(segments)
    x = (rng.normal(size=n) * 3).astype(np.float32)
    x[64 * 3:64 * 4] = 0.0                                   # an all-zero block: scale 1.0
    x[64 * 5:64 * 6] = np.float32(1e-39)                     # a block whose maximum makes 7 / max overflow: every nibble 0 (cvttps overflow)
    x[64 * 9:64 * 9 + 5] = 0.0
    x[n - 64:] = -0.0
    (qu, su), (qv, sv) = random_packed(rng, n), random_packed(rng, n)
    assert hip.lib.clv_rng_set_segments(segments) == 0
    try:
        st, o = hip.new_rng(77, 88), oracle.rng(77, 88)
        for _ in range(2):
            q, s = hip.v4_quantize(x, rng=st)
            qo, so = oracle.v4_quantize(x, o)
            assert same(q, qo) and same(s, so)
            r, sr = hip.v4_scale_and_add(qu, su, qv, sv, 0.75, rng=st)
            ro, sro = oracle.v4_scale_and_add(qu, su, qv, sv, 0.75, o)
            assert same(r, ro) and same(sr, sro)
        r, sr = hip.v4_scale_and_add(qu, su, qv, sv, -0.5, rng=st, in_place=True)      # x += a*v, the IHT update
        ro, sro = oracle.v4_scale_and_add(qu, su, qv, sv, -0.5, o)
        assert same(r, ro) and same(sr, sro)
        k1, k2 = hip.rng_get(st)
        o1, o2 = oracle.rng_keys(o)
        assert np.array_equal(k1, o1) and np.array_equal(k2, o2)
    finally:
        hip.lib.clv_rng_set_segments(0)


@pytest.mark.parametrize("shape", [(128, 128), (256, 384), (1024, 640), (192 * 2, 1280)])
def test_stochastic_matrix_quantize_and_mvm_same_stream(hip, oracle, shape):
    M, N = shape
    rng = np.random.default_rng(8 + M + N)
    A = rng.normal(size=(M, N)).astype(np.float32)
    st, o = hip.new_rng(445560390295639063, 2935984234003016713), oracle.rng(445560390295639063, 2935984234003016713)
    for _ in range(2):                       # the second call continues the stream
        qA, sA = hip.m4_quantize(A, rng=st)
        qAo, sAo = oracle.m4_quantize(A, o)
        assert same(qA, qAo) and same(sA, sAo)
    qx = oracle.v4_quantize(rng.normal(size=N).astype(np.float32))
    r, sr = hip.m4_mvm(qA, sA, M, N, *qx, rng=st)
    ro, sro = oracle.m4_mvm(qA, sA, M, N, *qx, o)
    assert same(r, ro) and same(sr, sro)
    assert np.array_equal(hip.rng_get(st)[1], oracle.rng_keys(o)[1])


# ---------------------------------------------------------------- error behaviour of the boundary
def test_bad_sizes_are_rejected(hip):
    from clover_amd.lib_binding import CloverHipError
    with pytest.raises(CloverHipError):
        hip.v4_quantize(np.zeros(100, np.float32))
    buf = hip.alloc(1024)
    assert hip.lib.clm4_mvm(buf.ptr, buf.ptr, 160, 128, buf.ptr, buf.ptr, buf.ptr, buf.ptr, None, None) == -1      # not a whole number of 64-row blocks
    assert hip.lib.clm4_mvm(buf.ptr, buf.ptr, 128, 192, buf.ptr, buf.ptr, buf.ptr, buf.ptr, None, None) == -1      # cols: multiples of 128 only


def test_gemm_prepared_operand_misuse_is_rejected(hip):
    """prepared operands carry their shape: a call that disagrees with it, or an odd K-block range of the integer GEMM without
    the nibbles, comes back as CLV_ERR_INVALID with a message, not as a wrong result"""
    import ctypes as C
    lib = hip.lib
    M, N, K = 256, 128, 256
    a, b, c = hip.alloc(M * K // 2), hip.alloc(N * K // 2), hip.alloc(M * N * 4)
    sa, sb = hip.alloc((M // 64) * (K // 64) * 4), hip.alloc((N // 64) * (K // 64) * 4)
    opA, opB = C.c_void_p(), C.c_void_p()
    hip.check(lib.clm4_gemm_prepare(a.ptr, M, K, C.byref(opA), None))
    hip.check(lib.clm4_gemm_prepare(b.ptr, N, K, C.byref(opB), None))
    try:
        assert lib.clm4_gemm_prepared(opA, None, sa.ptr, M + 128, K, opB, None, sb.ptr, N, c.ptr, None) == -1          # A was prepared as 256 rows
        assert b"prepared as" in lib.clv_last_error()
        assert lib.clm4_gemm_prepared(opA, None, sa.ptr, M, K, opA, None, sb.ptr, N, c.ptr, None) == -1                # "B" has M rows, not N
        assert lib.clm4_gemm_i32_prepared(opA, None, M, K, opB, None, N, 1, 1, c.ptr, None) == -1                      # odd range: needs the nibbles
        assert b"nibbles" in lib.clv_last_error()
        assert lib.clm4_gemm_i32_prepared(opA, a.ptr, M, K, opB, b.ptr, N, 1, 1, c.ptr, None) == 0                     # ... and runs with them
        assert lib.clm4_gemm_i32_prepared(opA, None, M, K, opB, None, N, 0, K // 64 + 2, c.ptr, None) == -1            # range beyond K
        assert lib.clm4_gemm_prepared(None, None, sa.ptr, M, K, opB, None, sb.ptr, N, c.ptr, None) == -1               # neither image nor nibbles for A
        hip.sync()
    finally:
        hip.check(lib.clm4_gemm_release(opA))
        hip.check(lib.clm4_gemm_release(opB))


def test_degenerate_sizes(hip):
    """n_pad == 0 / rows == 0 are accepted as no-ops (the reference's containers always hold >= 128 elements)"""
    buf = hip.alloc(256)
    lib = hip.lib
    assert lib.clv4_quantize(buf.ptr, 0, buf.ptr, buf.ptr, None, None) == 0
    assert lib.clv4_restore(buf.ptr, buf.ptr, 0, buf.ptr, None) == 0
    assert lib.clv4_scale_and_add(buf.ptr, buf.ptr, buf.ptr, buf.ptr, 1.0, 0, buf.ptr, buf.ptr, None, None) == 0
    assert lib.clm4_mvm(buf.ptr, buf.ptr, 0, 128, buf.ptr, buf.ptr, buf.ptr, buf.ptr, None, None) == 0
    assert lib.clm4_quantize(buf.ptr, 0, 0, buf.ptr, buf.ptr, None, None) == 0
    assert lib.clm4_transpose(buf.ptr, buf.ptr, 0, 0, buf.offset(128), buf.ptr, None) == 0
    out = hip.alloc(4)
    hip.check(hip.lib.clv_memset(out.ptr, 0xFF, 4, None))
    assert lib.clv4_dot(buf.ptr, buf.ptr, buf.ptr, buf.ptr, 0, 0, out.ptr, None, None) == 0
    assert out.download(np.float32, 1)[0] == 0.0
    hip.sync()


@pytest.mark.parametrize("tile", ["128", "256"])
def test_gemm_both_workgroup_tiles_bit_exact(tile):
    """clm4_gemm picks the 256 x 256 persistent kernel for large products and the 128 x 128 one otherwise; CLV_GEMM_TILE forces
    one (read once per process, hence the child).  Shapes that are multiples of 128 but not of 256 (partial tiles: waves outside
    the matrix compute on zero rows and must not store), odd stage counts (the three stage buffers must end where they began,
    tile after tile), more tiles than a grid of one workgroup per CU would take in one go, the int32 sums and prepared operands."""
    import os
    import subprocess
    import sys
    code = (
        "import numpy as np\n"
        "from clover_amd.lib_binding import CloverHip\n"
        "from oracle.binding import Oracle\n"
        "hip, o = CloverHip(), Oracle()\n"
        "rng = np.random.default_rng(31)\n"
        "def nib(n):\n"
        "    q = rng.integers(0, 256, n).astype(np.uint8); q[(q >> 4) == 8] ^= 0x10; q[(q & 15) == 8] ^= 0x01; return q\n"
        "for (M, N, K) in ((128, 128, 128), (384, 640, 384), (256, 256, 128), (896, 128, 640), (640, 1152, 256)):\n"
        "    qA, qB = nib(M * K // 2), nib(N * K // 2)\n"
        "    sA = rng.uniform(0.5, 2, (M // 64) * (K // 64)).astype(np.float32); sB = rng.uniform(0.5, 2, (N // 64) * (K // 64)).astype(np.float32)\n"
        "    C = hip.m4_gemm(qA, sA, M, K, qB, sB, N); Co = o.m4_gemm(qA, sA, M, K, qB, sB, N)\n"
        "    assert C.tobytes() == Co.tobytes(), (M, N, K)\n"
        "    S = o.m4_gemm_isums(qA, M, K, qB, N).astype(np.int64)\n"
        "    assert np.array_equal(hip.m4_gemm_i32(qA, M, K, qB, N), S.sum(2)), (M, N, K)\n"
        "    if K >= 256: assert np.array_equal(hip.m4_gemm_i32(qA, M, K, qB, N, 2, K // 64 - 2), S[:, :, 2:].sum(2)), (M, N, K)\n"
        "    assert hip.m4_gemm_prepared(qA, sA, M, K, qB, sB, N, prepare=('B',)).tobytes() == Co.tobytes(), (M, N, K)\n"
        "# many tiles per workgroup of the persistent grid: 4352 x 4352 = 17 x 17 tiles of 256 (289 > 256 CUs), sampled against the definition\n"
        "G, K = 4352, 256\n"
        "qA, qB = nib(G * K // 2), nib(G * K // 2)\n"
        "sA = rng.uniform(0.5, 2, (G // 64) * (K // 64)).astype(np.float32); sB = rng.uniform(0.5, 2, (G // 64) * (K // 64)).astype(np.float32)\n"
        "C = hip.m4_gemm(qA, sA, G, K, qB, sB, G)\n"
        "rows = [0, 63, 255, 256, 2047, 4095, 4096, 4351]\n"
        "sub = np.concatenate([qA[r * K // 2:(r + 1) * K // 2] for r in rows])\n"
        "ssub = np.concatenate([sA[(r >> 6) * (K // 64):((r >> 6) + 1) * (K // 64)] for r in rows])\n"
        "for i, r in enumerate(rows):\n"
        "    for c in (0, 255, 256, 4100, 4351):\n"
        "        S = o.v4_word_isums(qA[r * K // 2:(r + 1) * K // 2], qB[c * K // 2:(c + 1) * K // 2]).reshape(K // 64, 8).sum(1)\n"
        "        acc = np.float32(0)\n"
        "        for b in range(K // 64):\n"
        "            cb = np.float32(np.float32(sA[(r >> 6) * (K // 64) + b] * np.float32(1.0 / 49.0)) * sB[(c >> 6) * (K // 64) + b])\n"
        "            acc = np.float32(np.float64(cb) * np.float64(S[b]) + np.float64(acc))\n"
        "        assert acc.view(np.uint32) == C[r, c].view(np.uint32), (r, c)\n"
        "print('ok')\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, CLV_GEMM_TILE=tile)
    out = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "ok" in out.stdout, (out.stdout[-500:], out.stderr[-3000:])


def test_dot_fast_one_launch_gives_the_two_launch_bits(hip):
    """round 5: clv4_dot FAST is one launch (slots + collector workgroup); the round-1..4 form (k_v4_dot_partial + k_v4_dot_final), kept behind
    CLV_DOT_FAST_TWO_LAUNCHES (read once per process: a child process), must give the same bits -- same per-thread order, same final tree"""
    import json
    import os
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    code = (
        "import sys, json, numpy as np\n"
        f"sys.path.insert(0, {str(root)!r})\n"
        "from clover_amd.lib_binding import CloverHip, DOT_FAST\n"
        "hip = CloverHip(device=0); lib = hip.lib; out = hip.alloc(8); res = {}\n"
        "for n in (128, 8192, (1 << 16) + 384, 1 << 20, (1 << 24) + 128, 1 << 26):\n"
        "    q, s, q2, s2 = hip.alloc(n // 2), hip.alloc(n // 16), hip.alloc(n // 2), hip.alloc(n // 16)\n"
        "    hip.check(lib.clv_fill_random_nibbles(q.ptr, q.nbytes, 1, 0, None)); hip.check(lib.clv_fill_random_nibbles(q2.ptr, q2.nbytes, 2, 0, None))\n"
        "    hip.check(lib.clv_fill_random_scales(s.ptr, s.nbytes // 4, 3, 0, None)); hip.check(lib.clv_fill_random_scales(s2.ptr, s2.nbytes // 4, 4, 0, None))\n"
        "    for _ in range(3):\n"
        "        hip.check(lib.clv4_dot(q.ptr, s.ptr, q2.ptr, s2.ptr, n, DOT_FAST, out.ptr, None, None))\n"
        "    res[str(n)] = int(out.download(np.uint32)[0])\n"
        "print(json.dumps(res))\n")
    outs = []
    for extra in ({}, {"CLV_DOT_FAST_TWO_LAUNCHES": "1"}):
        env = dict(os.environ, **extra)
        env.pop("CLV_DOT_FAST_TWO_LAUNCHES", None) if not extra else None
        p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
        assert p.returncode == 0, p.stderr[-1500:]
        outs.append(json.loads(p.stdout.strip().splitlines()[-1]))
    assert outs[0] == outs[1], outs


def test_internal_workspaces_belong_to_their_stream(hip):
    """dot (FAST: the hand-over slots of the one-launch reduction; EXACT: the chain operands), threshold (the multi-kernel large-vector path
    and the REFERENCE heap walk) keep their scratch per (device, stream): the same calls enqueued alternately on two streams, different data
    on each, nothing synchronised in between, give what each gives alone on the default stream."""
    import ctypes as C

    from clover_amd.lib_binding import DOT_EXACT, DOT_FAST, THRESHOLD_FAST, THRESHOLD_REFERENCE
    lib = hip.lib
    rng = np.random.default_rng(4242)
    n_dot, n_thr, n_ref = 1 << 22, (1 << 20) + 128, 65536
    sets = []
    for _ in range(2):
        sets.append({"u": random_packed(rng, n_dot), "v": random_packed(rng, n_dot), "t": random_packed(rng, n_thr), "r": random_packed(rng, n_ref)})
    want = []
    for d in sets:                                                   # alone, default stream
        want.append({"fast": hip.v4_dot(*d["u"], *d["v"], mode=DOT_FAST), "exact": hip.v4_dot(*d["u"], *d["v"], mode=DOT_EXACT),
                     "thr": hip.v4_threshold(*d["t"], n_thr - 77, n_thr // 4, mode=THRESHOLD_FAST),
                     "ref": hip.v4_threshold(*d["r"], n_ref, n_ref // 8, mode=THRESHOLD_REFERENCE)})
    streams = [C.c_void_p(), C.c_void_p()]
    for st in streams:
        hip.check(lib.clv_stream_create(C.byref(st)))
    try:
        dev = [{k: [hip.to_device(a) for a in v] for k, v in d.items()} for d in sets]
        out = [{"fast": hip.alloc(4), "exact": hip.alloc(4)} for _ in sets]
        hip.sync()
        for rep in range(3):
            for k in (0, 1):
                b, st = dev[k], streams[k]
                hip.check(lib.clv4_dot(b["u"][0].ptr, b["u"][1].ptr, b["v"][0].ptr, b["v"][1].ptr, n_dot, DOT_FAST, out[k]["fast"].ptr, None, st))
            for k in (0, 1):
                b, st = dev[k], streams[k]
                hip.check(lib.clv4_dot(b["u"][0].ptr, b["u"][1].ptr, b["v"][0].ptr, b["v"][1].ptr, n_dot, DOT_EXACT, out[k]["exact"].ptr, None, st))
            if rep == 0:                                             # threshold works in place: once
                for k in (0, 1):
                    b, st = dev[k], streams[k]
                    hip.check(lib.clv4_threshold_mode(b["t"][0].ptr, b["t"][1].ptr, n_thr - 77, n_thr, n_thr // 4, THRESHOLD_FAST, None, st))
                for k in (0, 1):
                    b, st = dev[k], streams[k]
                    hip.check(lib.clv4_threshold_mode(b["r"][0].ptr, b["r"][1].ptr, n_ref, n_ref, n_ref // 8, THRESHOLD_REFERENCE, None, st))
        for st in streams:
            hip.check(lib.clv_stream_sync(st))
        for k in (0, 1):
            assert np.float32(out[k]["fast"].download(np.float32, 1)[0]).tobytes() == np.float32(want[k]["fast"]).tobytes()
            assert np.float32(out[k]["exact"].download(np.float32, 1)[0]).tobytes() == np.float32(want[k]["exact"]).tobytes()
            assert same(dev[k]["t"][0].download(np.uint8, n_thr // 2), want[k]["thr"])
            assert same(dev[k]["r"][0].download(np.uint8, n_ref // 2), want[k]["ref"])
    finally:
        for st in streams:
            hip.check(lib.clv_stream_destroy(st))


def test_abi_from_several_host_threads(hip):
    """four host threads, a stream each, the same mix of calls at once (ctypes drops the GIL inside a call): the registries behind the ABI --
    workspaces and hand-over slots per (device, stream), the error string per thread -- serve them without mixing anything up"""
    import ctypes as C
    import threading

    from clover_amd.lib_binding import DOT_EXACT, DOT_FAST
    lib = hip.lib
    n = 1 << 20
    rng = np.random.default_rng(7)
    data = [(random_packed(rng, n), random_packed(rng, n), rng.normal(size=n).astype(np.float32)) for _ in range(4)]
    want = [(hip.v4_dot(*u, *v, mode=DOT_FAST), hip.v4_dot(*u, *v, mode=DOT_EXACT), hip.v4_quantize(x)) for u, v, x in data]
    dev = [([hip.to_device(a) for a in u], [hip.to_device(a) for a in v], hip.to_device(x), hip.alloc(n // 2), hip.alloc(n // 16), hip.alloc(4), hip.alloc(4))
           for u, v, x in data]
    hip.sync()
    errors = []

    def work(k):
        try:
            st = C.c_void_p()
            hip.check(lib.clv_set_device(0))
            hip.check(lib.clv_stream_create(C.byref(st)))
            u, v, x, q, s, of, oe = dev[k]
            for _ in range(40):
                hip.check(lib.clv4_dot(u[0].ptr, u[1].ptr, v[0].ptr, v[1].ptr, n, DOT_FAST, of.ptr, None, st))
                hip.check(lib.clv4_quantize(x.ptr, n, q.ptr, s.ptr, None, st))
                hip.check(lib.clv4_dot(u[0].ptr, u[1].ptr, v[0].ptr, v[1].ptr, n, DOT_EXACT, oe.ptr, None, st))
            assert lib.clv4_dot(None, None, None, None, n, DOT_FAST, None, None, st) != 0        # an error of this thread's own
            hip.check(lib.clv_stream_sync(st))
            hip.check(lib.clv_stream_destroy(st))
        except Exception as e:                                       # noqa: BLE001
            errors.append(f"thread {k}: {type(e).__name__}: {e}")

    threads = [threading.Thread(target=work, args=(k,)) for k in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(120)
    assert not errors and not any(t.is_alive() for t in threads), errors
    hip.sync()
    for k in range(4):
        _, _, _, q, s, of, oe = dev[k]
        assert np.float32(of.download(np.float32, 1)[0]).tobytes() == np.float32(want[k][0]).tobytes()
        assert np.float32(oe.download(np.float32, 1)[0]).tobytes() == np.float32(want[k][1]).tobytes()
        assert same(q.download(np.uint8, n // 2), want[k][2][0]) and same(s.download(np.float32, n // 64), want[k][2][1])
