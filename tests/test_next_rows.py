"""SURVEY 8(f) "next" rows: scaleAndAdd (f1), transpose (f2), threshold (f3).

CPU part: the oracle's restatements against independent formulations.  GPU part (-m gpu): HIP vs oracle,
bit-exact for scaleAndAdd and transpose (the reference's own tests are exact there, 02_vector.cpp:341-447,
03_matrix.cpp:153-246); threshold in FAST mode compares the surviving multiset of magnitudes (the reference's test is a
10 % tolerance on sorted magnitudes, 02_vector.cpp:449-498; tie-breaking among equal magnitudes is heap-order
dependent in the reference and lowest-index-first in FAST mode), and in REFERENCE mode the whole output nibble for nibble
against the oracle's walk of the reference's min-heap (CloverVector4.h:1927-1972, CloverBase.h:208-249)."""
import subprocess
from pathlib import Path

import numpy as np
import pytest

from conftest import bits, random_packed
from clover_amd.lib_binding import THRESHOLD_REFERENCE

ROOT = Path(__file__).resolve().parent.parent


def nibbles(b):
    hi = (b.astype(np.int8) >> 4).astype(np.int32)
    lo = ((b << 4).astype(np.int8) >> 4).astype(np.int32)
    return np.stack([hi, lo], 1).reshape(-1)


def same(a, b):
    return np.array_equal(np.asarray(a).view(np.uint8), np.asarray(b).view(np.uint8))


# ------------------------------------------------------------------------------------------- CPU (oracle)
def test_oracle_scale_and_add_is_axpy_then_quantize(oracle):
    rng = np.random.default_rng(0)
    for n in (128, 640):
        (qu, su), (qv, sv) = random_packed(rng, n), random_packed(rng, n)
        a = np.float32(0.5)
        r, sr = oracle.v4_scale_and_add(qu, su, qv, sv, float(a))
        # independent numpy evaluation of CloverVector4.h:1226-1318: du = q*su7 ; val = fma(qv, sv7, du)
        su7 = (su / np.float32(7.0)).astype(np.float32)
        sv7 = ((sv * a).astype(np.float32) / np.float32(7.0)).astype(np.float32)
        du = (nibbles(qu).astype(np.float32) * np.repeat(su7, 64)).astype(np.float32)
        val = (nibbles(qv).astype(np.float64) * np.repeat(sv7, 64).astype(np.float64) + du.astype(np.float64)).astype(np.float32)
        q2, s2 = oracle.v4_quantize(val)
        assert same(r, q2) and same(sr, s2)
        # and it approximates u + a v within one quantisation step of the result
        u, v = oracle.v4_restore(qu, su), oracle.v4_restore(qv, sv)
        assert np.all(np.abs(oracle.v4_restore(r, sr) - (u + a * v)) <= np.repeat(sr, 64) / 7 + 1e-5)


def test_oracle_transpose(oracle):
    rng = np.random.default_rng(1)
    M, N = 128, 384
    q, _ = random_packed(rng, M * N)
    s = rng.uniform(0.5, 2, size=(M // 64) * (N // 64)).astype(np.float32)
    qt, st = oracle.m4_transpose(q, s, M, N)
    assert np.array_equal(nibbles(qt).reshape(N, M), nibbles(q).reshape(M, N).T)
    assert np.array_equal(st.reshape(N // 64, M // 64), s.reshape(M // 64, N // 64).T)
    q2, s2 = oracle.m4_transpose(qt, st, N, M)
    assert same(q2, q) and same(s2, s)                       # involution
    for (i, j) in [(0, 0), (5, 383), (127, 64), (64, 65)]:
        assert bits(oracle.m4_get(q, s, M, N, i, j)) == bits(oracle.m4_get(qt, st, N, M, j, i))


def test_oracle_threshold_matches_std_make_heap(oracle, tmp_path):
    exe = tmp_path / "thr"
    subprocess.run(["g++", "-O1", "-std=c++11", str(ROOT / "tests" / "cpp" / "threshold_stdheap.cpp"), "-o", str(exe)], check=True)
    rng = np.random.default_rng(2)
    for (n, npad, k) in ((128, 128, 64), (1000, 1024, 64), (777, 896, 5), (2047, 2048, 300)):
        q, s = random_packed(rng, npad)
        mags = np.abs(oracle.v4_restore(q, s))[:n]
        inp = f"{n} {k}\n" + "\n".join(repr(float(m)) for m in mags) + "\n"
        kept_ref = [int(x) for x in subprocess.run([str(exe)], input=inp, capture_output=True, text=True, check=True).stdout.split()]
        out = oracle.v4_threshold(q, s, n, k)
        before, after = nibbles(q), nibbles(out)
        expect = np.zeros_like(before)
        expect[kept_ref] = before[kept_ref]
        expect[n:] = before[n:]                              # padding untouched
        assert np.array_equal(after, expect)


# ------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.parametrize("n", [128, 256, 2048 + 128, 1 << 16, (1 << 18) - 128, 1 << 18, (1 << 18) + 128, (1 << 18) + 64 * 63 + 64, (1 << 20) + 128])
def test_gpu_scale_and_add_bit_exact(hip, oracle, n):
    rng = np.random.default_rng(n)
    (qu, su), (qv, sv) = random_packed(rng, n), random_packed(rng, n)
    qu[:32] = 0                                              # a block that sums to zero -> scale 1.0 path
    qv[:32] = 0
    for a in (0.5, -1.0, 1.7):
        r, sr = hip.v4_scale_and_add(qu, su, qv, sv, a)
        ro, sro = oracle.v4_scale_and_add(qu, su, qv, sv, a)
        assert same(r, ro) and same(sr, sro)
    r, sr = hip.v4_scale_and_add(qu, su, qv, sv, 0.5, in_place=True)       # x.scaleAndAdd(t, mu) of the IHT loop
    assert same(r, oracle.v4_scale_and_add(qu, su, qv, sv, 0.5)[0])
    assert sr[0] == 1.0


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1 << 16, 1 << 18, (1 << 19) + 640])
def test_gpu_scale_and_add_extreme_scales(hip, oracle, n):
    """both deterministic kernels (the plain one below 2^18 elements, the block-scalar one from there on) at the ends of the fp32 range: block
    scales near FLT_MAX (the 16-fold scale of the nibble -> q/16 conversion would overflow: the wave takes the plain conversion), denormal
    scales, zero blocks on either side, and scales so small that 7 / max overflows (every nibble of the block becomes 0, as in the reference)"""
    rng = np.random.default_rng(n + 1)
    (qu, su), (qv, sv) = random_packed(rng, n), random_packed(rng, n)
    nb = n // 64
    su, sv = su.copy(), sv.copy()
    su[0::7] = np.float32(2.0e38)
    sv[3::11] = np.float32(1.5e38)
    su[1::13] = np.float32(1e-42)                          # denormal
    sv[1::13] = np.float32(3e-43)
    su[5::17] = np.float32(1e-39)
    sv[5::17] = np.float32(1e-39)
    qu[32 * 2:32 * 3] = 0
    qv[32 * 4:32 * 5] = 0
    qu[32 * 6:32 * 7] = 0
    qv[32 * 6:32 * 7] = 0
    for a in (1.0, -0.25, 1e-3):
        with np.errstate(over="ignore", invalid="ignore"):
            ro, sro = oracle.v4_scale_and_add(qu, su, qv, sv, a)
        r, sr = hip.v4_scale_and_add(qu, su, qv, sv, a)
        ok = np.isfinite(sro)                              # a block whose maximum overflowed to inf is outside the reference's contract
        assert ok.sum() > nb // 2
        assert same(sr[ok], sro[ok]) and same(r.reshape(nb, 32)[ok], ro.reshape(nb, 32)[ok])


@pytest.mark.gpu
def test_gpu_scale_and_add_stochastic_same_stream(hip, oracle):
    rng = np.random.default_rng(3)
    n = 8192 + 384
    (qu, su), (qv, sv) = random_packed(rng, n), random_packed(rng, n)
    st, o = hip.new_rng(11, 22), oracle.rng(11, 22)
    for _ in range(2):
        r, sr = hip.v4_scale_and_add(qu, su, qv, sv, -1.0, rng=st)
        ro, sro = oracle.v4_scale_and_add(qu, su, qv, sv, -1.0, o)
        assert same(r, ro) and same(sr, sro)
    assert np.array_equal(hip.rng_get(st)[1], oracle.rng_keys(o)[1])


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(128, 128), (128, 384), (640, 256), (1024, 2048), (2048, 4096), (4096, 2048), (2304, 2048)])
def test_gpu_transpose_exact(hip, oracle, shape):
    # from 2048 x 2048 (8 x 8 tiles of 256) on, the kernel walks the tiles in per-XCD 8 x 8 blocks; 2304 rows: plain order again
    M, N = shape
    rng = np.random.default_rng(M + N)
    q, _ = random_packed(rng, M * N)
    s = rng.uniform(0.5, 2, size=(M // 64) * (N // 64)).astype(np.float32)
    qt, st = hip.m4_transpose(q, s, M, N)
    qo, so = oracle.m4_transpose(q, s, M, N)
    assert same(qt, qo) and same(st, so)
    q2, s2 = hip.m4_transpose(qt, st, N, M)
    assert same(q2, q) and same(s2, s)


@pytest.mark.gpu
@pytest.mark.parametrize("case", [(128, 128, 64), (1000, 1024, 64), (2047, 2048, 300), (65536, 65536, 16384),
                                  ((1 << 20) + 77, (1 << 20) + 128, 262144), (512, 512, 0), (512, 512, 511),
                                  (131072 - 5, 131072, 40000), (100000, 100096, 1), (131072 + 128, 131072 + 128, 999)])
def test_gpu_threshold_top_k(hip, oracle, case):
    n, npad, k = case
    rng = np.random.default_rng(n + k)
    x = np.zeros(npad, np.float32)
    x[:n] = rng.integers(-40, 41, size=n)                    # the reference's test data (02_vector.cpp:460)
    q, s = oracle.v4_quantize(x)
    out = hip.v4_threshold(q, s, n, k)
    before, after = nibbles(q), nibbles(out)
    assert np.array_equal(after[n:], before[n:])              # padding untouched
    kept = after[:n] != 0
    assert np.all((after[:n] == before[:n]) | (after[:n] == 0))   # survivors keep their value
    mags = np.abs(oracle.v4_restore(q, s))[:n]
    ref = oracle.v4_threshold(q, s, n, k)
    kept_ref = nibbles(ref)[:n] != 0
    # identical surviving multiset of magnitudes (zeros carry no information: a kept 0 nibble stays 0)
    assert np.array_equal(np.sort(mags[kept]), np.sort(mags[kept_ref]))
    if 0 < k < n:
        tau = np.sort(mags)[::-1][k - 1]
        assert np.all(mags[~kept] <= tau) and np.all(mags[kept] >= tau)
        assert kept.sum() <= k
        # ties: lowest indices survive
        tie_idx = np.flatnonzero(mags == tau)
        n_keep = k - int((mags > tau).sum())
        if tau > 0:
            assert np.array_equal(np.flatnonzero(kept & (mags == tau)), tie_idx[:n_keep])
    # FAST mode's own tie rule (lowest index), restated on the CPU: documents WHICH ties FAST keeps -- not a reference comparison
    assert same(out, _threshold_lowest_index(oracle, q, s, n, k))
    again = hip.v4_threshold(out, s, n, k)                    # idempotent
    assert same(again, out)
    # REFERENCE mode: the reference's survivor set, nibble for nibble, against the oracle's heap walk (CloverVector4.h:1927-1972)
    if n <= 131072 + 128:
        assert same(hip.v4_threshold(q, s, n, k, mode=THRESHOLD_REFERENCE), ref)


@pytest.mark.gpu
@pytest.mark.parametrize("case", [(128, 128, 1), (128, 128, 127), (130, 256, 64), (1000, 1024, 250), (8192, 8192, 1024), (8192, 8192, 2048),
                                  (16384, 16384, 2048), (65536, 65536, 8192), (131072, 131072, 16384), (40000, 40064, 16385),
                                  (65536, 65536, 40000), (5000, 5120, 4999), (777, 896, 2), (4096, 4096, 4095), (30000, 30080, 20000), (30000, 30080, 20001)])
@pytest.mark.parametrize("data", ["ints40", "ties", "distinct"])
def test_gpu_threshold_reference_order(hip, oracle, case, data):
    """CLV_THRESHOLD_REFERENCE: index-identical to the reference's min-heap walk (make_heap over the first k, strict > against the
    root, min_heapify with left-first ties) -- the heap whole in LDS for k <= 20000, its top 14 levels in LDS and the rest in global memory beyond.  `ties`: three magnitudes only
    (almost everything ties at tau); `distinct`: per-block scales make nearly all magnitudes different."""
    n, npad, k = case
    rng = np.random.default_rng(7 * n + k + len(data))
    x = np.zeros(npad, np.float32)
    if data == "ints40":
        x[:n] = rng.integers(-40, 41, size=n)
    elif data == "ties":
        x[:n] = rng.choice(np.array([-7, -3, 0, 3, 7], np.float32), size=n)
    else:
        x[:n] = rng.normal(size=n) * np.repeat(rng.uniform(0.1, 10, size=npad // 64), 64)[:n]
    q, s = oracle.v4_quantize(x)
    ref = oracle.v4_threshold(q, s, n, k)
    out = hip.v4_threshold(q, s, n, k, mode=THRESHOLD_REFERENCE)
    assert same(out, ref)
    assert int((nibbles(out)[:n] != 0).sum()) <= k and np.array_equal(nibbles(out)[n:], nibbles(q)[n:])


@pytest.mark.gpu
@pytest.mark.parametrize("case", [(4096, 512), (65536, 20000)])
def test_gpu_threshold_reference_order_with_nan_and_inf_scales(hip, oracle, case):
    """gt_idx_t (CloverBase.h:216-218) is (a.value > b.value) || isnan(a.value): with a NaN block scale (NaN magnitudes) or an infinite one
    (inf, and NaN over a zero nibble) among the first k elements the initial make_heap takes other turns than a plain `>` -- outside the
    reference's data contract, but the walk is the reference's for those inputs too.  LDS heap and global-memory heap."""
    n, k = case
    rng = np.random.default_rng(n + k)
    q, s = random_packed(rng, n)
    s = s.copy()
    s[1] = np.float32(np.nan)
    s[3] = np.float32(np.inf)
    s[(k // 64) + 2] = np.float32(np.nan)               # one beyond the first k elements: never enters (NaN > root is false)
    with np.errstate(invalid="ignore"):
        ref = oracle.v4_threshold(q, s, n, k)
        out = hip.v4_threshold(q, s, n, k, mode=THRESHOLD_REFERENCE)
    assert same(out, ref)


@pytest.mark.gpu
@pytest.mark.parametrize("k", [0, 1, 100])
def test_gpu_threshold_reference_order_leaves_the_padding_alone(hip, oracle, k):
    """raw ABI input whose nibbles beyond n are NOT zero (a container never produces that): both modes touch the first n elements only,
    also for k = 0 (everything cleared) -- n = 1003 ends in the middle of a word"""
    n, npad = 1003, 1024
    rng = np.random.default_rng(99 + k)
    q, s = random_packed(rng, npad)
    ref = oracle.v4_threshold(q, s, n, k)
    for mode in (THRESHOLD_REFERENCE, 0):
        out = hip.v4_threshold(q, s, n, k, mode=mode)
        assert np.array_equal(nibbles(out)[n:], nibbles(q)[n:]) and int((nibbles(out)[:n] != 0).sum()) <= k
        if mode == THRESHOLD_REFERENCE:
            assert same(out, ref)


def _threshold_lowest_index(oracle, q, s, n, k):
    """the HIP tie rule on the CPU: everything above the K-th magnitude, then the first ties by index"""
    mags = np.abs(oracle.v4_restore(q, s))[:n]
    out = nibbles(q).copy()
    if k < n:
        tau = np.sort(mags)[::-1][k - 1] if k > 0 else np.inf
        keep = mags > tau
        ties = np.flatnonzero(mags == tau)[: max(k - int(keep.sum()), 0)]
        keep[ties] = True
        out[:n] *= keep
    return (((out[0::2] & 0xF) << 4) | (out[1::2] & 0xF)).astype(np.uint8)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["iht_stream", "iht_plain", "gd_stream", "iht_reference"])
def test_gpu_iht_loop_matches_oracle_loop(hip, oracle, mode):
    """clm4_iht = Q_IHT / Q_GD of the reference (01_measure.h:923-946, 999-1021) on the device (own stream or the
    default stream), against the same step sequence evaluated with the oracle."""
    import ctypes as C
    rng = np.random.default_rng(42)
    m, n, K, iters, mu = 256, 512, 48, 6, np.float32(0.002)
    qPhi, _ = random_packed(rng, m * n)
    sPhi = rng.uniform(0.5, 2, size=(m // 64) * (n // 64)).astype(np.float32)
    qT, sT = oracle.m4_transpose(qPhi, sPhi, m, n)
    y = random_packed(rng, m)
    d = {k_: hip.to_device(v) for k_, v in dict(Phi=qPhi, sPhi=sPhi, PhiT=qT, sPhiT=sT, y=y[0], sy=y[1]).items()}
    bufs = {k_: hip.alloc(max(sz, 4)) for k_, sz in dict(x=n // 2, sx=n // 16, t1=m // 2, st1=m // 16, t2=m // 2, st2=m // 16,
                                                           t3=n // 2, st3=n // 16).items()}
    stream = C.c_void_p()
    if mode != "iht_plain":
        hip.check(hip.lib.clv_stream_create(C.byref(stream)))
    thr = 0 if mode.startswith("gd") else (2 if mode == "iht_reference" else 1)      # 2: threshold in the reference's survivor order
    hip.check(hip.lib.clm4_iht(d["Phi"].ptr, d["sPhi"].ptr, d["PhiT"].ptr, d["sPhiT"].ptr, m, n, bufs["x"].ptr, bufs["sx"].ptr, n,
                               d["y"].ptr, d["sy"].ptr, bufs["t1"].ptr, bufs["st1"].ptr, bufs["t2"].ptr, bufs["st2"].ptr,
                               bufs["t3"].ptr, bufs["st3"].ptr, iters, K, float(mu), thr, None, stream))
    hip.check(hip.lib.clv_stream_sync(stream))
    xq, xs = bufs["x"].download(np.uint8, n // 2), bufs["sx"].download(np.float32, n // 64)
    # oracle loop
    x = (np.zeros(n // 2, np.uint8), np.ones(n // 64, np.float32))
    for _ in range(iters):
        t1 = oracle.m4_mvm(qPhi, sPhi, m, n, *x)
        t2 = oracle.v4_scale_and_add(*y, *t1, -1.0)
        t3 = oracle.m4_mvm(qT, sT, n, m, *t2)
        x = oracle.v4_scale_and_add(*x, *t3, float(mu))
        if thr == 2:                                    # the oracle's heap walk = the reference's threshold: the reference's trajectory
            x = (oracle.v4_threshold(x[0], x[1], n, K), x[1])
        elif thr:
            x = (_threshold_lowest_index(oracle, x[0], x[1], n, K), x[1])
    assert same(xq, x[0]) and same(xs, x[1])
    if stream:
        hip.check(hip.lib.clv_stream_destroy(stream))


def test_oracle_mixed_mvm_is_dequantized_product(oracle):
    rng = np.random.default_rng(4)
    M, N = 128, 384
    qA, _ = random_packed(rng, M * N)
    sA = rng.uniform(0.5, 2, size=(M // 64) * (N // 64)).astype(np.float32)
    x = rng.normal(size=N).astype(np.float32)
    r = oracle.m4_mvm_f32(qA, sA, M, N, x)
    Ad = (nibbles(qA).reshape(M, N).astype(np.float64) * np.repeat(np.repeat(sA.reshape(M // 64, N // 64), 64, 0), 64, 1) / 7.0)
    assert np.allclose(r, Ad @ x.astype(np.float64), rtol=0, atol=2e-4)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(128, 128), (256, 640), (384, 16384 + 128), (128, 65536)])
def test_gpu_mixed_mvm_f32_bit_exact(hip, oracle, shape):
    # the reference's own test compares with a double-accumulated scalar loop at 0.01 (03_matrix.cpp:419-491);
    # here the 32-chain order is reproduced, so the fp32 results are identical
    M, N = shape
    rng = np.random.default_rng(M + N)
    qA, _ = random_packed(rng, M * N)
    sA = rng.uniform(0.5, 2, size=(M // 64) * (N // 64)).astype(np.float32)
    x = rng.normal(size=N).astype(np.float32)
    assert same(hip.m4_mvm_f32(qA, sA, M, N, x), oracle.m4_mvm_f32(qA, sA, M, N, x))


@pytest.mark.gpu
def test_gpu_mixed_mvm_f32_tiny_and_huge_scales(hip, oracle):
    """scales at both ends of the fp32 range: the kernel's (q / 16) * (16 c) form must step aside where 16 c would overflow (a whole
    16384-column chunk then takes the plain form) and is exact for tiny and denormal c; the result must not change either way"""
    M, N = 128, 16384 + 256
    rng = np.random.default_rng(77)
    qA, _ = random_packed(rng, M * N)
    sA = rng.uniform(0.5, 2, size=(M // 64) * (N // 64)).astype(np.float32)
    sA[3] = np.float32(1e-37)            # f32(s / 7) / 16 is a denormal: first chunk, first row group
    sA[(N // 64) + 5] = np.float32(3e-38)
    sA[N // 64 - 1] = np.float32(1e37)   # second chunk (plain scales around it)
    x = rng.normal(size=N).astype(np.float32)
    x[200:260] *= np.float32(1e30)       # the tiny blocks' products are visible in the sum
    sA[9] = np.float32(3e38)             # f32(s / 7) * 16 overflows: first chunk, first row group takes the plain form
    sA[2 * (N // 64) - 2] = np.float32(2.9e38)      # second chunk, second row group
    x[9 * 64:10 * 64] *= np.float32(1e-30)
    x[N - 128:N - 64] *= np.float32(1e-30)
    sA[11] = np.float32(1e-42)           # a denormal scale: the fast form is exact for it
    out = hip.m4_mvm_f32(qA, sA, M, N, x)
    assert np.isfinite(out).all()
    assert same(out, oracle.m4_mvm_f32(qA, sA, M, N, x))


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(128, 128), (256, 384), (1024, 65536 + 128)])
@pytest.mark.parametrize("stochastic", [False, True])
def test_gpu_fused_mvm_scale_and_add_equals_the_two_calls(hip, oracle, shape, stochastic):
    """clm4_mvm_scale_and_add == mvm then scaleAndAdd: same t, same r, same XORShift positions; with and without storing
    t, and with the in-place result (x += a * (A v))."""
    M, N = shape
    rng = np.random.default_rng(M + N + stochastic)
    qA, _ = random_packed(rng, M * N)
    sA = rng.uniform(0.5, 2, size=(M // 64) * (N // 64)).astype(np.float32)
    (qx, sx), (qu, su) = random_packed(rng, N), random_packed(rng, M)
    a = -0.37
    st, o = (hip.new_rng(31, 41), oracle.rng(31, 41)) if stochastic else (None, None)
    for want_t, in_place in ((True, False), (False, False), (True, True)):
        t, s_t, r, sr = hip.m4_mvm_scale_and_add(qA, sA, M, N, qx, sx, qu, su, a, rng=st, in_place=in_place, want_t=want_t)
        to, sto = oracle.m4_mvm(qA, sA, M, N, qx, sx, o)
        ro, sro = oracle.v4_scale_and_add(qu, su, to, sto, a, o)
        if want_t:
            assert same(t, to) and same(s_t, sto)
        assert same(r, ro) and same(sr, sro)
    if stochastic:
        k1, k2 = hip.rng_get(st)
        o1, o2 = oracle.rng_keys(o)
        assert np.array_equal(k1, o1) and np.array_equal(k2, o2)


@pytest.mark.gpu
def test_gpu_fused_mvm_scale_and_add_rejects_aliasing_the_input(hip):
    buf = hip.alloc(4096)
    assert hip.lib.clm4_mvm_scale_and_add(buf.ptr, buf.ptr, 128, 128, buf.ptr, buf.offset(64), buf.offset(128), buf.offset(256), 1.0,
                                          None, None, buf.ptr, buf.offset(512), None, None) == -1


@pytest.mark.gpu
@pytest.mark.parametrize("case", [(4096, 4096, 1000), (131072 + 256, 131072 + 256, 30000), ((1 << 21) + 64 * 5 + 17, (1 << 21) + 512, 400000)])
def test_gpu_threshold_with_raw_nibbles_including_minus_8(hip, oracle, case):
    """raw ABI input with ALL sixteen nibble values (the quantiser never produces -8, a caller's buffer may): |-8| = 8 is its own magnitude
    class in the per-block tables (bit-sliced counts, r4) and in the one-workgroup kernel; the surviving multiset equals the oracle's, the
    survivors follow the lowest-index rule, and REFERENCE mode equals the oracle's heap walk where it applies"""
    n, npad, k = case
    rng = np.random.default_rng(n + k)
    q = rng.integers(0, 256, size=npad // 2, dtype=np.uint8)
    s = rng.uniform(0.5, 2, size=npad // 64).astype(np.float32)
    out = hip.v4_threshold(q, s, n, k)
    mags = np.abs(oracle.v4_restore(q, s))[:n]
    ref = oracle.v4_threshold(q, s, n, k)
    kept, kept_ref = nibbles(out)[:n] != 0, nibbles(ref)[:n] != 0
    assert np.array_equal(np.sort(mags[kept]), np.sort(mags[kept_ref]))
    assert same(out, _threshold_lowest_index(oracle, q, s, n, k))
    if n <= 131072 + 256:
        assert same(hip.v4_threshold(q, s, n, k, mode=THRESHOLD_REFERENCE), ref)
