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
@pytest.mark.parametrize("segments", [0, 1, 4, 16, 64])
def test_gpu_v8_quantize_stochastic_same_stream(hip, oracle, segments):
    n = 64 * (32 * max(segments, 1) * 5 + 7 * max(segments, 1) + 3)
    n += (-n) % 128
    rng = np.random.default_rng(n)
    x = (rng.normal(size=n) * 2).astype(np.float32)
    assert hip.lib.clv_rng_set_segments(segments) == 0
    try:
        st, o = hip.new_rng(21, 43), oracle.rng(21, 43)
        for _ in range(2):
            q, s = hip.v8_quantize(x, rng=st)
            qo, so = oracle.v8_quantize(x, o)
            assert same(q, qo) and same(s, so)
        assert np.array_equal(hip.rng_get(st)[1], oracle.rng_keys(o)[1])
    finally:
        hip.lib.clv_rng_set_segments(0)


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


# ---------------------------------------------------------------- the 8-bit vector steps of the mixed IHT / GD loops
def _rand_v8(rng, n):
    q = rng.integers(-127, 128, n).astype(np.int8)
    s = rng.uniform(0.5, 2, n // 64).astype(np.float32)
    return q, s


def test_oracle_v8_scale_and_add_definition(oracle):
    rng = np.random.default_rng(2)
    n = 512
    (qu, su), (qv, sv) = _rand_v8(rng, n), _rand_v8(rng, n)
    a = np.float32(-0.625)
    r, sr = oracle.v8_scale_and_add(qu, su, qv, sv, float(a))
    want = oracle.v8_restore(qu, su).astype(np.float64) + float(a) * oracle.v8_restore(qv, sv).astype(np.float64)
    got = oracle.v8_restore(r, sr)
    assert np.all(np.abs(got - want) <= np.repeat(sr, 64) / 127.0 + 1e-5)
    assert np.allclose(sr, np.abs(want.reshape(-1, 64)).max(axis=1), rtol=1e-5)


def _threshold8_lowest_index(q, s, n, k):
    mags = np.abs((q.astype(np.float32) * np.repeat(s, 64)) / np.float32(127.0))[:n]
    out = q.copy()
    if k < n:
        tau = np.sort(mags)[::-1][k - 1] if k > 0 else np.inf
        keep = mags > tau
        ties = np.flatnonzero(mags == tau)[: max(k - int(keep.sum()), 0)]
        keep[ties] = True
        out[:n] = out[:n] * keep
    return out


@pytest.mark.parametrize("case", [(128, 128, 17), (1000, 1024, 100), (4096, 4096, 1024)])
def test_oracle_v8_threshold_keeps_the_k_largest(oracle, case):
    n, npad, k = case
    rng = np.random.default_rng(n)
    q, s = _rand_v8(rng, npad)
    out = oracle.v8_threshold(q, s, n, k)
    mags = np.abs((q.astype(np.float32) * np.repeat(s, 64)) / np.float32(127.0))
    kept = (out[:n] != 0) | ((q[:n] == 0) & False)
    ref = _threshold8_lowest_index(q, s, n, k)
    assert np.array_equal(out[n:], q[n:])
    assert np.array_equal(np.sort(mags[:n][kept]), np.sort(mags[:n][ref[:n] != 0]))     # same surviving multiset of magnitudes


@pytest.mark.gpu
@pytest.mark.parametrize("n", [128, 4096, 8192 + 384, (1 << 18) + 128, (1 << 22) + 256])
def test_gpu_v8_scale_and_add_exact(hip, oracle, n):
    rng = np.random.default_rng(n + 1)
    (qu, su), (qv, sv) = _rand_v8(rng, n), _rand_v8(rng, n)
    qu[:64] = 0
    qv[:64] = 0
    for a, in_place in ((-1.0, False), (0.001, True)):
        r, sr = hip.v8_scale_and_add(qu, su, qv, sv, a, in_place=in_place)
        ro, sro = oracle.v8_scale_and_add(qu, su, qv, sv, a)
        assert same(r, ro) and same(sr, sro)
    assert sr[0] == 1.0


def _saa8_cases():
    """(qu, su, qv, sv, a, in_place) for the block-kernel test: ragged last chunks, one chunk, scales at both ends of the fp32 range"""
    out = []
    for n in (128, 64 * 64, 64 * 65 + 64, (1 << 18) + 128 * 37):
        rng = np.random.default_rng(900 + n)
        (qu, su), (qv, sv) = _rand_v8(rng, n), _rand_v8(rng, n)
        idx = rng.integers(0, n // 64, 40)
        su[idx[:20]] = np.float32(1e-38) * rng.uniform(0.1, 9, 20).astype(np.float32)
        sv[idx[:20]] = np.float32(1e-39)
        su[idx[20:]] = np.float32(1e37)
        sv[idx[30:]] = np.float32(3e37)
        out += [(qu, su, qv, sv, 0.5, False), (qu, su, qv, sv, -2.0, True)]
    return out


@pytest.mark.gpu
def test_gpu_v8_scale_and_add_block_kernel(hip, oracle):
    """the once-per-block kernel of large vectors (k_v8_scale_and_add_blk, taken on its own from n = 2^27: tests/test_gpu_large.py) forced on
    small ones in a child process (CLV_SAA8_BLK_MIN_BLOCKS is read once per process): ragged last chunks, a single partial chunk, scales at
    both ends of the fp32 range (127 / max overflowing -> the block's bytes are 0, as CloverVector8.h:1262-1290 leaves them), in place and
    out of place -- equal to the oracle wherever the oracle's scale is finite, and to the plain kernel everywhere"""
    import hashlib
    import json
    import os
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    code = (
        "import sys, json, hashlib\n"
        f"sys.path.insert(0, {str(root)!r}); sys.path.insert(0, {str(root / 'tests')!r})\n"
        "import test_mixed8 as T\n"
        "from clover_amd.lib_binding import CloverHip\n"
        "hip = CloverHip(device=0); res = []\n"
        "for qu, su, qv, sv, a, ip in T._saa8_cases():\n"
        "    r, sr = hip.v8_scale_and_add(qu, su, qv, sv, a, in_place=ip)\n"
        "    res.append(hashlib.sha256(r.tobytes() + sr.tobytes()).hexdigest())\n"
        "print(json.dumps(res))\n")
    p = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, CLV_SAA8_BLK_MIN_BLOCKS="1"), capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-1500:]
    forced = json.loads(p.stdout.strip().splitlines()[-1])
    for (qu, su, qv, sv, a, ip), h in zip(_saa8_cases(), forced):
        r, sr = hip.v8_scale_and_add(qu, su, qv, sv, a, in_place=ip)                       # this process: the plain kernel at these sizes
        assert hashlib.sha256(r.tobytes() + sr.tobytes()).hexdigest() == h
        ro, sro = oracle.v8_scale_and_add(qu, su, qv, sv, a)
        ok = np.isfinite(sro)
        assert same(sr[ok], sro[ok]) and same(r.reshape(-1, 64)[ok], ro.reshape(-1, 64)[ok])


@pytest.mark.gpu
@pytest.mark.parametrize("segments", [1, 4, 16, 64])
def test_gpu_v8_scale_and_add_stochastic_every_kernel_shape(hip, oracle, segments):
    n = 64 * (32 * segments * 5 + 7 * segments + 3)
    n += (-n) % 128
    rng = np.random.default_rng(segments + 40)
    (qu, su), (qv, sv) = _rand_v8(rng, n), _rand_v8(rng, n)
    assert hip.lib.clv_rng_set_segments(segments) == 0
    try:
        st, o = hip.new_rng(7, 8), oracle.rng(7, 8)
        for in_place in (False, True):
            r, sr = hip.v8_scale_and_add(qu, su, qv, sv, 0.75, rng=st, in_place=in_place)
            ro, sro = oracle.v8_scale_and_add(qu, su, qv, sv, 0.75, o)
            assert same(r, ro) and same(sr, sro)
        assert np.array_equal(hip.rng_get(st)[1], oracle.rng_keys(o)[1])
    finally:
        hip.lib.clv_rng_set_segments(0)


@pytest.mark.gpu
@pytest.mark.parametrize("case", [(128, 128, 64), (1000, 1024, 64), (2047, 2048, 300), (8192, 8192, 2048), (32768 - 3, 32768, 5000),
                                  (32768 + 128, 32768 + 128, 999), ((1 << 20) + 77, (1 << 20) + 128, 262144), (512, 512, 0), (512, 512, 511),
                                  ((1 << 18) + 128, (1 << 18) + 128, 0)])
def test_gpu_v8_threshold_top_k(hip, oracle, case):
    n, npad, k = case
    rng = np.random.default_rng(n + k)
    x = np.zeros(npad, np.float32)
    x[:n] = rng.integers(-40, 41, size=n)                    # the reference's test data (02_vector.cpp:460): many ties
    q, s = oracle.v8_quantize(x)
    out = hip.v8_threshold(q, s, n, k)
    assert same(out, _threshold8_lowest_index(q, s, n, k))   # the whole output under the lowest-index tie rule
    ref = oracle.v8_threshold(q, s, n, k)                     # and the reference's heap keeps the same multiset of magnitudes
    mags = np.abs((q.astype(np.float32) * np.repeat(s, 64)) / np.float32(127.0))[:n]
    assert np.array_equal(np.sort(mags[out[:n] != 0]), np.sort(mags[ref[:n] != 0]))
    assert same(hip.v8_threshold(out, s, n, k), out)          # idempotent
    if n <= 32768 + 128:                                       # REFERENCE mode: the reference's survivor set, byte for byte (CloverVector8.h:1680-1740)
        from clover_amd.lib_binding import THRESHOLD_REFERENCE
        assert same(hip.v8_threshold(q, s, n, k, mode=THRESHOLD_REFERENCE), ref)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(128, 128), (256, 384), (1024, 32768 + 128)])
@pytest.mark.parametrize("stochastic", [False, True])
def test_gpu_fused_mixed_mvm_scale_and_add_equals_the_two_calls(hip, oracle, shape, stochastic):
    M, N = shape
    rng = np.random.default_rng(M + N + stochastic)
    qA, sA, x = _inputs(rng, M, N)
    qx, sx = oracle.v8_quantize(x)
    qu, su = _rand_v8(rng, M)
    a = -0.37
    st, o = (hip.new_rng(31, 41), oracle.rng(31, 41)) if stochastic else (None, None)
    lib = hip.lib
    d = [hip.to_device(v) for v in (qA, sA, qx, sx)]
    for want_t, in_place in ((True, False), (False, False), (True, True)):
        du, dsu = hip.to_device(qu), hip.to_device(su)
        dt, dst = (hip.alloc(M), hip.alloc(M // 16)) if want_t else (None, None)
        dr, dsr = (du, dsu) if in_place else (hip.alloc(M), hip.alloc(M // 16))
        hip.check(lib.clm4_mvm_v8_scale_and_add(d[0].ptr, d[1].ptr, M, N, d[2].ptr, d[3].ptr, du.ptr, dsu.ptr, a,
                                                dt.ptr if dt else None, dst.ptr if dst else None, dr.ptr, dsr.ptr,
                                                st.ptr if st else None, None))
        to, sto = oracle.m4_mvm_v8(qA, sA, M, N, qx, sx, o)
        ro, sro = oracle.v8_scale_and_add(qu, su, to, sto, a, o)
        if want_t:
            assert same(dt.download(np.int8, M), to) and same(dst.download(np.float32, M // 64), sto)
        assert same(dr.download(np.int8, M), ro) and same(dsr.download(np.float32, M // 64), sro)
    if stochastic:
        assert np.array_equal(hip.rng_get(st)[1], oracle.rng_keys(o)[1])


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["iht", "gd", "iht_stochastic"])
def test_gpu_mixed_iht_loop_matches_oracle_loop(hip, oracle, mode):
    """clm4_iht_v8 = the reference's published 4-bit IHT / GD configuration (CloverMatrix4 + CloverVector8, 02_bit04.cpp:140)"""
    rng = np.random.default_rng(12)
    m, n, K, iters = 256, 512, 64, 4
    Phi = oracle.m4_quantize(rng.uniform(-1, 1, size=(m, n)).astype(np.float32))
    PhiT = oracle.m4_transpose(*Phi, m, n)
    y = oracle.v8_quantize((rng.normal(size=m) * 3).astype(np.float32))
    thr = mode != "gd"
    st, o = (hip.new_rng(3, 5), oracle.rng(3, 5)) if mode == "iht_stochastic" else (None, None)
    x = (np.zeros(n, np.int8), np.ones(n // 64, np.float32))
    for _ in range(iters):
        t1 = oracle.m4_mvm_v8(*Phi, m, n, *x, o)
        t2 = oracle.v8_scale_and_add(*y, *t1, -1.0, o)
        t3 = oracle.m4_mvm_v8(*PhiT, n, m, *t2, o)
        x = oracle.v8_scale_and_add(*x, *t3, 0.01, o)
        if thr:
            x = (_threshold8_lowest_index(x[0], x[1], n, K), x[1])
    d = [hip.to_device(v) for v in (*Phi, *PhiT, *y)]
    bufs = [hip.alloc(k) for k in (n, n // 16, m, m // 16, m, m // 16, n, n // 16)]     # x, sx, t1, st1, t2, st2, t3, st3
    hip.check(hip.lib.clv_memset(bufs[0].ptr, 0x55, n, None))                           # clm4_iht_v8 must clear x itself
    hip.check(hip.lib.clm4_iht_v8(d[0].ptr, d[1].ptr, d[2].ptr, d[3].ptr, m, n, bufs[0].ptr, bufs[1].ptr, n, d[4].ptr, d[5].ptr,
                                  bufs[2].ptr, bufs[3].ptr, bufs[4].ptr, bufs[5].ptr, bufs[6].ptr, bufs[7].ptr, iters, K, 0.01,
                                  1 if thr else 0, st.ptr if st else None, None))
    assert same(bufs[0].download(np.int8, n), x[0]) and same(bufs[1].download(np.float32, n // 64), x[1])
    assert same(bufs[6].download(np.int8, n), t3[0]) and same(bufs[4].download(np.int8, m), t2[0])
    if st:
        assert np.array_equal(hip.rng_get(st)[1], oracle.rng_keys(o)[1])


# ---------------------------------------------------------------- CloverVector8::dot (round 5)
def _v8_pair(rng, n, kind):
    """two quantized 8-bit vectors; kind: normal data, integers (the reference's dot test data, 02_vector.cpp:258-295), extremes (+-127
    everywhere: the largest block integers, 64 x 127^2), zero blocks"""
    if kind == "ints":
        x, y = rng.integers(-10, 11, size=n).astype(np.float32), rng.integers(-10, 11, size=n).astype(np.float32)
    else:
        x, y = (rng.normal(size=n) * 3).astype(np.float32), (rng.normal(size=n) * 0.01).astype(np.float32)
    if kind == "extreme":
        x = np.where(rng.random(n) < 0.5, -5.0, 5.0).astype(np.float32)
        y = np.where(rng.random(n) < 0.5, -0.25, 0.25).astype(np.float32)
    if kind == "zeros":
        x[64:192] = 0.0
        y[n - 64:] = 0.0
    return x, y


@pytest.mark.parametrize("n", [128, 1024, 64 * 33 + 64])
def test_oracle_v8_dot_orders_agree(oracle, n):
    """the reference's own check for dot is SIMD vs scalar within 0.02 (test/validate/02_vector.cpp); the README-style known answer
    (a = 1, b = 2 -> 2 n) holds in the 8-bit format too"""
    rng = np.random.default_rng(n)
    x, y = _v8_pair(rng, n, "ints")
    qx, sx = oracle.v8_quantize(x)
    qy, sy = oracle.v8_quantize(y)
    d, ds, d64 = oracle.v8_dot(qx, sx, qy, sy), oracle.v8_dot_scalar(qx, sx, qy, sy), oracle.v8_dot_f64(qx, sx, qy, sy)
    assert abs(float(d) - float(ds)) <= 0.02 * max(1.0, abs(d64) * 1e-3) and abs(float(d) - d64) <= 1e-5 * max(abs(d64), 1.0) * (n // 64)
    a, b = oracle.v8_quantize(np.ones(n, np.float32)), oracle.v8_quantize(2 * np.ones(n, np.float32))
    assert float(oracle.v8_dot(*a, *b)) == 2.0 * n


@pytest.mark.gpu
@pytest.mark.parametrize("n", [128, 256, 1024 + 128, 64 * 16 * 7, (1 << 16) + 128, (1 << 20) + 384])
@pytest.mark.parametrize("kind", ["normal", "ints", "extreme", "zeros"])
def test_gpu_v8_dot_exact_and_fast(hip, oracle, n, kind):
    """clv8_dot EXACT == the reference's order (8 fma chains over all blocks + the hadd tree), bit for bit; FAST within the fast order's bound"""
    from clover_amd.lib_binding import DOT_EXACT, DOT_FAST
    rng = np.random.default_rng(n + len(kind))
    x, y = _v8_pair(rng, n, kind)
    qx, sx = oracle.v8_quantize(x)
    qy, sy = oracle.v8_quantize(y)
    want = oracle.v8_dot(qx, sx, qy, sy)
    got = hip.v8_dot(qx, sx, qy, sy, mode=DOT_EXACT)
    assert np.float32(got).tobytes() == np.float32(want).tobytes(), (got, want)
    d64 = oracle.v8_dot_f64(qx, sx, qy, sy)
    terms = np.abs(np.repeat(sx * sy, 64).astype(np.float64) * qx.astype(np.float64) * qy.astype(np.float64)).sum() / (127.0 * 127.0)
    fast = hip.v8_dot(qx, sx, qy, sy, mode=DOT_FAST)
    assert abs(float(fast) - d64) <= 2e-6 * terms + 1e-6
    assert abs(float(got) - float(oracle.v8_dot_scalar(qx, sx, qy, sy))) <= 0.02 + 1e-5 * abs(d64)


@pytest.mark.gpu
def test_gpu_v8_dot_empty_and_bad_arguments(hip):
    lib = hip.lib
    out, buf = hip.alloc(8), hip.alloc(1024)
    assert lib.clv8_dot(buf.ptr, buf.ptr, buf.ptr, buf.ptr, 0, 0, out.ptr, None, None) == 0
    assert out.download(np.float32, 1)[0] == 0.0
    assert lib.clv8_dot(buf.ptr, buf.ptr, buf.ptr, buf.ptr, 192, 0, out.ptr, None, None) != 0            # not a multiple of 128
    assert lib.clv8_dot(buf.ptr, buf.ptr, buf.ptr, buf.ptr, 128, 7, out.ptr, None, None) != 0            # unknown mode
    assert lib.clv8_dot(None, buf.ptr, buf.ptr, buf.ptr, 128, 0, out.ptr, None, None) != 0
