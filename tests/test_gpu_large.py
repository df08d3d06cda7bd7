"""GPU parity at BASELINE.json's full sizes, through size-independent properties and sampled oracle checks
(the scalar oracle cannot finish a 65536 x 65536 mvm in seconds; the AVX2 port and sampling can).

  C2  quantize + dot, n = 2^24         : bit-exact vs the AVX2 restatement (itself == scalar oracle)
  C3  mvm 65536 x 65536                : sampled 64-row output blocks vs the scalar oracle; shard == whole
  C4  gemm 8192^3 (and 2048^3)         : sampled elements of the full result vs the oracle's definition
  C5  mvm 2^20 x 2^16, row-sharded     : the whole 32 GiB matrix on ONE GPU: the 8 shards of 131072 x 65536 an 8-GPU node would
                                         hold, run one after the other, == the unsharded call byte for byte; sampled 64-row blocks of
                                         every shard vs the scalar oracle
"""
import ctypes as C

import numpy as np
import pytest

from conftest import bits

pytestmark = pytest.mark.gpu


def same(a, b):
    return np.array_equal(np.asarray(a).view(np.uint8), np.asarray(b).view(np.uint8))


def test_c2_quantize_dot_2p24(hip, fast_oracle):
    from clover_amd.lib_binding import DOT_EXACT, DOT_FAST
    n = 1 << 24
    rng = np.random.default_rng(24)
    x = rng.integers(-10, 11, size=n).astype(np.float32)
    y = rng.integers(-10, 11, size=n).astype(np.float32)
    qx, sx = hip.v4_quantize(x)
    qy, sy = hip.v4_quantize(y)
    fx, fsx = fast_oracle.v4_quantize(x)
    assert same(qx, fx) and same(sx, fsx)
    d_exact = hip.v4_dot(qx, sx, qy, sy, mode=DOT_EXACT)
    assert bits(d_exact) == bits(fast_oracle.v4_dot(qx, sx, qy, sy))            # reference order, 131072 steps per chain
    d_fast = hip.v4_dot(qx, sx, qy, sy, mode=DOT_FAST)
    # only the fp32 summation order differs: relative 1e-5 (the two CPU orders of the reference differ by 372 ulp here)
    assert abs(float(d_fast) - float(d_exact)) <= 1e-5 * abs(float(d_exact)) + 1e-3
    # encode -> decode round trip: |x - restore(q)| <= scale / 7 per block (truncation), sign preserved
    xr = hip.v4_restore(qx, sx)
    assert np.all(np.abs(x - xr) <= np.repeat(sx, 64) / np.float32(7.0) + 1e-6)
    assert np.all(np.sign(xr) * np.sign(x) >= 0)


def _device_matrix(hip, rows, cols, seed):
    A = hip.alloc(rows * cols // 2)
    sA = hip.alloc((rows // 64) * (cols // 64) * 4)
    x = hip.alloc(cols // 2)
    sx = hip.alloc(cols // 64 * 4)
    hip.check(hip.lib.clv_fill_random_nibbles(A.ptr, A.nbytes, seed, 0, None))
    hip.check(hip.lib.clv_fill_random_scales(sA.ptr, sA.nbytes // 4, seed + 1, 0, None))
    hip.check(hip.lib.clv_fill_random_nibbles(x.ptr, x.nbytes, seed + 2, 0, None))
    hip.check(hip.lib.clv_fill_random_scales(sx.ptr, sx.nbytes // 4, seed + 3, 0, None))
    return A, sA, x, sx


def _download(hip, ptr, nbytes, dtype):
    out = np.empty(nbytes // np.dtype(dtype).itemsize, dtype=dtype)
    hip.check(hip.lib.clv_memcpy_d2h(out.ctypes.data, ptr, nbytes, None))
    return out


def test_c3_mvm_65536_sampled_and_sharded(hip, oracle):
    rows = cols = 65536
    hb = cols // 64
    A, sA, x, sx = _device_matrix(hip, rows, cols, 0xC3)
    r = hip.alloc(rows // 2)
    sr = hip.alloc(rows // 64 * 4)
    hip.check(hip.lib.clm4_mvm(A.ptr, sA.ptr, rows, cols, x.ptr, sx.ptr, r.ptr, sr.ptr, None, None))
    r_h, sr_h = r.download(np.uint8), sr.download(np.float32)
    # the synthetic fill really is nibbles in [-7,7] and scales in [0.5,2)
    qx, sxh = x.download(np.uint8), sx.download(np.float32)
    assert ((qx >> 4) != 8).all() and ((qx & 0xF) != 8).all() and sxh.min() >= 0.5 and sxh.max() < 2.0
    # sampled 64-row output blocks against the scalar oracle
    for g in (0, 1, 511, 777, 1023):
        blk = _download(hip, A.ptr + g * 64 * cols // 2, 64 * cols // 2, np.uint8)
        sblk = _download(hip, sA.ptr + g * hb * 4, hb * 4, np.float32)
        d = oracle.m4_rowdots(blk, sblk, 64, cols, qx, sxh)
        ro, sro = oracle.v4_quantize(np.concatenate([d, np.zeros(64, np.float32)]))
        assert same(r_h[g * 32:(g + 1) * 32], ro[:32]) and bits(sr_h[g]) == bits(sro[0])
    # C5 property: 8 contiguous row shards (as 8 GPUs would hold them) reproduce the whole result byte for byte
    shard = rows // 8
    parts_r, parts_s = [], []
    for k in range(8):
        rk, srk = hip.alloc(shard // 2), hip.alloc(shard // 64 * 4)
        hip.check(hip.lib.clm4_mvm(A.ptr + k * shard * cols // 2, sA.ptr + k * (shard // 64) * hb * 4, shard, cols,
                                   x.ptr, sx.ptr, rk.ptr, srk.ptr, None, None))
        parts_r.append(rk.download(np.uint8))
        parts_s.append(srk.download(np.float32))
    assert same(np.concatenate(parts_r), r_h) and same(np.concatenate(parts_s), sr_h)


def test_c5_whole_matrix_and_its_eight_shards_on_one_gpu(hip, oracle):
    """BASELINE config 5 (mvm 2^20 x 2^16, 8 row shards of 131072 x 65536 = 4 GiB + 8 MiB each) on a single MI355X: the 32 GiB
    matrix fits its 288 GB.  Each shard is generated where an 8-GPU rank would generate it (same seed, byte offset of the shard:
    what bench.py --gpus 8 --rows-per-gpu 131072 does), multiplied on its own, and the concatenation must equal the unsharded call."""
    rows, cols, parts = 1 << 20, 1 << 16, 8
    hb, shard = cols // 64, (1 << 20) // 8
    A = hip.alloc(rows * cols // 2)
    sA = hip.alloc((rows // 64) * hb * 4)
    x, sx = hip.alloc(cols // 2), hip.alloc(hb * 4)
    lib = hip.lib
    for k in range(parts):                   # per-shard generation == whole-matrix generation (counter-based fill)
        hip.check(lib.clv_fill_random_nibbles(A.ptr + k * shard * cols // 2, shard * cols // 2, 0xC5, k * shard * cols // 2, None))
        hip.check(lib.clv_fill_random_scales(sA.ptr + k * (shard // 64) * hb * 4, (shard // 64) * hb, 0xC6, k * (shard // 64) * hb, None))
    hip.check(lib.clv_fill_random_nibbles(x.ptr, x.nbytes, 0xC7, 0, None))
    hip.check(lib.clv_fill_random_scales(sx.ptr, hb, 0xC8, 0, None))
    r, sr = hip.alloc(rows // 2), hip.alloc(rows // 64 * 4)
    hip.check(lib.clm4_mvm(A.ptr, sA.ptr, rows, cols, x.ptr, sx.ptr, r.ptr, sr.ptr, None, None))
    r_h, sr_h = r.download(np.uint8), sr.download(np.float32)
    qx, sxh = x.download(np.uint8), sx.download(np.float32)
    parts_r, parts_s = [], []
    for k in range(parts):
        rk, srk = hip.alloc(shard // 2), hip.alloc(shard // 64 * 4)
        hip.check(lib.clm4_mvm(A.ptr + k * shard * cols // 2, sA.ptr + k * (shard // 64) * hb * 4, shard, cols, x.ptr, sx.ptr, rk.ptr, srk.ptr, None, None))
        parts_r.append(rk.download(np.uint8))
        parts_s.append(srk.download(np.float32))
    assert same(np.concatenate(parts_r), r_h) and same(np.concatenate(parts_s), sr_h)
    assert len(set(r_h[:: 4099].tolist())) > 16            # not a constant
    # sampled 64-row blocks, at least one in every shard incl. first / last block of a shard, vs the scalar oracle
    groups = sorted({0, 1, rows // 64 - 1} | {k * (shard // 64) + o for k in range(parts) for o in (0, 1027, shard // 64 - 1)})
    for g in groups[::2] + [groups[-1]]:
        blk = _download(hip, A.ptr + g * 64 * cols // 2, 64 * cols // 2, np.uint8)
        sblk = _download(hip, sA.ptr + g * hb * 4, hb * 4, np.float32)
        d = oracle.m4_rowdots(blk, sblk, 64, cols, qx, sxh)
        ro, sro = oracle.v4_quantize(np.concatenate([d, np.zeros(64, np.float32)]))
        assert same(r_h[g * 32:(g + 1) * 32], ro[:32]) and bits(sr_h[g]) == bits(sro[0]), g
    # the fill of a shard is the whole matrix's bytes at that offset (checked on the last shard's first KiB against a fresh fill)
    probe = hip.alloc(1024)
    hip.check(lib.clv_fill_random_nibbles(probe.ptr, 1024, 0xC5, (parts - 1) * shard * cols // 2, None))
    assert same(probe.download(np.uint8), _download(hip, A.ptr + (parts - 1) * shard * cols // 2, 1024, np.uint8))


def test_mvm_few_row_groups_streaming_matrix(hip, oracle):
    """a matrix that streams from HBM (> 256 MiB) but has only 16 row groups takes the 8-lanes-per-row kernel; 11 LDS chunks
    of columns; plain and fused-with-scaleAndAdd, against the whole oracle result"""
    rows, cols = 1024, 655360 + 128
    A, sA, x, sx = _device_matrix(hip, rows, cols, 0x51)
    qA, sAh, qx, sxh = A.download(np.uint8), sA.download(np.float32), x.download(np.uint8), sx.download(np.float32)
    ro, sro = oracle.m4_mvm(qA, sAh, rows, cols, qx, sxh)
    r, sr = hip.alloc(rows // 2), hip.alloc(rows // 64 * 4)
    hip.check(hip.lib.clm4_mvm(A.ptr, sA.ptr, rows, cols, x.ptr, sx.ptr, r.ptr, sr.ptr, None, None))
    assert same(r.download(np.uint8), ro) and same(sr.download(np.float32), sro)
    rng = np.random.default_rng(5)
    qu = rng.integers(0, 256, rows // 2).astype(np.uint8)
    qu[(qu >> 4) == 8] ^= 0x80
    qu[(qu & 0xF) == 8] ^= 0x08
    su = rng.uniform(0.5, 2, rows // 64).astype(np.float32)
    du, dsu = hip.to_device(qu), hip.to_device(su)
    r2, sr2 = hip.alloc(rows // 2), hip.alloc(rows // 64 * 4)
    hip.check(hip.lib.clm4_mvm_scale_and_add(A.ptr, sA.ptr, rows, cols, x.ptr, sx.ptr, du.ptr, dsu.ptr, 0.25, None, None,
                                             r2.ptr, sr2.ptr, None, None))
    r2o, sr2o = oracle.v4_scale_and_add(qu, su, ro, sro, 0.25)
    assert same(r2.download(np.uint8), r2o) and same(sr2.download(np.float32), sr2o)


@pytest.mark.parametrize("G", [2048, 8192])
def test_gemm_sampled_against_definition(hip, oracle, G):
    """8192^3 is BASELINE config 4 -- the very call bench.py times"""
    M = N = K = G
    kb = K // 64
    A, sA, _, _ = _device_matrix(hip, M, K, 0x6E)
    B, sB, _, _ = _device_matrix(hip, N, K, 0x6F)
    Cd = hip.alloc(M * N * 4)
    hip.check(hip.lib.clm4_gemm(A.ptr, sA.ptr, M, K, B.ptr, sB.ptr, N, Cd.ptr, None))
    Ch = Cd.download(np.float32).reshape(M, N)
    qA, qB = A.download(np.uint8), B.download(np.uint8)
    sAh, sBh = sA.download(np.float32), sB.download(np.float32)
    rng = np.random.default_rng(1)
    rows = sorted(set([0, 63, 64, 127, 128, M - 1] + rng.integers(0, M, 6).tolist()))
    cols = sorted(set([0, 15, 16, 64, 129, N - 1] + rng.integers(0, N, 6).tolist()))
    # the GEMM definition evaluated element by element (exact product + one rounding = fmaf)
    for i in rows:
        for j in cols:
            acc = np.float32(0)
            S = oracle.v4_word_isums(qA[i * K // 2:(i + 1) * K // 2], qB[j * K // 2:(j + 1) * K // 2]).reshape(kb, 8).sum(1)
            for b in range(kb):
                c = np.float32(np.float32(sAh[(i >> 6) * kb + b] * np.float32(1.0 / 49.0)) * sBh[(j >> 6) * kb + b])
                acc = np.float32(np.float64(c) * np.float64(S[b]) + np.float64(acc))
            assert bits(acc) == bits(Ch[i, j]), (i, j)


def test_gemm_4096_two_pipelines_agree():
    """4096^3 in full: the FP6 kernel (LDS-DMA staging, block-scaled MFMA) and the int8-MFMA kernel (register staging, int32 results)
    share no code below the ABI, so equal digests over the whole of C check every tile of both; three runs of the FP6 kernel
    must also be identical (a staging race would show up as tiles that come and go).  The kernel switch is read once per
    process, hence child processes."""
    import os
    import subprocess
    import sys
    code = (
        "import ctypes as C, hashlib, numpy as np\n"
        "from clover_amd.lib_binding import CloverHip\n"
        "hip = CloverHip(); lib = hip.lib; G = 4096\n"
        "A, B = hip.alloc(G * G // 2), hip.alloc(G * G // 2)\n"
        "sA, sB = hip.alloc((G // 64) ** 2 * 4), hip.alloc((G // 64) ** 2 * 4)\n"
        "Cc = hip.alloc(G * G * 4)\n"
        "for t, sd in ((A, 11), (B, 12)): hip.check(lib.clv_fill_random_nibbles(t.ptr, t.nbytes, sd, 0, None))\n"
        "for t, sd in ((sA, 13), (sB, 14)): hip.check(lib.clv_fill_random_scales(t.ptr, t.nbytes // 4, sd, 0, None))\n"
        "for rep in range(3):\n"
        "    hip.check(lib.clv_memset(Cc.ptr, 0xFF, Cc.nbytes, None))\n"
        "    hip.check(lib.clm4_gemm(A.ptr, sA.ptr, G, G, B.ptr, sB.ptr, G, Cc.ptr, None))\n"
        "    print('digest', hashlib.sha256(Cc.download(np.uint8).tobytes()).hexdigest())\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    digests = {}
    for kern in ("fp6", "i8"):
        env = dict(os.environ)
        env.pop("CLV_GEMM_KERNEL", None)
        if kern == "i8":
            env["CLV_GEMM_KERNEL"] = "i8"
        out = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        digests[kern] = [ln.split()[1] for ln in out.stdout.splitlines() if ln.startswith("digest")]
        assert len(digests[kern]) == 3
    assert len(set(digests["fp6"])) == 1, digests
    assert digests["fp6"][0] == digests["i8"][0], digests


def test_sharded_c_api_single_process(hip, oracle):
    """clm4_sharded_* (one process, N devices, RCCL gather): with the devices visible here (1 on the test
    box; 8 on a full node) the gathered result must equal the unsharded clm4_mvm byte for byte."""
    vp, u64 = C.c_void_p, C.c_uint64
    lib = hip.lib
    rows, cols = 1024, 2048
    ndev = hip.device_count
    ctx = vp()
    hip.check(lib.clm4_sharded_create(C.byref(ctx), ndev, None, rows, cols))
    try:
        hip.check(lib.clm4_sharded_fill_random(ctx, 77))
        # the same matrix, unsharded
        A, sA = hip.alloc(rows * cols // 2), hip.alloc((rows // 64) * (cols // 64) * 4)
        hip.check(lib.clv_fill_random_nibbles(A.ptr, A.nbytes, 77, 0, None))
        hip.check(lib.clv_fill_random_scales(sA.ptr, sA.nbytes // 4, 78, 0, None))
        rng = np.random.default_rng(5)
        qx = (rng.integers(0, 256, size=cols // 2, dtype=np.uint8) & 0x77).astype(np.uint8)
        sx = rng.uniform(0.5, 2, size=cols // 64).astype(np.float32)
        r, sr = np.zeros(rows // 2, np.uint8), np.zeros(rows // 64, np.float32)
        hip.check(lib.clm4_sharded_mvm(ctx, qx.ctypes.data, sx.ctypes.data, 1, r.ctypes.data, sr.ctypes.data))
        r1, sr1 = hip.m4_mvm(A.download(np.uint8), sA.download(np.float32), rows, cols, qx, sx)
        assert same(r, r1) and same(sr, sr1)
        ro, sro = oracle.m4_mvm(A.download(np.uint8), sA.download(np.float32), rows, cols, qx, sx)
        assert same(r, ro) and same(sr, sro)
        # upload path: scatter a host matrix
        A_h, sA_h = A.download(np.uint8), sA.download(np.float32)      # keep the host arrays alive across the call
        hip.check(lib.clm4_sharded_upload(ctx, A_h.ctypes.data, sA_h.ctypes.data))
        hip.check(lib.clm4_sharded_mvm(ctx, qx.ctypes.data, sx.ctypes.data, 1, r.ctypes.data, sr.ctypes.data))
        assert same(r, ro)
        b, c = u64(), u64()
        hip.check(lib.clm4_sharded_info(ctx, ndev - 1, None, C.byref(b), C.byref(c), None, None))
        assert b.value + c.value == rows
        # row-sharded GEMM against the unsharded call and the oracle (rows / ndev stays a multiple of 128 for ndev = 1, 2, 4, 8)
        N = 256
        rngb = np.random.default_rng(6)
        qB = (rngb.integers(0, 256, size=N * cols // 2, dtype=np.uint8) & 0x77).astype(np.uint8)
        sB = rngb.uniform(0.5, 2, size=(N // 64) * (cols // 64)).astype(np.float32)
        Cs = np.zeros(rows * N, np.float32)
        for _ in range(2):                                             # second call reuses the buffers
            hip.check(lib.clm4_sharded_gemm(ctx, qB.ctypes.data, sB.ctypes.data, N, 1, Cs.ctypes.data))
        assert same(Cs, hip.m4_gemm(A_h, sA_h, rows, cols, qB, sB, N).reshape(-1))
        assert same(Cs[: 128 * N], oracle.m4_gemm(A_h[: 128 * cols // 2], sA_h[: 2 * (cols // 64)], 128, cols, qB, sB, N).reshape(-1))
        cd = vp()
        hip.check(lib.clm4_sharded_gemm_result(ctx, 0, C.byref(cd)))
        assert cd.value
    finally:
        hip.check(lib.clm4_sharded_destroy(ctx))


@pytest.mark.parametrize("parts,rows,selftest", [(3, 896, None), (4, 1024, None), (1, 256, None), (1, 256, "1"), (1, 384, "ragged")])
def test_sharded_loop_calls_equal_the_unsharded_mvm(hip, parts, rows, selftest, monkeypatch):
    """the loop form of the one-process API (clm4_sharded_set_x / _loop_begin / _mvm_enqueue / _sync / _step_timing / _result_buf): ragged
    (7 blocks over 3 shards) and equal shards on device 0 (exchanges are copies on the exchange stream), both result buffers, x replaced
    between steps -- every shard's full result == clm4_mvm of the whole matrix"""
    vp = C.c_void_p
    lib = hip.lib
    cols = 2048
    if selftest:         # one shard through RCCL with a communicator of one rank: the grouped in-place all-gather pair / the per-owner broadcasts
        monkeypatch.setenv("CLV_SHARDED_RCCL_SELFTEST", selftest)
    else:
        monkeypatch.delenv("CLV_SHARDED_RCCL_SELFTEST", raising=False)
    devs = (C.c_int * parts)(*([0] * parts))
    ctx = vp()
    hip.check(lib.clm4_sharded_create(C.byref(ctx), parts, devs, rows, cols))
    try:
        ranks, equal = C.c_int(), C.c_int()
        hip.check(lib.clm4_sharded_comm_info(ctx, C.byref(ranks), C.byref(equal)))
        assert ranks.value == (1 if selftest else 0) and (selftest != "ragged" or equal.value == 0)
        hip.check(lib.clm4_sharded_fill_random(ctx, 91))
        A, sA = hip.alloc(rows * cols // 2), hip.alloc((rows // 64) * (cols // 64) * 4)
        hip.check(lib.clv_fill_random_nibbles(A.ptr, A.nbytes, 91, 0, None))
        hip.check(lib.clv_fill_random_scales(sA.ptr, sA.nbytes // 4, 92, 0, None))
        qA, sAh = A.download(np.uint8), sA.download(np.float32)
        hip.check(lib.clm4_sharded_loop_begin(ctx, 4))
        rng = np.random.default_rng(3)
        want, keep = [], []
        for step in range(4):
            qx = (rng.integers(0, 256, size=cols // 2, dtype=np.uint8) & 0x77).astype(np.uint8)
            sx = rng.uniform(0.5, 2, size=cols // 64).astype(np.float32)
            keep.append((qx, sx))                           # the copy out of host memory is only enqueued
            hip.check(lib.clm4_sharded_set_x(ctx, qx.ctypes.data, sx.ctypes.data, 1))
            hip.check(lib.clm4_sharded_mvm_enqueue(ctx, step, 1))
            want.append(hip.m4_mvm(qA, sAh, rows, cols, qx, sx))
            if step >= 2:                                   # buffers alternate: after steps 2 and 3 both hold the latest two results
                hip.check(lib.clm4_sharded_sync(ctx))
                for part in range(parts):
                    for buf, ref in ((step & 1, want[step]), ((step - 1) & 1, want[step - 1])):
                        rp, sp = vp(), vp()
                        hip.check(lib.clm4_sharded_result_buf(ctx, part, buf, C.byref(rp), C.byref(sp)))
                        assert same(_download(hip, rp.value, rows // 2, np.uint8), ref[0]), (step, part, buf)
                        assert same(_download(hip, sp.value, rows // 16, np.float32), ref[1]), (step, part, buf)
        km, gm = C.c_float(), C.c_float()
        hip.check(lib.clm4_sharded_step_timing(ctx, parts - 1, 3, C.byref(km), C.byref(gm)))
        assert km.value > 0 and gm.value >= 0
        assert lib.clm4_sharded_mvm_enqueue(ctx, 4, 1) != 0              # no event slot reserved for step 4
        assert lib.clm4_sharded_step_timing(ctx, 0, 9, C.byref(km), C.byref(gm)) != 0
    finally:
        hip.check(lib.clm4_sharded_destroy(ctx))


@pytest.mark.parametrize("stochastic", [False, True])
def test_scale_and_add_2p28_streaming_kernels_whole_result(hip, oracle, stochastic):
    """n = 2^28 (three 128 MiB vectors: beyond the Infinity Cache): the streaming (non-temporal) instances of the round-5 scaleAndAdd kernels --
    k_v4_scale_and_add_blk<true> and k_v4_scale_and_add_st<64, true> with its block-scalar phases -- which no smaller test reaches.  The
    WHOLE result, every nibble and scale, against the scalar oracle (device-generated operands, two calls: the second continues the stream)."""
    lib = hip.lib
    n = 1 << 28
    qu, su, qv, sv = hip.alloc(n // 2), hip.alloc(n // 16), hip.alloc(n // 2), hip.alloc(n // 16)
    r, sr = hip.alloc(n // 2), hip.alloc(n // 16)
    hip.check(lib.clv_fill_random_nibbles(qu.ptr, qu.nbytes, 81, 0, None))
    hip.check(lib.clv_fill_random_nibbles(qv.ptr, qv.nbytes, 82, 0, None))
    hip.check(lib.clv_fill_random_scales(su.ptr, su.nbytes // 4, 83, 0, None))
    hip.check(lib.clv_fill_random_scales(sv.ptr, sv.nbytes // 4, 84, 0, None))
    hqu, hsu, hqv, hsv = qu.download(np.uint8), su.download(np.float32), qv.download(np.uint8), sv.download(np.float32)
    st, o = (hip.new_rng(2028, 5), oracle.rng(2028, 5)) if stochastic else (None, None)
    for a in (0.5, -1.25):
        hip.check(lib.clv4_scale_and_add(qu.ptr, su.ptr, qv.ptr, sv.ptr, a, n, r.ptr, sr.ptr, st.ptr if st else None, None))
        ro, sro = oracle.v4_scale_and_add(hqu, hsu, hqv, hsv, a, o)
        assert same(r.download(np.uint8), ro) and same(sr.download(np.float32), sro), a
    if stochastic:
        assert np.array_equal(hip.rng_get(st)[1], oracle.rng_keys(o)[1])


def test_mvm_f32_streaming_kernel_whole_result(hip, oracle):
    """clm4_mvm_f32 on a matrix beyond the Infinity Cache (8320 x 65664 nibbles = 273 MB): the non-temporal instance of the kernel with its
    round-5 q / 16 conversion, five x chunks incl. a ragged one -- every row dot against the scalar oracle, bit for bit"""
    lib = hip.lib
    M, N = 8320, 65536 + 128
    A, sA = hip.alloc(M * N // 2), hip.alloc((M // 64) * (N // 64) * 4)
    hip.check(lib.clv_fill_random_nibbles(A.ptr, A.nbytes, 91, 0, None))
    hip.check(lib.clv_fill_random_scales(sA.ptr, sA.nbytes // 4, 92, 0, None))
    x = (np.random.default_rng(93).normal(size=N) * 2).astype(np.float32)
    hA, hsA = A.download(np.uint8), sA.download(np.float32)
    xd, rd = hip.to_device(x), hip.alloc(4 * M)
    hip.check(lib.clm4_mvm_f32(A.ptr, sA.ptr, M, N, xd.ptr, rd.ptr, None))
    assert same(rd.download(np.float32), oracle.m4_mvm_f32(hA, hsA, M, N, x))


def test_v8_scale_and_add_2p27_block_kernel_whole_result(hip, oracle):
    """CloverVector8::scaleAndAdd where it takes the once-per-block kernel on its own (n >= 2^27: the operands leave the Infinity Cache):
    the whole result of a ragged n against the oracle"""
    n = (1 << 27) + 128 * 9
    rng = np.random.default_rng(31)
    qu, qv = rng.integers(-127, 128, n, dtype=np.int8), rng.integers(-127, 128, n, dtype=np.int8)
    su, sv = rng.uniform(0.5, 2, n // 64).astype(np.float32), rng.uniform(0.5, 2, n // 64).astype(np.float32)
    r, sr = hip.v8_scale_and_add(qu, su, qv, sv, -0.75)
    ro, sro = oracle.v8_scale_and_add(qu, su, qv, sv, -0.75)
    assert r.tobytes() == ro.tobytes() and sr.tobytes() == sro.tobytes()
