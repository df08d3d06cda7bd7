"""CloverVector4::threshold(K) on vectors beyond one workgroup, FAST mode (CloverVector4.h:1913-2060), in THREE launches:
k_th4_count_hist0 -> k_th4_select_persist (one resident workgroup per CU, two hand-overs inside the launch) -> k_th4_apply3
(clover_amd/csrc/threshold4.hip).  Checked three ways:
  (1) against the CPU restatement of FAST's rule -- everything above the K-th magnitude, then the lowest-index ties -- on the oracle's
      restored values, and against the oracle's own threshold for the surviving multiset;
  (2) against the six-launch form of rounds 2-5 (CLV_THRESHOLD_THREE_LAUNCH=0, read per call) bit for bit, on the data shapes where the
      persistent kernel changes its path: tau = 0 (K beyond the non-zero elements), one scale for all blocks, scales over 60 octaves,
      subnormal, infinite and NaN scales, the raw nibble -8, ragged n; with the candidate words in registers and through memory;
  (3) at n = 2^28 through what the size allows: equality with the six-launch form, idempotence, survivor count."""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def same(a, b):
    return np.array_equal(np.asarray(a).view(np.uint8), np.asarray(b).view(np.uint8))


def nibbles(b):
    hi = (b.astype(np.int8) >> 4).astype(np.int32)
    lo = ((b << 4).astype(np.int8) >> 4).astype(np.int32)
    return np.stack([hi, lo], 1).reshape(-1)


def lowest_index_rule(oracle, q, s, n, k):
    mags = np.abs(oracle.v4_restore(q, s))[:n]
    out = nibbles(q).copy()
    tau = np.sort(mags)[::-1][k - 1]
    keep = mags > tau
    ties = np.flatnonzero(mags == tau)[: max(k - int(keep.sum()), 0)]
    keep[ties] = True
    out[:n] *= keep
    return (((out[0::2] & 0xF) << 4) | (out[1::2] & 0xF)).astype(np.uint8)


def make(rng, n_pad, kind):
    q = rng.integers(0, 256, size=n_pad // 2, dtype=np.uint8)              # raw nibbles, -8 included
    s = rng.uniform(0.5, 2, size=n_pad // 64).astype(np.float32)
    if kind == "sparse":
        q[rng.random(q.size) < 0.9] = 0
    elif kind == "equal":
        s[:] = 1.25
    elif kind == "wide":
        s = np.exp2(rng.uniform(-30, 30, size=s.size)).astype(np.float32)
    elif kind == "subnormal":
        s = (rng.uniform(0.5, 2, size=s.size) * 1e-39).astype(np.float32)
    elif kind == "edge":
        s = np.where(rng.random(s.size) < 0.5, s * 1e-37, s * 1e37).astype(np.float32)
    elif kind == "inf":
        s[rng.random(s.size) < 0.02] = np.inf
        s[rng.random(s.size) < 0.01] = np.nan
    elif kind == "zeroscale":
        s[rng.random(s.size) < 0.5] = 0.0
    return q, s


def run(hip, q, s, n, k, three=True, force_cand=False):
    os.environ["CLV_THRESHOLD_THREE_LAUNCH"] = "1" if three else "0"
    if force_cand:
        os.environ["CLV_THRESHOLD_FORCE_CAND"] = "1"
    try:
        return hip.v4_threshold(q, s, n, k)
    finally:
        os.environ.pop("CLV_THRESHOLD_THREE_LAUNCH", None)
        os.environ.pop("CLV_THRESHOLD_FORCE_CAND", None)


@pytest.mark.parametrize("kind", ["uniform", "sparse", "equal", "wide", "subnormal", "edge", "zeroscale"])
@pytest.mark.parametrize("n_pad,ragged", [(131072 + 128, 0), (1 << 18, 37), ((1 << 20) + 640, 1), (1 << 22, 0)])
def test_three_launch_threshold_follows_the_lowest_index_rule(hip, oracle, kind, n_pad, ragged):
    rng = np.random.default_rng(n_pad + ragged + len(kind))
    n = n_pad - ragged
    q, s = make(rng, n_pad, kind)
    nz = int((nibbles(q)[:n] != 0).sum())
    for k in (1, n // 4, min(n - 1, nz + 5), n - 1, int(rng.integers(2, n - 1))):
        want = lowest_index_rule(oracle, q, s, n, k)
        got = run(hip, q, s, n, k)
        assert same(got, want), (kind, n, k)
        assert same(run(hip, q, s, n, k, force_cand=True), want), (kind, n, k, "candidate words through memory")
        assert same(run(hip, q, s, n, k, three=False), want), (kind, n, k, "six launches")
        assert same(run(hip, got, s, n, k), got), (kind, n, k, "idempotent")
    # the reference's surviving multiset (the oracle's heap walk; which of several equal magnitudes survive is the reference's own order)
    k = n // 4
    mags = np.abs(oracle.v4_restore(q, s))[:n]
    kept = nibbles(run(hip, q, s, n, k))[:n] != 0
    kept_ref = nibbles(oracle.v4_threshold(q, s, n, k))[:n] != 0
    assert np.array_equal(np.sort(mags[kept]), np.sort(mags[kept_ref]))


@pytest.mark.parametrize("seed", range(6))
def test_three_launch_threshold_with_infinite_and_nan_scales_equals_the_six_launch_form(hip, seed):
    """blocks with an infinite / NaN scale put ALL their magnitudes on one key: the bin of level 0 that holds them is not a one-candidate
    bin (k_th4_select_persist takes its general loops there).  The CPU restatement sorts NaN differently from the key order, so the
    six-launch kernels of rounds 2-5 are the reference here."""
    rng = np.random.default_rng(900 + seed)
    n_pad = (1 << 19) + 128 * seed
    n = n_pad - (seed % 3) * 21
    q, s = make(rng, n_pad, "inf")
    bad = int(np.isinf(s).sum() + np.isnan(s).sum()) * 64
    for k in (1, bad // 2 + 1, bad + 1000, n // 2, n - 1):
        a, b = run(hip, q, s, n, k), run(hip, q, s, n, k, three=False)
        assert same(a, b), (seed, k)
        assert same(run(hip, q, s, n, k, force_cand=True), b), (seed, k)


def test_three_launch_threshold_at_full_size(hip):
    """n = 2^28 (BASELINE's largest vector): the six-launch form's result bit for bit, idempotent, exactly k survivors when tau > 0"""
    lib = hip.lib
    n = 1 << 28
    q, q2, s = hip.alloc(n // 2), hip.alloc(n // 2), hip.alloc(n // 16)
    hip.check(lib.clv_fill_random_scales(s.ptr, n // 64, 8, 0, None))
    for k in (n // 4, 12345, n - 1000):
        for buf, three in ((q, "1"), (q2, "0")):
            hip.check(lib.clv_fill_random_nibbles(buf.ptr, buf.nbytes, 7, 0, None))
            os.environ["CLV_THRESHOLD_THREE_LAUNCH"] = three
            hip.check(lib.clv4_threshold(buf.ptr, s.ptr, n, n, k, None, None))
        hip.sync()
        a, b = q.download(np.uint8, n // 2), q2.download(np.uint8, n // 2)
        assert np.array_equal(a, b), k
        os.environ["CLV_THRESHOLD_THREE_LAUNCH"] = "1"
        hip.check(lib.clv4_threshold(q.ptr, s.ptr, n, n, k, None, None))
        hip.sync()
        assert np.array_equal(q.download(np.uint8, n // 2), a), (k, "idempotent")
        survivors = int(np.count_nonzero(a & 0xF0) + np.count_nonzero(a & 0x0F))
        assert survivors <= k
        if k <= n // 4:                                           # 15/16 of random nibbles are non-zero: tau > 0, nothing is a zero tie
            assert survivors == k
    os.environ.pop("CLV_THRESHOLD_THREE_LAUNCH", None)


def test_three_launch_threshold_beyond_the_register_resident_range(hip):
    """n = 2^29 - 5 inside a padding of 2^29: a workgroup of the persistent kernel owns more than 16 x 1024 blocks, so the candidate words
    go through memory (k_th4_select_persist<false>) -- at the real size, not forced; the six-launch form's result bit for bit"""
    lib = hip.lib
    n_pad = 1 << 29
    n = n_pad - 5
    q, q2, s = hip.alloc(n_pad // 2), hip.alloc(n_pad // 2), hip.alloc(n_pad // 16)
    hip.check(lib.clv_fill_random_scales(s.ptr, n_pad // 64, 18, 0, None))
    for k in (n // 4, 4321):
        for buf, three in ((q, "1"), (q2, "0")):
            hip.check(lib.clv_fill_random_nibbles(buf.ptr, buf.nbytes, 17, 0, None))
            os.environ["CLV_THRESHOLD_THREE_LAUNCH"] = three
            hip.check(lib.clv4_threshold(buf.ptr, s.ptr, n, n_pad, k, None, None))
        hip.sync()
        assert np.array_equal(q.download(np.uint8, n_pad // 2), q2.download(np.uint8, n_pad // 2)), k
    os.environ.pop("CLV_THRESHOLD_THREE_LAUNCH", None)


def test_three_launch_threshold_on_two_streams_and_in_a_graph(hip, oracle):
    """the persistent kernel joins the chain of persistent launches (one at a time per device, iht_persist.hip): two host threads on two
    streams, every result right; and the three launches replay from a captured graph (the control block goes back zero after every call;
    the first, ordinary call on the stream allocates it -- a capture cannot)"""
    import threading
    lib = hip.lib
    rng = np.random.default_rng(77)
    n = 1 << 20
    data = [make(rng, n, "uniform") for _ in range(2)]
    k = n // 8
    want = [lowest_index_rule(oracle, q, s, n, k) for q, s in data]
    streams = []
    for _ in range(2):
        st = C.c_void_p()
        hip.check(lib.clv_stream_create(C.byref(st)))
        streams.append(st)
    dev = [(hip.to_device(q), hip.to_device(s), hip.to_device(q)) for q, s in data]
    errors = []

    def worker(i):
        try:
            dq, ds, d0 = dev[i]
            for _ in range(10):
                hip.check(lib.clv_memcpy_d2d(dq.ptr, d0.ptr, dq.nbytes, streams[i]))
                hip.check(lib.clv4_threshold(dq.ptr, ds.ptr, n, n, k, None, streams[i]))
            hip.check(lib.clv_stream_sync(streams[i]))
        except Exception as e:                                     # noqa: BLE001
            errors.append(e)

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(120)
        assert not t.is_alive(), "a threshold call did not finish"
    assert not errors, errors
    for i in range(2):
        assert same(dev[i][0].download(np.uint8, n // 2), want[i]), i
    # a captured graph on stream 0 (which has made ordinary calls: workspace and control block exist)
    rt = C.CDLL("libamdhip64.so")
    graph, gexec = C.c_void_p(), C.c_void_p()
    dq, ds, d0 = dev[0]
    assert rt.hipStreamBeginCapture(streams[0], 0) == 0
    hip.check(lib.clv_memcpy_d2d(dq.ptr, d0.ptr, dq.nbytes, streams[0]))
    hip.check(lib.clv4_threshold(dq.ptr, ds.ptr, n, n, k, None, streams[0]))
    assert rt.hipStreamEndCapture(streams[0], C.byref(graph)) == 0
    assert rt.hipGraphInstantiate(C.byref(gexec), graph, None, None, 0) == 0
    for rep in range(3):
        hip.check(lib.clv_memset(dq.ptr, 0x5A, dq.nbytes, streams[0]))
        assert rt.hipGraphLaunch(gexec, streams[0]) == 0
        hip.check(lib.clv_stream_sync(streams[0]))
        assert same(dq.download(np.uint8, n // 2), want[0]), rep
    assert rt.hipGraphExecDestroy(gexec) == 0 and rt.hipGraphDestroy(graph) == 0
    for st in streams:
        hip.check(lib.clv_stream_destroy(st))
