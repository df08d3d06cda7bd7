"""The persistent, LDS-resident Q_IHT / Q_GD kernel (clover_amd/csrc/iht_persist.hip; reference loop: test/performance/01_measure.h:923-946,
999-1021) against (1) the oracle's step sequence -- x AND t1, t2, t3 with their scales, bit for bit -- and (2) the launch-per-step loop
of iht4.hip on the same inputs, at the sizes where its row dealing, partial step groups and workgroup counts change.

clm4_iht picks the persistent kernel by itself when the problem qualifies (rounding disabled, threshold FAST or none, m, n <= 8192);
CLV_IHT_PERSISTENT=0 (read per call) forces the launch-per-step loop: that is how the two are compared inside one process."""
import ctypes as C
import os
import threading

import numpy as np
import pytest

from conftest import random_packed

pytestmark = pytest.mark.gpu


def same(a, b):
    return np.array_equal(np.asarray(a).view(np.uint8), np.asarray(b).view(np.uint8))


def nibbles(b):
    hi = (b.astype(np.int8) >> 4).astype(np.int32)
    lo = ((b << 4).astype(np.int8) >> 4).astype(np.int32)
    return np.stack([hi, lo], 1).reshape(-1)


def pack(q):
    return (((q[0::2] & 0xF) << 4) | (q[1::2] & 0xF)).astype(np.uint8)


class Problem:
    def __init__(self, hip, oracle, m, n, seed):
        rng = np.random.default_rng(seed)
        self.m, self.n = m, n
        self.qPhi, _ = random_packed(rng, m * n)
        self.sPhi = rng.uniform(0.5, 2, size=(m // 64) * (n // 64)).astype(np.float32)
        self.qT, self.sT = oracle.m4_transpose(self.qPhi, self.sPhi, m, n)
        self.y = random_packed(rng, m)
        self.d = {k: hip.to_device(v) for k, v in dict(Phi=self.qPhi, sPhi=self.sPhi, PhiT=self.qT, sPhiT=self.sT, y=self.y[0], sy=self.y[1]).items()}
        sizes = dict(x=n // 2, sx=n // 16, t1=m // 2, st1=m // 16, t2=m // 2, st2=m // 16, t3=n // 2, st3=n // 16)
        self.bufs = {k: hip.alloc(max(sz, 4)) for k, sz in sizes.items()}
        self.hip = hip

    def run(self, iters, K, mu, thr, x_len=None, persistent=True, stream=None, sync=True, seed=None):
        """seed = (key1, key2): stochastic rounding from a freshly seeded XORShift state; the state the call leaves behind is returned as "rng" """
        hip, d, b = self.hip, self.d, self.bufs
        os.environ["CLV_IHT_PERSISTENT"] = "1" if persistent else "0"
        rng = hip.new_rng(*seed) if seed else None
        for v in b.values():                                       # the call must write every output itself
            hip.check(hip.lib.clv_memset(v.ptr, 0x5A, v.nbytes, stream))
        hip.check(hip.lib.clm4_iht(d["Phi"].ptr, d["sPhi"].ptr, d["PhiT"].ptr, d["sPhiT"].ptr, self.m, self.n, b["x"].ptr, b["sx"].ptr,
                                   self.n if x_len is None else x_len, d["y"].ptr, d["sy"].ptr, b["t1"].ptr, b["st1"].ptr, b["t2"].ptr, b["st2"].ptr,
                                   b["t3"].ptr, b["st3"].ptr, iters, K, float(mu), thr, rng.ptr if rng else None, stream))
        if not sync:
            return None
        hip.check(hip.lib.clv_stream_sync(stream))
        out = self.outputs()
        if rng:
            k1, k2 = hip.rng_get(rng)
            out["rng"] = np.concatenate([np.asarray(k1, np.uint64), np.asarray(k2, np.uint64)]).view(np.uint8)
        return out

    def outputs(self):
        m, n, b = self.m, self.n, self.bufs
        return {k: b[k].download(np.uint8, sz) for k, sz in dict(x=n // 2, sx=n // 16, t1=m // 2, st1=m // 16, t2=m // 2, st2=m // 16,
                                                                 t3=n // 2, st3=n // 16).items()}


def lowest_index_threshold(oracle, q, s, n, k):
    """FAST mode's tie rule on the CPU: everything above the K-th magnitude, then the first ties by index"""
    mags = np.abs(oracle.v4_restore(q, s))[:n]
    out = nibbles(q).copy()
    if k < n:
        tau = np.sort(mags)[::-1][k - 1] if k > 0 else np.inf
        keep = mags > tau
        ties = np.flatnonzero(mags == tau)[: max(k - int(keep.sum()), 0)]
        keep[ties] = True
        out[:n] *= keep
    return pack(out)


@pytest.mark.parametrize("shape,thr", [((512, 1024), 1), ((384, 640), 1), ((1024, 512), 0)])
def test_persistent_loop_matches_oracle_loop_all_vectors(hip, oracle, shape, thr):
    """x, t1, t2, t3 and their scales after the loop = the oracle's step sequence (the launch-per-step test of test_next_rows.py checks x only)"""
    m, n = shape
    P = Problem(hip, oracle, m, n, 7 + m + n)
    iters, K, mu = 5, n // 4, np.float32(0.002)
    got = P.run(iters, K, mu, thr)
    x = (np.zeros(n // 2, np.uint8), np.ones(n // 64, np.float32))
    t1 = t2 = t3 = None
    for _ in range(iters):
        t1 = oracle.m4_mvm(P.qPhi, P.sPhi, m, n, *x)
        t2 = oracle.v4_scale_and_add(*P.y, *t1, -1.0)
        t3 = oracle.m4_mvm(P.qT, P.sT, n, m, *t2)
        x = oracle.v4_scale_and_add(*x, *t3, float(mu))
        if thr:
            x = (lowest_index_threshold(oracle, x[0], x[1], n, K), x[1])
    for name, want in (("x", x[0]), ("sx", x[1]), ("t1", t1[0]), ("st1", t1[1]), ("t2", t2[0]), ("st2", t2[1]), ("t3", t3[0]), ("st3", t3[1])):
        assert same(got[name], want), name


# (m, n): one unit per workgroup and fewer workgroups than CUs; cols % 512 != 0 (a partial step group); the largest that fits (two units of
# PhiT per workgroup); GD's 1.5 : 1 shape; a single row group against 8192 columns and the reverse
SHAPES = [(128, 128), (256, 512), (384, 640), (640, 384), (2048, 4096), (4096, 8192), (6144, 4096), (128, 8192), (8192, 128)]


@pytest.mark.parametrize("rounding", ["deterministic", "stochastic"])
@pytest.mark.parametrize("shape", SHAPES)
def test_persistent_equals_launch_per_step(hip, oracle, shape, rounding):
    """stochastic: both paths draw from one XORShift stream in the reference's order (2 draws per block for each mvm re-quantisation, then 2
    per block for the scaleAndAdd behind it); the outputs AND the state the call leaves behind must agree.  (m + n < 512 and stochastic:
    clm4_iht takes the launch-per-step loop either way -- the comparison is then trivial, by design.)"""
    m, n = shape
    P = Problem(hip, oracle, m, n, 100 + m + n)
    seed = (12345, 67890) if rounding == "stochastic" else None
    cases = [(1, n, n // 4, 5, 1e-3), (1, n - 37, n // 8 + 3, 3, 0.05), (1, n - 37, n // 8 + 3, 3, 1e-3), (1, n, 0, 2, 1e-3), (1, n, n, 2, 1e-3),
             (1, n, 1, 4, 0.05), (0, n, 0, 4, 1e-3)]
    for thr, x_len, K, iters, mu in cases:
        a = P.run(iters, K, mu, thr, x_len=x_len, persistent=True, seed=seed)
        b = P.run(iters, K, mu, thr, x_len=x_len, persistent=False, seed=seed)
        for name in a:
            assert same(a[name], b[name]), (name, thr, x_len, K, iters, mu)
        if K and thr:
            assert int((nibbles(a["x"])[:x_len] != 0).sum()) <= K


@pytest.mark.parametrize("seed", range(12))
def test_persistent_threshold_on_clustered_magnitudes(hip, oracle, seed):
    """Blocks in which one magnitude dominates: the per-block magnitude tables of the in-kernel threshold are 7-bit fields summed over 8
    lanes, and the field of |q| = 4 straddles bit 32 of the 64-bit table (a carry lost there dropped or kept the wrong ties: found by the
    stochastic sweep in round 6, where one block had 16 elements of magnitude 4).  y = a few distinct values makes x + mu Phi' t2 cluster."""
    m, n = 256, 1024
    P = Problem(hip, oracle, m, n, 900 + seed)
    rng = np.random.default_rng(seed)
    yq = rng.choice(np.array([-4, 4, 4, 4, 0], np.int8), size=m)
    P.y = (pack(yq.astype(np.int32)), np.full(m // 64, np.float32(1.75)))
    P.d["y"], P.d["sy"] = hip.to_device(P.y[0]), hip.to_device(P.y[1])
    for K, mu in ((n // 4, 0.01), (n // 2, 0.2), (37, 1.0)):
        a = P.run(4, K, mu, 1, persistent=True)
        b = P.run(4, K, mu, 1, persistent=False)
        for name in a:
            assert same(a[name], b[name]), (name, K, mu)


@pytest.mark.parametrize("rounding", ["deterministic", "stochastic"])
@pytest.mark.parametrize("shape", [(128, 128), (256, 512), (384, 640), (640, 384), (2048, 4096), (4096, 8192), (6144, 4096), (128, 8192), (8192, 128),
                                   (3072, 4096), (3968, 8064)])
def test_persistent_v8_equals_launch_per_step(hip, oracle, shape, rounding):
    """clm4_iht_v8 -- CloverMatrix4 with CloverVector8 vectors, the reference's published "4-bit" IHT / GD (02_bit04.cpp:140) -- as one
    persistent launch (k_iht8_persist: matrix words re-dealt per fma chain, three nibble images of the int8 vector, v_dot8_i32_i4) against
    its launch-per-step loop (k_m4_mvm8 + k_thresh8_small): x, t1, t2, t3 and their scales, every bit.  (The oracle-loop comparison of
    tests/test_mixed8.py runs through the persistent kernel as well: its sizes qualify.)  Stochastic: the same XORShift stream, draw for
    draw (two draws per 64-element block for each re-quantisation, mvm's in front of scaleAndAdd's: CloverMatrix4.h:1246-1440,
    CloverVector8.h:1104-1126), and the state the call leaves behind; the draws live in LDS that is dead while they are needed, so shapes
    whose draws do not fit there (m > 0.75 n) take the launch-per-step loop -- the counter says which ran."""
    m, n = shape
    seed = (777 + m, 999 + n) if rounding == "stochastic" else None
    rng = np.random.default_rng(300 + m + n)
    qPhi, _ = random_packed(rng, m * n)
    sPhi = rng.uniform(0.5, 2, size=(m // 64) * (n // 64)).astype(np.float32)
    qT, sT = oracle.m4_transpose(qPhi, sPhi, m, n)
    y = rng.integers(-127, 128, size=m).astype(np.int8)
    y[:8] = [127, -127, 120, 119, -120, -121, 112, 0]                 # the carry image of the int8 split: x >= 120
    sy = rng.uniform(0.5, 2, size=m // 64).astype(np.float32)
    d = {k: hip.to_device(v) for k, v in dict(Phi=qPhi, sPhi=sPhi, PhiT=qT, sPhiT=sT, y=y.view(np.uint8), sy=sy).items()}
    sizes = dict(x=n, sx=n // 16, t1=m, st1=m // 16, t2=m, st2=m // 16, t3=n, st3=n // 16)
    b = {k: hip.alloc(max(sz, 4)) for k, sz in sizes.items()}

    def run(iters, K, mu, thr, x_len, persistent):
        os.environ["CLV_IHT_PERSISTENT"] = "1" if persistent else "0"
        rng = hip.new_rng(*seed) if seed else None
        for v in b.values():
            hip.check(hip.lib.clv_memset(v.ptr, 0x5A, v.nbytes, None))
        hip.check(hip.lib.clm4_iht_v8(d["Phi"].ptr, d["sPhi"].ptr, d["PhiT"].ptr, d["sPhiT"].ptr, m, n, b["x"].ptr, b["sx"].ptr, x_len, d["y"].ptr, d["sy"].ptr,
                                      b["t1"].ptr, b["st1"].ptr, b["t2"].ptr, b["st2"].ptr, b["t3"].ptr, b["st3"].ptr, iters, K, float(mu), thr,
                                      rng.ptr if rng else None, None))
        hip.sync()
        out = {k: b[k].download(np.uint8, sz) for k, sz in sizes.items()}
        if rng:
            k1, k2 = hip.rng_get(rng)
            out["rng"] = np.concatenate([np.asarray(k1, np.uint64), np.asarray(k2, np.uint64)])
        return out

    # which shapes the persistent kernel takes: everything deterministic; stochastic where both phases' draws fit their overlays
    groups1, groups2 = (-(-(n // 64) // 4) + 1) & ~1, (-(-(m // 64) // 4) + 1) & ~1        # groups of 4 blocks, an even number (ihtp8_layout)
    fits = (-(-(m // 64) // 4)) * 512 <= 3 * groups1 * 128 and (-(-(n // 64) // 4)) * 512 <= 3 * groups2 * 128 + 10240
    expect_persistent = seed is None or (fits and (m + n) // 64 * 4 >= 32)     # (the per-iteration jump T^(D - 16) wants D >= 32 draws)
    launches = hip.lib.clv_iht_persistent_launches()
    cases = [(1, n, n // 4, 5, 1e-3), (1, n - 37, n // 8 + 3, 3, 0.05), (1, n, 0, 2, 1e-3), (1, n, n, 2, 1e-3), (1, n, 1, 4, 0.05), (0, n, 0, 4, 1e-3),
             (1, n, n // 2, 3, 0.5)]
    for thr, x_len, K, iters, mu in cases:
        a_, b_ = run(iters, K, mu, thr, x_len, True), run(iters, K, mu, thr, x_len, False)
        for name in a_:
            assert same(a_[name], b_[name]), (name, thr, x_len, K, iters, mu)
    assert hip.lib.clv_iht_persistent_launches() - launches == (len(cases) if expect_persistent else 0)
    os.environ.pop("CLV_IHT_PERSISTENT", None)


def test_persistent_calls_on_two_streams_are_chained(hip, oracle):
    """two persistent launches at once could each hold part of the chip and wait for the rest for ever: launches on different streams are
    chained by an event (iht_persist.hip persist_chain).  Two host threads, two streams, alternating calls: every result is the right one."""
    m, n = 1024, 2048
    P = [Problem(hip, oracle, m, n, 500 + i) for i in range(2)]
    want = [p.run(6, n // 4, 1e-3, 1, persistent=False) for p in P]
    os.environ["CLV_IHT_PERSISTENT"] = "1"
    streams = []
    for _ in range(2):
        s = C.c_void_p()
        hip.check(hip.lib.clv_stream_create(C.byref(s)))
        streams.append(s)
    errors = []

    def worker(i):
        try:
            for _ in range(20):
                p, d, b = P[i], P[i].d, P[i].bufs
                hip.check(hip.lib.clm4_iht(d["Phi"].ptr, d["sPhi"].ptr, d["PhiT"].ptr, d["sPhiT"].ptr, m, n, b["x"].ptr, b["sx"].ptr, n, d["y"].ptr,
                                           d["sy"].ptr, b["t1"].ptr, b["st1"].ptr, b["t2"].ptr, b["st2"].ptr, b["t3"].ptr, b["st3"].ptr, 6, n // 4,
                                           1e-3, 1, None, streams[i]))
            hip.check(hip.lib.clv_stream_sync(streams[i]))
        except Exception as e:                                     # noqa: BLE001
            errors.append(e)

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(120)
        assert not t.is_alive(), "a persistent launch did not finish"
    assert not errors, errors
    for i in range(2):
        got = P[i].outputs()
        for name in got:
            assert same(got[name], want[i][name]), (i, name)
    for s in streams:
        hip.check(hip.lib.clv_stream_destroy(s))


def test_persistent_is_the_path_taken(hip, oracle):
    """guard against a silent fall-back at the size the benchmarks quote (N = 8192): the library's own counter says that the calls ran as
    ONE persistent launch each (4-bit and CloverVector8 vectors, both rounding modes), and -- informative, with a wide margin -- such a
    call is well inside the time of the launch-per-step loop's 3 launches per iteration"""
    import time
    m, n = 4096, 8192
    P = Problem(hip, oracle, m, n, 9)
    c0 = hip.lib.clv_iht_persistent_launches()
    P.run(20, n // 4, 1e-3, 1, persistent=True)
    P.run(20, n // 4, 1e-3, 1, persistent=True, seed=(7, 9))
    P.run(20, 0, 1e-3, 0, persistent=True)
    assert hip.lib.clv_iht_persistent_launches() == c0 + 3
    times = {}
    for persistent in (True, False):
        P.run(50, n // 4, 1e-3, 1, persistent=persistent)
        t0 = time.perf_counter()
        P.run(200, n // 4, 1e-3, 1, persistent=persistent)
        times[persistent] = time.perf_counter() - t0
    assert times[True] < 0.9 * times[False], times
    os.environ.pop("CLV_IHT_PERSISTENT", None)


def test_persistent_launch_counter(hip, oracle):
    """clv_iht_persistent_launches counts exactly the calls that ran as one persistent launch: qualifying calls with the switch on"""
    P = Problem(hip, oracle, 512, 1024, 11)
    c0 = hip.lib.clv_iht_persistent_launches()
    P.run(3, 256, 1e-3, 1, persistent=True)
    P.run(3, 256, 1e-3, 1, persistent=True, seed=(3, 4))
    assert hip.lib.clv_iht_persistent_launches() == c0 + 2
    P.run(3, 256, 1e-3, 1, persistent=False)
    P.run(3, 256, 1e-3, 2, persistent=True)                        # REFERENCE threshold mode: the launch-per-step loop
    assert hip.lib.clv_iht_persistent_launches() == c0 + 2
    os.environ.pop("CLV_IHT_PERSISTENT", None)
