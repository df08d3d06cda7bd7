#!/usr/bin/env python3
"""The three-launch large-vector threshold (threshold4.hip: k_th4_count_hist0, k_th4_select_persist, k_th4_apply6) against the six-launch form
(CLV_THRESHOLD_THREE_LAUNCH=0, read per call) on the same inputs, and both timed.

    python tools/thresh3_probe.py check            random sizes / data kinds, bit-exact comparison
    python tools/thresh3_probe.py time [logn ...]  us per call (k = n / 4), back-to-back calls on one stream"""
import os
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from clover_amd.lib_binding import CloverHip  # noqa: E402

hip = CloverHip(path=os.environ.get("CLV_LIB"))      # CLV_LIB: another build of the library, for same-box A/B runs
lib = hip.lib


def make(rng, n_pad, kind):
    q = rng.integers(0, 256, size=n_pad // 2, dtype=np.uint8)
    s = rng.uniform(0.5, 2, size=n_pad // 64).astype(np.float32)
    if kind == "sparse":                                   # mostly zeros: tau = 0 for larger k (the general path of the persistent kernel)
        q[rng.random(q.size) < 0.9] = 0
    elif kind == "equal":                                  # one scale: every tie class is huge
        s[:] = 1.25
    elif kind == "wide":                                   # scales over 60 octaves
        s = np.exp2(rng.uniform(-30, 30, size=s.size)).astype(np.float32)
    elif kind == "subnormal":                              # tiny scales: subnormal keys
        s = (rng.uniform(0.5, 2, size=s.size) * 1e-39).astype(np.float32)
    elif kind == "mixed":                                  # a few large blocks among tiny ones
        s = np.where(rng.random(s.size) < 0.01, s * 1e6, s * 1e-3).astype(np.float32)
    elif kind == "zeroscale":
        s[rng.random(s.size) < 0.5] = 0.0
    elif kind == "inf":                                    # a few infinite / NaN scales: their blocks' magnitudes all share one key
        s[rng.random(s.size) < 0.02] = np.inf
        s[rng.random(s.size) < 0.01] = np.nan
    elif kind == "edge":                                   # scales next to the subnormal border and to overflow
        s = np.where(rng.random(s.size) < 0.5, s * 1e-37, s * 1e37).astype(np.float32)
    return q, s


def run(q, s, n, n_pad, k, three):
    os.environ["CLV_THRESHOLD_THREE_LAUNCH"] = "1" if three else "0"
    dq, ds = hip.to_device(q), hip.to_device(s)
    hip.check(lib.clv4_threshold(dq.ptr, ds.ptr, n, n_pad, k, None, None))
    hip.sync()
    return dq.download(np.uint8, q.size)


def check(count=200, seed0=0):
    bad = 0
    for case in range(count):
        rng = np.random.default_rng(seed0 + case)
        logn = int(rng.integers(17, 25))
        n_pad = (int(rng.integers(1 << logn, 2 << logn)) + 127) // 128 * 128
        n_pad = max(n_pad, 131072 + 128)
        n = n_pad if rng.random() < 0.5 else int(rng.integers(max(131073, n_pad - 127), n_pad + 1))
        kind = str(rng.choice(["uniform", "sparse", "equal", "wide", "subnormal", "mixed", "zeroscale", "inf", "edge"]))
        q, s = make(rng, n_pad, kind)
        nz = int(n * (0.1 if kind == "sparse" else 1))
        k = int(rng.choice([1, 2, n // 4, n // 2, n - 1, int(rng.integers(1, n)), min(n - 1, nz + 5), max(1, nz // 2)]))
        a, b = run(q, s, n, n_pad, k, True), run(q, s, n, n_pad, k, False)
        os.environ["CLV_THRESHOLD_FORCE_CAND"] = "1"
        c = run(q, s, n, n_pad, k, True)
        del os.environ["CLV_THRESHOLD_FORCE_CAND"]
        if not np.array_equal(a, b) or not np.array_equal(c, b):
            bad += 1
            d = np.flatnonzero(a != b)
            print(f"MISMATCH case={seed0 + case} n={n} n_pad={n_pad} k={k} kind={kind}: {d.size} bytes differ, first at {d[0]}", flush=True)
    print(f"thresh3 check: {count} cases, {bad} mismatches")
    return bad


def timing(logns):
    for logn in logns:
        n = 1 << logn
        q, s = hip.alloc(n // 2), hip.alloc(n // 16)
        hip.check(lib.clv_fill_random_scales(s.ptr, n // 64, 8, 0, None))
        for three in (0, 1):
            os.environ["CLV_THRESHOLD_THREE_LAUNCH"] = str(three)
            best = 1e9
            for _ in range(5):
                hip.check(lib.clv_fill_random_nibbles(q.ptr, q.nbytes, 7, 0, None))
                hip.sync()
                t0 = time.perf_counter()
                hip.check(lib.clv4_threshold(q.ptr, s.ptr, n, n, n // 4, None, None))
                hip.sync()
                best = min(best, time.perf_counter() - t0)
            print(f"n=2^{logn} three_launch={three}: {best * 1e6:.1f} us per call (host clock, one call + sync)", flush=True)


def stamps(logn):
    """phase stamps of k_th4_select_persist (100 MHz wall clock), one call"""
    n = 1 << logn
    q, s = hip.alloc(n // 2), hip.alloc(n // 16)
    hip.check(lib.clv_fill_random_scales(s.ptr, n // 64, 8, 0, None))
    dbg = hip.alloc(256 * 16 * 8)
    os.environ["CLV_THRESHOLD_THREE_LAUNCH"] = "1"
    for rep in range(3):
        hip.check(lib.clv_fill_random_nibbles(q.ptr, q.nbytes, 7, 0, None))
        hip.check(lib.clv_memset(dbg.ptr, 0, dbg.nbytes, None))
        os.environ["CLV_THRESHOLD_DEBUG_STAMPS"] = hex(dbg.ptr)
        hip.check(lib.clv4_threshold(q.ptr, s.ptr, n, n, n // 4, None, None))
        hip.sync()
        del os.environ["CLV_THRESHOLD_DEBUG_STAMPS"]
    st = dbg.download(np.uint64, 256 * 16).reshape(256, 16)
    used = st[:, 0] != 0
    st = st[used].astype(np.int64)
    t0 = st[:, 0].min()
    names = ["start", "select0", "level1 loop", "hist1 flush", "hand-over 1", "level2 loop", "hist2 flush", "hand-over 2", "ties", "scan+end"]
    print(f"n=2^{logn}: {used.sum()} workgroups; stamps relative to the first workgroup's start, us (min / median / max over workgroups)")
    for i, nm in enumerate(names):
        col = (st[:, i] - t0) / 100.0
        print(f"  {i} {nm:15s} {col.min():7.2f} {np.median(col):7.2f} {col.max():7.2f}")


if __name__ == "__main__":
    args = sys.argv[1:]
    rc = 0
    if "check" in args:
        nums = [int(a) for a in args if a.isdigit()]
        rc = check(*(nums[:2] if nums else []))
    if "stamps" in args:
        for logn in [int(a) for a in args if a.isdigit()] or [20, 28]:
            stamps(logn)
    if "time" in args:
        timing([int(a) for a in args if a.isdigit()] or [18, 20, 24, 28])
    sys.exit(1 if rc else 0)
