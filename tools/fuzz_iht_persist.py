#!/usr/bin/env python3
"""Randomised sweep of the persistent IHT / GD kernels (iht_persist.hip) against the launch-per-step loops: random shapes (multiples of 128
up to 8192 that fit the LDS budget -- others fall back, which the sweep counts), K, x_len, mu, iteration counts, data kinds (uniform nibbles,
clustered magnitudes, tiny / huge scales, zero blocks), both rounding modes for 4-bit vectors, and CloverVector8 vectors.
    python tools/fuzz_iht_persist.py [seed0] [count]"""
import os
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from clover_amd.lib_binding import CloverHip  # noqa: E402

hip = CloverHip()
lib = hip.lib
seed0 = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 300


def pack(q):
    return (((q[0::2].astype(np.uint8) & 0xF) << 4) | (q[1::2].astype(np.uint8) & 0xF)).astype(np.uint8)


def make_vec4(rng, n, kind):
    if kind == "cluster":
        q = rng.choice(np.array([-4, 4, 4, 0, 7, -7], np.int8), size=n)
    elif kind == "sparse":
        q = (rng.integers(-7, 8, size=n) * (rng.random(n) < 0.1)).astype(np.int8)
    else:
        q = rng.integers(-7, 8, size=n).astype(np.int8)
    s = rng.uniform(0.5, 2.0, size=n // 64).astype(np.float32)
    if kind == "scales":
        s *= np.float32(10.0) ** rng.integers(-12, 12, size=n // 64).astype(np.float32)
    if kind == "equal":
        s[:] = np.float32(1.25)
    return pack(q), s


bad = fell_back = 0
for case in range(count):
    rng = np.random.default_rng(seed0 + case)
    m = 128 * int(rng.integers(1, 65))
    n = 128 * int(rng.integers(1, 65))
    if rng.random() < 0.3:
        m, n = (n // 256 or 1) * 128, n                       # the reference's cols = 2 rows
    while (m * n) > 4096 * 8192:                               # beyond what 256 CUs hold: make it smaller rather than test the fall-back only
        m = max(128, m // 2 // 128 * 128)
    v8 = rng.random() < 0.3
    st = rng.random() < 0.4
    if v8 and st and 4 * m > 3 * n and rng.random() < 0.7:    # v8 stochastic keeps its draws in dead LDS: m <= 0.75 n or the loop runs launch by launch
        m = max(128, int(rng.integers(1, max(2, 3 * n // 4 // 128 + 1))) * 128)
    kind = str(rng.choice(["uniform", "cluster", "sparse", "scales", "equal"]))
    thr = int(rng.random() < 0.8)
    iters = int(rng.integers(1, 7))
    x_len = n if rng.random() < 0.6 else int(rng.integers(1, n + 1))
    K = int(rng.integers(0, x_len + 2)) if rng.random() < 0.8 else x_len // 4
    mu = float(rng.choice([1e-3, 0.05, 0.5, 2.0]))
    qPhi = pack(rng.integers(-7, 8, size=m * n).astype(np.int8))
    sPhi = rng.uniform(0.5, 2, size=(m // 64) * (n // 64)).astype(np.float32)
    dPhi, dsPhi = hip.to_device(qPhi), hip.to_device(sPhi)
    dT, dsT = hip.alloc(m * n // 2), hip.alloc(sPhi.nbytes)
    hip.check(lib.clm4_transpose(dPhi.ptr, dsPhi.ptr, m, n, dT.ptr, dsT.ptr, None))
    if v8:
        y = rng.integers(-127, 128, size=m).astype(np.int8).view(np.uint8)
        if kind == "cluster":
            y = rng.choice(np.array([127, -127, 120, 119, 0, 64], np.int8), size=m).view(np.uint8)
        sy = rng.uniform(0.5, 2.0, size=m // 64).astype(np.float32)
        vb = 1
    else:
        y, sy = make_vec4(rng, m, kind)
        vb = 2
    dy, dsy = hip.to_device(y), hip.to_device(sy)
    sizes = dict(x=n // vb, sx=n // 16, t1=m // vb, st1=m // 16, t2=m // vb, st2=m // 16, t3=n // vb, st3=n // 16)
    b = {k: hip.alloc(max(sz, 4)) for k, sz in sizes.items()}
    fn = lib.clm4_iht_v8 if v8 else lib.clm4_iht
    outs = []
    for persistent in (1, 0):
        os.environ["CLV_IHT_PERSISTENT"] = str(persistent)
        c0 = lib.clv_iht_persistent_launches()
        rs = hip.new_rng(seed0 + case, 77) if st else None
        for v in b.values():
            hip.check(lib.clv_memset(v.ptr, 0x5A, v.nbytes, None))
        hip.check(fn(dPhi.ptr, dsPhi.ptr, dT.ptr, dsT.ptr, m, n, b["x"].ptr, b["sx"].ptr, x_len, dy.ptr, dsy.ptr, b["t1"].ptr, b["st1"].ptr, b["t2"].ptr,
                     b["st2"].ptr, b["t3"].ptr, b["st3"].ptr, iters, K, mu, thr, rs.ptr if rs else None, None))
        hip.sync()
        o = {k: b[k].download(np.uint8, sz) for k, sz in sizes.items()}
        if rs:
            k1, k2 = hip.rng_get(rs)
            o["rng"] = np.concatenate([np.asarray(k1, np.uint64), np.asarray(k2, np.uint64)]).view(np.uint8)
        outs.append(o)
        if persistent and lib.clv_iht_persistent_launches() == c0:
            fell_back += 1
    diff = [k for k in outs[0] if not np.array_equal(outs[0][k], outs[1][k])]
    if diff:
        bad += 1
        print(f"MISMATCH seed={seed0 + case} m={m} n={n} v8={v8} st={st} kind={kind} thr={thr} iters={iters} x_len={x_len} K={K} mu={mu}: {diff}", flush=True)
print(f"fuzz_iht_persist: {count} cases from seed {seed0}, {bad} mismatches, {fell_back} not eligible for the persistent kernel")
sys.exit(1 if bad else 0)
