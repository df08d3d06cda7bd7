#!/usr/bin/env python3
"""Persistent IHT kernel (iht_persist.hip) against the launch-per-step loop (iht4.hip): same inputs, every output bit for bit, then timing.
Usage: iht_persist_probe.py [check] [time] [N ...]"""
import os
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from clover_amd.lib_binding import CloverHip  # noqa: E402

hip = CloverHip(path=os.environ.get("CLV_LIB"))      # CLV_LIB: another build of the library, for same-box A/B runs
lib = hip.lib


def problem(m, n, seed):
    Phi, PhiT = hip.alloc(m * n // 2), hip.alloc(m * n // 2)
    sPhi, sPhiT = hip.alloc((m // 64) * (n // 64) * 4), hip.alloc((m // 64) * (n // 64) * 4)
    hip.check(lib.clv_fill_random_nibbles(Phi.ptr, Phi.nbytes, seed, 0, None))
    hip.check(lib.clv_fill_random_scales(sPhi.ptr, sPhi.nbytes // 4, seed + 1, 0, None))
    hip.check(lib.clm4_transpose(Phi.ptr, sPhi.ptr, m, n, PhiT.ptr, sPhiT.ptr, None))

    def vec(k, sd):
        q, s = hip.alloc(k // 2), hip.alloc(k // 16)
        hip.check(lib.clv_fill_random_nibbles(q.ptr, q.nbytes, sd, 0, None))
        hip.check(lib.clv_fill_random_scales(s.ptr, s.nbytes // 4, sd + 1, 0, None))
        return q, s
    return (Phi, sPhi, PhiT, sPhiT), [vec(n, seed + 2), vec(m, seed + 4), vec(m, seed + 6), vec(m, seed + 8), vec(n, seed + 10)]


def problem8(m, n, seed):
    mat, _ = problem(m, n, seed)

    def vec8(k, sd):
        q, s = hip.alloc(k), hip.alloc(k // 16)
        hip.check(lib.clv_fill_random_nibbles(q.ptr, q.nbytes, sd, 0, None))      # bytes whose nibbles are in [-7, 7]: any int8 but -128
        hip.check(lib.clv_fill_random_scales(s.ptr, s.nbytes // 4, sd + 1, 0, None))
        return q, s
    return mat, [vec8(n, seed + 2), vec8(m, seed + 4), vec8(m, seed + 6), vec8(m, seed + 8), vec8(n, seed + 10)]


def run8(mat, vecs, m, n, x_len, iters, K, mu, thr, persistent, seed=None):
    os.environ["CLV_IHT_PERSISTENT"] = "1" if persistent else "0"
    x, y, t1, t2, t3 = vecs
    rng = hip.new_rng(*seed) if seed else None
    for v in (x, t1, t2, t3):
        hip.check(lib.clv_memset(v[0].ptr, 0x5A, v[0].nbytes, None))
        hip.check(lib.clv_memset(v[1].ptr, 0x3C, v[1].nbytes, None))
    hip.check(lib.clm4_iht_v8(mat[0].ptr, mat[1].ptr, mat[2].ptr, mat[3].ptr, m, n, x[0].ptr, x[1].ptr, x_len, y[0].ptr, y[1].ptr, t1[0].ptr, t1[1].ptr,
                              t2[0].ptr, t2[1].ptr, t3[0].ptr, t3[1].ptr, iters, K, mu, thr, rng.ptr if rng else None, None))
    hip.sync()
    out = [np.concatenate([v[0].download(np.uint8, v[0].nbytes), v[1].download(np.uint8, v[1].nbytes)]) for v in (x, t1, t2, t3)]
    if rng:
        k1, k2 = hip.rng_get(rng)
        out.append(np.concatenate([np.asarray(k1, np.uint64), np.asarray(k2, np.uint64)]).view(np.uint8))
    return out


def check8(seed=None):
    bad = 0
    taken = 0
    cases = [(128, 128), (256, 512), (384, 640), (640, 384), (1024, 2048), (2048, 4096), (4096, 8192), (6144, 4096), (8192, 1024), (128, 8192),
             (1536, 2048), (3072, 4096), (6144, 8192), (2176, 4224), (3968, 8064), (4096, 7680), (2048, 8192)]
    for (m, n) in cases:
        mat, vecs = problem8(m, n, 300 + m + n)
        for thr in (1, 0):
            for (x_len, K, iters) in ((n, n // 4, 5), (n - 37, n // 8 + 3, 3), (n, 0, 2), (n, n, 2), (n, 1, 4)):
                if thr == 0 and K != n // 4:
                    continue
                for mu in (1e-3, 0.05):
                    c0 = lib.clv_iht_persistent_launches()
                    a = run8(mat, vecs, m, n, x_len, iters, K, mu, thr, True, seed)
                    p = lib.clv_iht_persistent_launches() - c0
                    taken += p
                    b = run8(mat, vecs, m, n, x_len, iters, K, mu, thr, False, seed)
                    ok = all(np.array_equal(u, v) for u, v in zip(a, b))
                    if not ok:
                        bad += 1
                        which = [(nm, int(np.flatnonzero(u != v)[0]), int((u != v).sum())) for nm, u, v in zip(("x", "t1", "t2", "t3", "rng state"), a, b) if not np.array_equal(u, v)]
                        print(f"MISMATCH v8 m={m} n={n} thr={thr} x_len={x_len} K={K} iters={iters} mu={mu} persistent={p}: {which}")
                    else:
                        print(f"ok v8 m={m} n={n} thr={thr} x_len={x_len} K={K} iters={iters} mu={mu} persistent={p} (nonzero x bytes {int((a[0][:n] != 0).sum())})")
    print("CHECK v8", "stochastic" if seed else "deterministic", "FAILED" if bad else "PASSED", bad, "persistent launches", taken)
    return bad


def timing8(Ns, seed=None):
    for N in Ns:
        m, n = N // 2, N
        mat, vecs = problem8(m, n, 31)
        x, y, t1, t2, t3 = vecs
        rng = hip.new_rng(*seed) if seed else None
        for persistent in (0, 1):
            os.environ["CLV_IHT_PERSISTENT"] = str(persistent)
            for thr, K in ((1, n // 4), (0, 0)):
                def call(iters):
                    hip.check(lib.clm4_iht_v8(mat[0].ptr, mat[1].ptr, mat[2].ptr, mat[3].ptr, m, n, x[0].ptr, x[1].ptr, n, y[0].ptr, y[1].ptr, t1[0].ptr,
                                              t1[1].ptr, t2[0].ptr, t2[1].ptr, t3[0].ptr, t3[1].ptr, iters, K, 1e-3, thr, rng.ptr if rng else None, None))
                    hip.sync()
                call(10)
                res = {}
                for iters in (100, 1000):
                    best = 1e9
                    for _ in range(3):
                        t0 = time.perf_counter()
                        call(iters)
                        best = min(best, time.perf_counter() - t0)
                    res[iters] = best
                print(f"v8 {'stochastic ' if seed else ''}N={N} persistent={persistent} thr={thr} K={K}: {(res[1000] - res[100]) / 900 * 1e6:.2f} us/iteration", flush=True)


def run(mat, vecs, m, n, x_len, iters, K, mu, thr, persistent, seed=None):
    os.environ["CLV_IHT_PERSISTENT"] = "1" if persistent else "0"
    x, y, t1, t2, t3 = vecs
    rng = hip.new_rng(*seed) if seed else None
    for v in (x, t1, t2, t3):
        hip.check(lib.clv_memset(v[0].ptr, 0x5A, v[0].nbytes, None))
        hip.check(lib.clv_memset(v[1].ptr, 0x3C, v[1].nbytes, None))
    hip.check(lib.clm4_iht(mat[0].ptr, mat[1].ptr, mat[2].ptr, mat[3].ptr, m, n, x[0].ptr, x[1].ptr, x_len, y[0].ptr, y[1].ptr, t1[0].ptr, t1[1].ptr,
                           t2[0].ptr, t2[1].ptr, t3[0].ptr, t3[1].ptr, iters, K, mu, thr, rng.ptr if rng else None, None))
    hip.sync()
    out = [np.concatenate([v[0].download(np.uint8, v[0].nbytes), v[1].download(np.uint8, v[1].nbytes)]) for v in (x, t1, t2, t3)]
    if rng:
        k1, k2 = hip.rng_get(rng)
        out.append(np.concatenate([np.asarray(k1, np.uint64), np.asarray(k2, np.uint64)]).view(np.uint8))
    return out


def check(seed=None):
    bad = 0
    cases = [(128, 128), (256, 512), (384, 640), (640, 384), (1024, 2048), (1536, 1024), (2048, 4096), (4096, 8192), (6144, 4096), (8192, 1024), (128, 8192)]
    for (m, n) in cases:
        mat, vecs = problem(m, n, 100 + m + n)
        for thr in (1, 0):
            for (x_len, K, iters) in ((n, n // 4, 5), (n - 37, n // 8 + 3, 3), (n, 0, 2), (n, n, 2), (n, 1, 4)):
                if thr == 0 and K != n // 4:
                    continue
                for mu in (1e-3, 0.05):
                    a = run(mat, vecs, m, n, x_len, iters, K, mu, thr, True, seed)
                    b = run(mat, vecs, m, n, x_len, iters, K, mu, thr, False, seed)
                    ok = all(np.array_equal(u, v) for u, v in zip(a, b))
                    nz = int((a[0][: n // 2] != 0).sum())
                    if not ok:
                        bad += 1
                        which = [nm for nm, u, v in zip(("x", "t1", "t2", "t3", "rng state"), a, b) if not np.array_equal(u, v)]
                        print(f"MISMATCH m={m} n={n} thr={thr} x_len={x_len} K={K} iters={iters} mu={mu}: {which}")
                    else:
                        print(f"ok m={m} n={n} thr={thr} x_len={x_len} K={K} iters={iters} mu={mu} (nonzero x bytes {nz})")
    print("CHECK", "FAILED" if bad else "PASSED", bad)
    return bad


def timing(Ns, seed=None):
    for N in Ns:
        m, n = N // 2, N
        mat, vecs = problem(m, n, 31)
        x, y, t1, t2, t3 = vecs
        rng = hip.new_rng(*seed) if seed else None
        for persistent in (0, 1):
            os.environ["CLV_IHT_PERSISTENT"] = str(persistent)
            for thr, K in ((1, n // 4), (1, m // 4), (0, 0)):
                def call(iters):
                    hip.check(lib.clm4_iht(mat[0].ptr, mat[1].ptr, mat[2].ptr, mat[3].ptr, m, n, x[0].ptr, x[1].ptr, n, y[0].ptr, y[1].ptr, t1[0].ptr,
                                           t1[1].ptr, t2[0].ptr, t2[1].ptr, t3[0].ptr, t3[1].ptr, iters, K, 1e-3, thr, rng.ptr if rng else None, None))
                    hip.sync()
                call(10)
                res = {}
                for iters in (1, 100, 1000):
                    best = 1e9
                    for _ in range(3):
                        t0 = time.perf_counter()
                        call(iters)
                        best = min(best, time.perf_counter() - t0)
                    res[iters] = best
                per = (res[1000] - res[100]) / 900 * 1e6
                print(f"N={N} persistent={persistent} thr={thr} K={K}: 1 it {res[1]*1e6:.1f} us, 100 it {res[100]*1e6:.1f} us, 1000 it {res[1000]*1e6:.1f} us -> {per:.2f} us/iteration", flush=True)


def stamps(N, thr=1):
    m, n = N // 2, N
    mat, vecs = problem(m, n, 31)
    x, y, t1, t2, t3 = vecs
    G = 256
    buf = hip.alloc(G * 16 * 32 * 8)
    hip.check(lib.clv_memset(buf.ptr, 0, buf.nbytes, None))
    os.environ["CLV_IHT_PERSISTENT"] = "1"
    os.environ["CLV_IHT_DEBUG_STAMPS"] = hex(buf.ptr)
    hip.check(lib.clm4_iht(mat[0].ptr, mat[1].ptr, mat[2].ptr, mat[3].ptr, m, n, x[0].ptr, x[1].ptr, n, y[0].ptr, y[1].ptr, t1[0].ptr,
                           t1[1].ptr, t2[0].ptr, t2[1].ptr, t3[0].ptr, t3[1].ptr, int(os.environ.get("IHT_PROBE_ITERS", "16")), n // 4, 1e-3, thr, None, None))
    hip.sync()
    del os.environ["CLV_IHT_DEBUG_STAMPS"]
    full = buf.download(np.uint64, G * 16 * 32).reshape(G, 16, 32).astype(np.int64)
    st, st15 = full[:, :, :16], full[:, :, 16:]
    ck = full[0, 15, 28:32]
    if ck[3] > ck[1]:
        print(f"  core clock over the launch: {(ck[2] - ck[0]) / ((ck[3] - ck[1]) / 100.0):.0f} MHz ({(ck[3] - ck[1]) / 100.0:.1f} us)")
    names = ["P1", "gather1", "requant1", "barrier", "P2", "gather2", "requant2", "threshold", "tail"]
    used = st[:, 0, 0] != 0
    print(f"N={N} thr={thr}: {int(used.sum())} workgroups; per-phase mean / max over workgroups, iterations 4..15, in us (wave 0's view)")
    seg = (st[used][:, 4:, 1:10] - st[used][:, 4:, 0:9]) / 100.0
    for k, nm in enumerate(names):
        print(f"  {nm:10s} mean {seg[:, :, k].mean():6.2f}  max {seg[:, :, k].max():6.2f}  min {seg[:, :, k].min():6.2f}")
    if thr:
        inner = st[used][:, 4:, :]
        marks = [("count", 7, 10), ("level0 atomics+barrier", 10, 11), ("level0 scan..level1 barrier", 11, 12), ("level1 scan..level2 barrier", 12, 13),
                 ("level2 scan..level3 barrier", 13, 14), ("level3 scan", 14, 15), ("cut-offs, ties, apply", 15, 8)]
        for nm, a, b in marks:
            dseg = (inner[:, :, b] - inner[:, :, a]) / 100.0
            print(f"    threshold/{nm:30s} mean {dseg.mean():6.2f}  max {dseg.max():6.2f}")
    # the global view: the last producer's P-done stamp against every consumer's gather-done stamp (the 100 MHz counter is chip-wide)
    u = st[used]
    for nm, a, b in (("exchange 1", 1, 2), ("exchange 2", 5, 6)):
        lastpub = u[:, 4:, a].max(axis=0)                       # per iteration: when the slowest workgroup finished its row dots
        firstpub = u[:, 4:, a].min(axis=0)
        done = u[:, 4:, b]
        lat = (done - lastpub[None, :]) / 100.0
        lat15 = (st15[used][:, 4:, b] - lastpub[None, :]) / 100.0
        print(f"  {nm}: producers' P-done spread {((lastpub - firstpub) / 100.0).mean():5.2f} us; gather done after the LAST P-done: wave 0 "
              f"mean {lat.mean():5.2f}  min {lat.min():5.2f}  max {lat.max():5.2f}; wave 15 mean {lat15.mean():5.2f}  max {lat15.max():5.2f}")
    per_it = (u[:, 5:, 0] - u[:, 4:-1, 0]) / 100.0
    print(f"  iteration period (stamp 0 to stamp 0): mean {per_it.mean():6.2f}  min {per_it.min():6.2f}  max {per_it.max():6.2f}")
    tot = (st[used][:, 15, 9] - st[used][:, 4, 0]) / 100.0 / 12
    print(f"  iteration  mean {tot.mean():6.2f}")
    start = st[used][:, 0, 0]
    print(f"  first stamp spread over workgroups: {(start.max() - start.min()) / 100.0:.2f} us")


def stamps8(N, thr=1, seed=None):
    """phase stamps of k_iht8_persist (CloverVector8 vectors), as stamps()"""
    m, n = N // 2, N
    mat, vecs = problem8(m, n, 31)
    x, y, t1, t2, t3 = vecs
    G = 256
    buf = hip.alloc(G * 16 * 32 * 8)
    hip.check(lib.clv_memset(buf.ptr, 0, buf.nbytes, None))
    rng = hip.new_rng(*seed) if seed else None
    os.environ["CLV_IHT_PERSISTENT"] = "1"
    os.environ["CLV_IHT_DEBUG_STAMPS"] = hex(buf.ptr)
    hip.check(lib.clm4_iht_v8(mat[0].ptr, mat[1].ptr, mat[2].ptr, mat[3].ptr, m, n, x[0].ptr, x[1].ptr, n, y[0].ptr, y[1].ptr, t1[0].ptr,
                              t1[1].ptr, t2[0].ptr, t2[1].ptr, t3[0].ptr, t3[1].ptr, 16, n // 4, 1e-3, thr, rng.ptr if rng else None, None))
    hip.sync()
    del os.environ["CLV_IHT_DEBUG_STAMPS"]
    full = buf.download(np.uint64, G * 16 * 32).reshape(G, 16, 32).astype(np.int64)
    st = full[:, :, :16]
    names = ["P1", "publish+gather1", "requant1+images", "barrier", "P2", "publish+gather2", "requant2", "threshold", "images+barrier"]
    used = st[:, 0, 0] != 0
    print(f"v8 N={N} thr={thr} {'stochastic' if seed else 'deterministic'}: {int(used.sum())} workgroups; per-phase mean / max over workgroups, "
          f"iterations 4..15, in us (wave 0's view)")
    seg = (st[used][:, 4:, 1:10] - st[used][:, 4:, 0:9]) / 100.0
    for k, nm in enumerate(names):
        print(f"  {nm:18s} mean {seg[:, :, k].mean():6.2f}  max {seg[:, :, k].max():6.2f}  min {seg[:, :, k].min():6.2f}")
    u = st[used]
    if thr:
        inner = u[:, 4:, :]
        for nm, a, b in (("re-quant done .. keys", 7, 10), ("level 0 (atomics, 2 barriers, scan)", 10, 11), ("level 1", 11, 12), ("level 2", 12, 13), ("level 3", 13, 14),
                         ("ties counted", 14, 15), ("rank + apply", 15, 8)):
            dseg = (inner[:, :, b] - inner[:, :, a]) / 100.0
            print(f"    threshold/{nm:36s} mean {dseg.mean():6.2f}  max {dseg.max():6.2f}")
    per_it = (u[:, 5:, 0] - u[:, 4:-1, 0]) / 100.0
    print(f"  iteration period (stamp 0 to stamp 0): mean {per_it.mean():6.2f}")


if __name__ == "__main__":
    args = sys.argv[1:]
    rc = 0
    if not args or "check" in args:
        rc = check()
    if "check8" in args:
        rc = check8()
    if "check8_st" in args:
        rc = check8(seed=(2468, 1357)) or rc
    if "time8" in args:
        timing8([int(a) for a in args if a.isdigit()] or [256, 4096, 8192])
    if "time8_st" in args:
        timing8([int(a) for a in args if a.isdigit()] or [256, 4096, 8192], seed=(5, 6))
    if "check_st" in args:
        rc = check(seed=(12345, 67890))
    if "stamps8" in args:
        for N in [int(a) for a in args if a.isdigit()] or [8192]:
            stamps8(N, 1)
            stamps8(N, 0)
            stamps8(N, 1, seed=(5, 6))
    if "stamps" in args:
        for N in [int(a) for a in args if a.isdigit()] or [256, 8192]:
            stamps(N, 1)
            stamps(N, 0)
    if not args or "time" in args:
        Ns = [int(a) for a in args if a.isdigit()] or [256, 1024, 2048, 4096, 8192]
        timing(Ns)
    if "time_st" in args:
        timing([int(a) for a in args if a.isdigit()] or [256, 4096, 8192], seed=(5, 6))
    sys.exit(1 if rc else 0)
