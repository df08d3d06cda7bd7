#!/usr/bin/env python3
"""One-off extended fuzz on the GPU box: the randomised parity sweeps of tests/test_gpu_random_shapes.py (GPU == oracle, bit for bit) over many
more seeds than the suite runs, plus random (n, a) scaleAndAdd / dot cases around the kernel-switch sizes of round 5.  Prints a summary line;
exit code 1 on the first mismatch (with the seed).     python tools/fuzz_parity.py [first_seed] [count] [a|b]
Body `b` (the rows the random-shape tests do not draw): threshold in both modes, 4- and 8-bit (FAST == the lowest-index restatement, REFERENCE ==
the oracle's heap walk, nibble for nibble) with random (n, K) and tie-heavy data; CloverVector8 scaleAndAdd in both rounding modes; the fused
mvm + scaleAndAdd; stochastic vector ops with every segment shape forced (1 / 4 / 16 / 64) at sizes that are not multiples of the shape.
Body `c`: matrices whose column count sits around the kernels' LDS chunk sizes (16384 for the fp32 vector, 32768 for the 8-bit one, 65536 for
the 4-bit one) -- mvm in both rounding modes, mvm_v8, mvm_f32, the fused mvm + scaleAndAdd -- and transposes / stochastic matrix quantize
on shapes that mix the 256-tile and the 64-tile kernels."""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import test_gpu_random_shapes as T  # noqa: E402
import test_next_rows as NR  # noqa: E402
from clover_amd.lib_binding import DOT_EXACT, DOT_FAST, THRESHOLD_REFERENCE, CloverHip  # noqa: E402
from conftest import random_packed  # noqa: E402
from oracle.binding import Oracle  # noqa: E402

first, count = (int(sys.argv[1]) if len(sys.argv) > 1 else 100), (int(sys.argv[2]) if len(sys.argv) > 2 else 150)
body = sys.argv[3] if len(sys.argv) > 3 else "a"
hip, orc = CloverHip(device=0), Oracle()
t0, done = time.time(), 0
same = NR.same


def body_b(seed):
    rng = np.random.default_rng(50000 + seed)
    # ---- threshold, 4-bit: n ragged inside its padding, K anywhere in [0, n]; data kinds as tests/test_next_rows.py
    n = int(rng.integers(1, 40000)) if seed % 5 else int(rng.integers(131072 - 300, 131072 + 4000))
    npad = (n + 127) // 128 * 128
    k = int(rng.integers(0, n + 1)) if seed % 3 else int(rng.integers(0, min(n, 64) + 1))
    x = np.zeros(npad, np.float32)
    kind = seed % 4
    if kind == 0:
        x[:n] = rng.integers(-40, 41, size=n)
    elif kind == 1:
        x[:n] = rng.choice(np.array([-7, -3, 0, 3, 7], np.float32), size=n)
    elif kind == 2:
        x[:n] = rng.normal(size=n) * np.repeat(rng.uniform(0.1, 10, size=npad // 64), 64)[:n]
    else:
        x[:n] = T._data(rng, npad, 3)[:n]
    q, s = orc.v4_quantize(x)
    ref = orc.v4_threshold(q, s, n, k)
    assert same(hip.v4_threshold(q, s, n, k, mode=THRESHOLD_REFERENCE), ref), f"v4 threshold REFERENCE n={n} k={k} kind={kind}"
    assert same(hip.v4_threshold(q, s, n, k), NR._threshold_lowest_index(orc, q, s, n, k)), f"v4 threshold FAST n={n} k={k} kind={kind}"
    # ---- threshold, 8-bit
    n8 = int(rng.integers(1, 30000))
    npad8 = (n8 + 127) // 128 * 128
    k8 = int(rng.integers(0, n8 + 1))
    x8 = np.zeros(npad8, np.float32)
    x8[:n8] = rng.integers(-40, 41, size=n8) if seed % 2 else rng.normal(size=n8)
    q8, s8 = orc.v8_quantize(x8)
    ref8 = orc.v8_threshold(q8, s8, n8, k8)
    assert same(hip.v8_threshold(q8, s8, n8, k8, mode=THRESHOLD_REFERENCE), ref8), f"v8 threshold REFERENCE n={n8} k={k8}"
    mags = np.abs((q8.astype(np.float32) * np.repeat(s8, 64)) / np.float32(127.0))[:n8]
    fast8 = hip.v8_threshold(q8, s8, n8, k8)
    assert np.array_equal(np.sort(mags[fast8[:n8] != 0]), np.sort(mags[ref8[:n8] != 0])) and same(fast8[n8:], q8[n8:]), f"v8 threshold FAST n={n8} k={k8}"
    # ---- CloverVector8 scaleAndAdd, both rounding modes, one shared stream
    m = 128 * int(rng.integers(1, 600))
    u, v = T._data(rng, m, seed % 4), T._data(rng, m, (seed + 3) % 4)
    (qu, su), (qv, sv) = orc.v8_quantize(u), orc.v8_quantize(v)
    a = float(rng.uniform(-2, 2))
    r, sr = hip.v8_scale_and_add(qu, su, qv, sv, a)
    ro, sro = orc.v8_scale_and_add(qu, su, qv, sv, a)
    assert same(r, ro) and same(sr, sro), f"v8 scaleAndAdd m={m}"
    g, o = hip.new_rng(3 + seed, 11), orc.rng(3 + seed, 11)
    r, sr = hip.v8_scale_and_add(qu, su, qv, sv, a, rng=g)
    ro, sro = orc.v8_scale_and_add(qu, su, qv, sv, a, o)
    assert same(r, ro) and same(sr, sro), f"v8 scaleAndAdd stochastic m={m}"
    # ---- fused mvm + scaleAndAdd == the two calls on the oracle, both rounding modes
    M, N = 128 * int(rng.integers(1, 5)), 128 * int(rng.integers(1, 10))
    qA, sA = random_packed(rng, M * N)[0], rng.uniform(0.5, 2.0, size=(M // 64) * (N // 64)).astype(np.float32)
    (qx, sx), (qy, sy) = random_packed(rng, N), random_packed(rng, M)
    for st in (False, True):
        gg, oo = (hip.new_rng(9 + seed, 5), orc.rng(9 + seed, 5)) if st else (None, None)
        t_, st_, r, sr = hip.m4_mvm_scale_and_add(qA, sA, M, N, qx, sx, qy, sy, a, rng=gg)
        to, sto = orc.m4_mvm(qA, sA, M, N, qx, sx, oo)
        ro, sro = orc.v4_scale_and_add(qy, sy, to, sto, a, oo)
        assert same(t_, to) and same(st_, sto) and same(r, ro) and same(sr, sro), f"fused mvm+scaleAndAdd {M}x{N} stochastic={st}"
    # ---- stochastic vector ops with a forced segment shape, size not a multiple of the shape
    seg = (1, 4, 16, 64)[seed % 4]
    nb = int(rng.integers(1, 3 * 32 * seg + 40))
    nv = 128 * ((nb + 1) // 2)
    xv, yv = T._data(rng, nv, (seed + 1) % 4), T._data(rng, nv, (seed + 2) % 4)
    (qa, sa), (qb, sb) = orc.v4_quantize(xv), orc.v4_quantize(yv)
    assert hip.lib.clv_rng_set_segments(seg) == 0
    try:
        g, o = hip.new_rng(77 + seed, 3), orc.rng(77 + seed, 3)
        qs, ss = hip.v4_quantize(xv, rng=g)
        qso, sso = orc.v4_quantize(xv, o)
        assert same(qs, qso) and same(ss, sso), f"quantize stochastic seg={seg} n={nv}"
        r, sr = hip.v4_scale_and_add(qa, sa, qb, sb, a, rng=g)
        ro, sro = orc.v4_scale_and_add(qa, sa, qb, sb, a, o)
        assert same(r, ro) and same(sr, sro), f"scaleAndAdd stochastic seg={seg} n={nv}"
        q8s, s8s = hip.v8_quantize(xv, rng=g)
        q8o, s8o = orc.v8_quantize(xv, o)
        assert same(q8s, q8o) and same(s8s, s8o), f"v8 quantize stochastic seg={seg} n={nv}"
        assert np.array_equal(hip.rng_get(g)[1], orc.rng_keys(o)[1]), f"stream position seg={seg}"
    finally:
        hip.lib.clv_rng_set_segments(0)


def body_c(seed):
    rng = np.random.default_rng(70000 + seed)
    chunk = (16384, 32768, 65536)[seed % 3]
    N = max(128, chunk * int(rng.integers(1, 3)) + 128 * int(rng.integers(-3, 4)))
    M = 128 * int(rng.integers(1, 4))
    qA, sA = random_packed(rng, M * N)[0], rng.uniform(0.5, 2.0, size=(M // 64) * (N // 64)).astype(np.float32)
    sA[rng.integers(0, sA.size, 3)] = np.float32(10.0 ** rng.integers(-20, 20))
    (qx, sx), (qy, sy) = random_packed(rng, N), random_packed(rng, M)
    a = float(rng.uniform(-2, 2))
    for st in (False, True):
        g, o = (hip.new_rng(1 + seed, 2), orc.rng(1 + seed, 2)) if st else (None, None)
        r, sr = hip.m4_mvm(qA, sA, M, N, qx, sx, rng=g)
        ro, sro = orc.m4_mvm(qA, sA, M, N, qx, sx, o)
        assert same(r, ro) and same(sr, sro), f"mvm {M}x{N} stochastic={st}"
        t_, st_, r2, sr2 = hip.m4_mvm_scale_and_add(qA, sA, M, N, qx, sx, qy, sy, a, rng=g)
        to, sto = orc.m4_mvm(qA, sA, M, N, qx, sx, o)
        r2o, sr2o = orc.v4_scale_and_add(qy, sy, to, sto, a, o)
        assert same(t_, to) and same(st_, sto) and same(r2, r2o) and same(sr2, sr2o), f"fused mvm+scaleAndAdd {M}x{N} stochastic={st}"
    x32 = T._data(rng, N, seed % 4)
    q8, s8 = orc.v8_quantize(x32)
    r8, sr8 = hip.m4_mvm_v8(qA, sA, M, N, q8, s8)
    r8o, sr8o = orc.m4_mvm_v8(qA, sA, M, N, q8, s8)
    assert same(r8, r8o) and same(sr8, sr8o), f"mvm_v8 {M}x{N}"
    assert same(hip.m4_mvm_f32(qA, sA, M, N, x32), orc.m4_mvm_f32(qA, sA, M, N, x32)), f"mvm_f32 {M}x{N}"
    # transposes and stochastic matrix quantize: shapes that are / are not multiples of 256
    Mt, Nt = 128 * int(rng.integers(1, 9)), 128 * int(rng.integers(1, 13))
    qT, sT = random_packed(rng, Mt * Nt)[0], rng.uniform(0.5, 2.0, size=(Mt // 64) * (Nt // 64)).astype(np.float32)
    qt, st2 = hip.m4_transpose(qT, sT, Mt, Nt)
    qto, sto2 = orc.m4_transpose(qT, sT, Mt, Nt)
    assert same(qt, qto) and same(st2, sto2), f"transpose {Mt}x{Nt}"
    A = T._data(rng, Mt * Nt, (seed + 1) % 4).reshape(Mt, Nt)
    g, o = hip.new_rng(21 + seed, 4), orc.rng(21 + seed, 4)
    for _ in range(2):
        qs, ss = hip.m4_quantize(A, rng=g)
        qso, sso = orc.m4_quantize(A, o)
        assert same(qs, qso) and same(ss, sso), f"matrix quantize stochastic {Mt}x{Nt}"
    assert np.array_equal(hip.rng_get(g)[1], orc.rng_keys(o)[1]), "stream position after matrix quantize"


for seed in range(first, first + count):
    try:
        if body == "c":
            body_c(seed)
            done += 1
            continue
        if body == "b":
            body_b(seed)
            done += 1
            continue
        T.test_vector_ops_random(hip, orc, seed)
        T.test_matrix_ops_random(hip, orc, seed)
        if seed % 4 == 0:
            T.test_gemm_random(hip, orc, seed)
        rng = np.random.default_rng(90000 + seed)
        # scaleAndAdd around the switch to the block-scalar kernel (2^18 elements) and with ragged 64-block chunks; dot FAST / EXACT, 4- and 8-bit
        n = 128 * int(rng.integers((1 << 18) // 128 - 40, (1 << 18) // 128 + 4000))
        (qu, su), (qv, sv) = random_packed(rng, n), random_packed(rng, n)
        su[rng.integers(0, n // 64, 5)] = np.float32(10.0 ** rng.integers(-38, 38))
        a = float(rng.uniform(-3, 3))
        r, sr = hip.v4_scale_and_add(qu, su, qv, sv, a)
        ro, sro = orc.v4_scale_and_add(qu, su, qv, sv, a)
        ok = np.isfinite(sro)
        assert np.array_equal(sr[ok].view(np.uint32), sro[ok].view(np.uint32)) and np.array_equal(r.reshape(-1, 32)[ok], ro.reshape(-1, 32)[ok]), "scaleAndAdd"
        m = 128 * int(rng.integers(1, 3000))
        (qa, sa), (qb, sb) = random_packed(rng, m), random_packed(rng, m)
        assert np.float32(hip.v4_dot(qa, sa, qb, sb, mode=DOT_EXACT)).tobytes() == np.float32(orc.v4_dot(qa, sa, qb, sb)).tobytes(), "dot4 exact"
        terms = float(np.abs(np.repeat(sa * sb, 64)).sum()) * 49.0 / 49.0
        assert abs(float(hip.v4_dot(qa, sa, qb, sb, mode=DOT_FAST)) - orc.v4_dot_f64(qa, sa, qb, sb)) <= 2e-6 * terms + 1e-6, "dot4 fast"
        x8, y8 = (rng.normal(size=m) * 3).astype(np.float32), (rng.normal(size=m)).astype(np.float32)
        (q8a, s8a), (q8b, s8b) = orc.v8_quantize(x8), orc.v8_quantize(y8)
        assert np.float32(hip.v8_dot(q8a, s8a, q8b, s8b, mode=DOT_EXACT)).tobytes() == np.float32(orc.v8_dot(q8a, s8a, q8b, s8b)).tobytes(), "dot8 exact"
    except AssertionError as e:
        print(f"MISMATCH at seed {seed}: {e}")
        sys.exit(1)
    done += 1
print(f"fuzz ok: seeds {first}..{first + count - 1} ({done} rounds) in {time.time() - t0:.1f} s")
