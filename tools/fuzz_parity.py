#!/usr/bin/env python3
"""One-off extended fuzz on the GPU box: the randomised parity sweeps of tests/test_gpu_random_shapes.py (GPU == oracle, bit for bit) over many
more seeds than the suite runs, plus random (n, a) scaleAndAdd / dot cases around the kernel-switch sizes of round 5.  Prints a summary line;
exit code 1 on the first mismatch (with the seed).     python tools/fuzz_parity.py [first_seed] [count]"""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import test_gpu_random_shapes as T  # noqa: E402
from clover_amd.lib_binding import DOT_EXACT, DOT_FAST, CloverHip  # noqa: E402
from conftest import random_packed  # noqa: E402
from oracle.binding import Oracle  # noqa: E402

first, count = (int(sys.argv[1]) if len(sys.argv) > 1 else 100), (int(sys.argv[2]) if len(sys.argv) > 2 else 150)
hip, orc = CloverHip(device=0), Oracle()
t0, done = time.time(), 0
for seed in range(first, first + count):
    try:
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
