#!/usr/bin/env python3
"""Stochastic and deterministic quantize / scaleAndAdd at n = 2^28, three launches each (for rocprofv3 --pmc passes)."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from clover_amd.lib_binding import CloverHip  # noqa: E402

hip = CloverHip()
lib = hip.lib
n = 1 << 28
x = hip.alloc(4 * n)
hip.check(lib.clv_fill_random_ints_f32(x.ptr, n, 10, 5, 0, None))
q, s = hip.alloc(n // 2), hip.alloc(n // 16)
q2, s2 = hip.alloc(n // 2), hip.alloc(n // 16)
rng = hip.new_rng(1, 2)
for _ in range(3):
    hip.check(lib.clv4_quantize(x.ptr, n, q.ptr, s.ptr, None, None))
    hip.check(lib.clv4_quantize(x.ptr, n, q.ptr, s.ptr, rng.ptr, None))
    hip.check(lib.clv4_scale_and_add(q.ptr, s.ptr, q.ptr, s.ptr, 0.5, n, q2.ptr, s2.ptr, None, None))
    hip.check(lib.clv4_scale_and_add(q.ptr, s.ptr, q.ptr, s.ptr, 0.5, n, q2.ptr, s2.ptr, rng.ptr, None))
hip.sync()
