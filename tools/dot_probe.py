#!/usr/bin/env python3
"""dot EXACT / FAST at n = 2^24 a few times (for rocprofv3 --kernel-trace --stats)."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from clover_amd.lib_binding import DOT_EXACT, DOT_FAST, CloverHip  # noqa: E402

hip = CloverHip()
lib = hip.lib
n = 1 << 24
qa, qb = hip.alloc(n // 2), hip.alloc(n // 2)
sa, sb = hip.alloc(n // 16), hip.alloc(n // 16)
out = hip.alloc(8)
for t, sd in ((qa, 1), (qb, 2)):
    hip.check(lib.clv_fill_random_nibbles(t.ptr, t.nbytes, sd, 0, None))
for t, sd in ((sa, 3), (sb, 4)):
    hip.check(lib.clv_fill_random_scales(t.ptr, t.nbytes // 4, sd, 0, None))
for _ in range(5):
    hip.check(lib.clv4_dot(qa.ptr, sa.ptr, qb.ptr, sb.ptr, n, DOT_EXACT, out.ptr, None, None))
    hip.check(lib.clv4_dot(qa.ptr, sa.ptr, qb.ptr, sb.ptr, n, DOT_FAST, out.ptr + 4, None, None))
hip.sync()
print("dot probe done")
