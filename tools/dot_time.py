#!/usr/bin/env python3
"""dot EXACT / FAST timings with HIP events (n = 2^24 by default; DOT_N=...)."""
import ctypes as C
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from clover_amd.lib_binding import DOT_EXACT, DOT_FAST, CloverHip  # noqa: E402

hip = CloverHip()
lib = hip.lib
for n in [int(v) for v in os.environ.get("DOT_N", str(1 << 24)).split(",")]:
    qa, qb = hip.alloc(n // 2), hip.alloc(n // 2)
    sa, sb = hip.alloc(n // 16), hip.alloc(n // 16)
    out = hip.alloc(8)
    for t, sd in ((qa, 1), (qb, 2)):
        hip.check(lib.clv_fill_random_nibbles(t.ptr, t.nbytes, sd, 0, None))
    for t, sd in ((sa, 3), (sb, 4)):
        hip.check(lib.clv_fill_random_scales(t.ptr, t.nbytes // 4, sd, 0, None))
    a, b = C.c_void_p(), C.c_void_p()
    hip.check(lib.clv_event_create(C.byref(a)))
    hip.check(lib.clv_event_create(C.byref(b)))
    for mode, name in ((DOT_EXACT, "exact"), (DOT_FAST, "fast")):
        for _ in range(3):
            hip.check(lib.clv4_dot(qa.ptr, sa.ptr, qb.ptr, sb.ptr, n, mode, out.ptr, None, None))
        ts = []
        for _ in range(5):
            hip.check(lib.clv_event_record(a, None))
            for _ in range(4):
                hip.check(lib.clv4_dot(qa.ptr, sa.ptr, qb.ptr, sb.ptr, n, mode, out.ptr, None, None))
            hip.check(lib.clv_event_record(b, None))
            hip.check(lib.clv_event_sync(b))
            ms = C.c_float()
            hip.check(lib.clv_event_elapsed_ms(a, b, C.byref(ms)))
            ts.append(ms.value / 4)
        ms = sorted(ts)[2]
        print(f"n={n} dot {name}: {ms:.4f} ms  {1.125 * n / ms / 1e6:.1f} GB/s  ({ms * 1e6 / (n / 128):.2f} ns per block pair)")
