#!/usr/bin/env python3
"""clv4_dot FAST at the reference's published sizes (n = 2^24, 2^26, 2^29) and 2^30: ms per call, warm (same operands) -- run once per setting of
CLV_DOT_FAST_U / CLV_DOT_FAST_TWO_LAUNCHES (read once per process)."""
import ctypes as C
import json
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from clover_amd.lib_binding import DOT_FAST, CloverHip  # noqa: E402

hip = CloverHip(path=os.environ.get("CLV_LIB"))
lib = hip.lib
vp = C.c_void_p
res = {"CLV_DOT_FAST_U": os.environ.get("CLV_DOT_FAST_U"), "two_launches": os.environ.get("CLV_DOT_FAST_TWO_LAUNCHES")}
out = hip.alloc(8)
for logn in (24, 26, 29, 30):
    n = 1 << logn
    q, s = hip.alloc(n // 2), hip.alloc(n // 16)
    q2, s2 = hip.alloc(n // 2), hip.alloc(n // 16)
    for b, sd in ((q, 1), (q2, 2)):
        hip.check(lib.clv_fill_random_nibbles(b.ptr, b.nbytes, sd, 0, None))
    for b, sd in ((s, 3), (s2, 4)):
        hip.check(lib.clv_fill_random_scales(b.ptr, b.nbytes // 4, sd, 0, None))
    fn = lambda: hip.check(lib.clv4_dot(q.ptr, s.ptr, q2.ptr, s2.ptr, n, DOT_FAST, out.ptr, None, None))
    reps = 200 if logn <= 26 else 20
    for _ in range(5):
        fn()
    hip.sync()
    a, b = vp(), vp()
    hip.check(lib.clv_event_create(C.byref(a)))
    hip.check(lib.clv_event_create(C.byref(b)))
    ts = []
    for _ in range(5):
        hip.check(lib.clv_event_record(a, None))
        for _ in range(reps):
            fn()
        hip.check(lib.clv_event_record(b, None))
        hip.check(lib.clv_event_sync(b))
        ms = C.c_float()
        hip.check(lib.clv_event_elapsed_ms(a, b, C.byref(ms)))
        ts.append(ms.value / reps)
    ms = sorted(ts)[2]
    import numpy as np
    res[f"n2^{logn}"] = {"us": round(ms * 1e3, 2), "GB/s": round(1.125 * n / ms / 1e6, 1), "bits": hex(int(out.download(np.uint32)[0]))}
    del q, s, q2, s2
print(json.dumps(res))
