#!/usr/bin/env python3
"""clv4_threshold_mode(REFERENCE) at a few (n, k): where the heap walk's time goes (make_heap vs the walk).  Wall time per call via HIP events."""
import ctypes as C
import json
import os
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from clover_amd.lib_binding import THRESHOLD_REFERENCE, CloverHip  # noqa: E402

hip = CloverHip(path=os.environ.get("CLV_LIB"))
lib = hip.lib
vp = C.c_void_p
res = {}
rng = np.random.default_rng(1)
CASES = [tuple(int(v) for v in c.split("x")) for c in os.environ["TRP_CASES"].split(",")] if os.environ.get("TRP_CASES") else ((8192, 1024), (1152, 1024), (8192, 64), (8192, 4096), (65536, 8192), (8320, 8192))
for n, k in CASES:
    q = hip.alloc(n // 2)
    s = hip.alloc(n // 16)
    qq = hip.alloc(n // 2)
    hip.check(lib.clv_fill_random_nibbles(q.ptr, q.nbytes, 5, 0, None))
    hip.check(lib.clv_fill_random_scales(s.ptr, s.nbytes // 4, 6, 0, None))
    a, b = vp(), vp()
    hip.check(lib.clv_event_create(C.byref(a)))
    hip.check(lib.clv_event_create(C.byref(b)))
    ts = []
    for _ in range(5):
        hip.check(lib.clv_memcpy_d2d(qq.ptr, q.ptr, n // 2, None))
        hip.check(lib.clv_event_record(a, None))
        hip.check(lib.clv4_threshold_mode(qq.ptr, s.ptr, n, n, k, THRESHOLD_REFERENCE, None, None))
        hip.check(lib.clv_event_record(b, None))
        hip.check(lib.clv_event_sync(b))
        ms = C.c_float()
        hip.check(lib.clv_event_elapsed_ms(a, b, C.byref(ms)))
        ts.append(ms.value)
    res[f"n{n}_k{k}"] = round(sorted(ts)[2] * 1e3, 1)
print(json.dumps(res))
