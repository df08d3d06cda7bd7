#!/usr/bin/env python3
"""Same buffer, same process, alternating: vector quantize of 2^30 floats vs matrix quantize of it as 32768 x 32768."""
import ctypes as C
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from clover_amd.lib_binding import CloverHip  # noqa: E402

hip = CloverHip()
lib = hip.lib
n = 1 << 30
x = hip.alloc(4 * n)
hip.check(lib.clv_fill_random_ints_f32(x.ptr, n, 10, 5, 0, None))
q, s = hip.alloc(n // 2), hip.alloc(n // 16)
a, b = C.c_void_p(), C.c_void_p()
hip.check(lib.clv_event_create(C.byref(a)))
hip.check(lib.clv_event_create(C.byref(b)))


def t(fn, reps=5):
    fn()
    hip.check(lib.clv_event_record(a, None))
    for _ in range(reps):
        fn()
    hip.check(lib.clv_event_record(b, None))
    hip.check(lib.clv_event_sync(b))
    ms = C.c_float()
    hip.check(lib.clv_event_elapsed_ms(a, b, C.byref(ms)))
    return ms.value / reps


vec = lambda: hip.check(lib.clv4_quantize(x.ptr, n, q.ptr, s.ptr, None, None))
mat = lambda: hip.check(lib.clm4_quantize(x.ptr, 32768, 32768, q.ptr, s.ptr, None, None))
for r in range(4):
    print(f"round {r}: vector {t(vec):.4f} ms   matrix {t(mat):.4f} ms")
