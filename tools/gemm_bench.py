#!/usr/bin/env python3
"""GEMM timing at 8192^3 / 4096^3 (GB_SIZES=... for others; CLV_GEMM_KERNEL=i8 for the int8-MFMA kernel)."""
import ctypes as C
import os
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from clover_amd.lib_binding import CloverHip  # noqa: E402

# GB_LIB=probe: the bench-only build with the loop's timing-only variants (CLV_GEMM_LOOP=vN; clover_amd/build.py build_probe_library)
from clover_amd.build import probe_library_path  # noqa: E402

hip = CloverHip(path=probe_library_path() if os.environ.get("GB_LIB") == "probe" else None, allow_probe=os.environ.get("GB_LIB") == "probe")
lib = hip.lib
for G in [int(g) for g in os.environ.get("GB_SIZES", "4096,8192").split(",")]:
    A, B = hip.alloc(G * G // 2), hip.alloc(G * G // 2)
    sA, sB = hip.alloc((G // 64) ** 2 * 4), hip.alloc((G // 64) ** 2 * 4)
    Cc = hip.alloc(G * G * 4)
    for t, sd in ((A, 1), (B, 2)):
        hip.check(lib.clv_fill_random_nibbles(t.ptr, t.nbytes, sd, 0, None))
    for t, sd in ((sA, 3), (sB, 4)):
        hip.check(lib.clv_fill_random_scales(t.ptr, t.nbytes // 4, sd, 0, None))
    mode = os.environ.get("GB_MODE", "gemm")          # gemm | i32 (exact int32 GEMM, no scales) | prepared (both FP6 images cached) | preparedB
    fn = lambda: hip.check(lib.clm4_gemm(A.ptr, sA.ptr, G, G, B.ptr, sB.ptr, G, Cc.ptr, None))
    if mode == "i32":
        fn = lambda: hip.check(lib.clm4_gemm_i32(A.ptr, G, G, B.ptr, G, 0, G // 64, Cc.ptr, None))
    elif mode == "i32prepared":                         # the plain FP6 GEMM: exact int32 sums of ALL K-blocks from cached FP6 images, no fold
        opA, opB = C.c_void_p(), C.c_void_p()
        hip.check(lib.clm4_gemm_prepare(A.ptr, G, G, C.byref(opA), None))
        hip.check(lib.clm4_gemm_prepare(B.ptr, G, G, C.byref(opB), None))
        fn = lambda: hip.check(lib.clm4_gemm_i32_prepared(opA, None, G, G, opB, None, G, 0, G // 64, Cc.ptr, None))
    elif mode.startswith("prepared"):
        opA, opB = C.c_void_p(), C.c_void_p()
        if mode == "prepared":
            hip.check(lib.clm4_gemm_prepare(A.ptr, G, G, C.byref(opA), None))
        hip.check(lib.clm4_gemm_prepare(B.ptr, G, G, C.byref(opB), None))
        fn = lambda: hip.check(lib.clm4_gemm_prepared(opA, None if opA else A.ptr, sA.ptr, G, G, opB, None, sB.ptr, G, Cc.ptr, None))
    for _ in range(int(os.environ.get("GB_WARM", "80"))):     # the clocks settle after ~50 calls (profiles/r02_gemm_warmup_series.txt)
        fn()
    hip.sync()
    a, b = C.c_void_p(), C.c_void_p()
    hip.check(lib.clv_event_create(C.byref(a)))
    hip.check(lib.clv_event_create(C.byref(b)))
    ts = []
    for _ in range(5):
        hip.check(lib.clv_event_record(a, None))
        for _ in range(5):
            fn()
        hip.check(lib.clv_event_record(b, None))
        hip.check(lib.clv_event_sync(b))
        ms = C.c_float()
        hip.check(lib.clv_event_elapsed_ms(a, b, C.byref(ms)))
        ts.append(ms.value / 5)
    ms = sorted(ts)[2]
    chk = float(np.abs(Cc.download(np.float32, 4096)).sum())
    print(f"mode={mode} kernel={os.environ.get('CLV_GEMM_KERNEL', 'fp6')} G={G} {ms:.4f} ms {2.0 * G ** 3 / ms / 1e9:.1f} TOP/s checksum={chk:.6e}")
