#!/usr/bin/env python3
"""Kernel-variant sweep on the GPU box (not part of the product): read-bandwidth ceiling and mvm variants."""
import ctypes as C
import json
import os
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from clover_amd.build import build_probe_library  # noqa: E402
from clover_amd.lib_binding import CloverHip  # noqa: E402

hip = CloverHip(path=build_probe_library(), allow_probe=True)      # clvx_* live in the bench-only probe build
lib = hip.lib
vp, u64 = C.c_void_p, C.c_uint64
lib.clvx_read_bw.argtypes = [vp, u64, C.c_int, C.c_int, vp, vp]
lib.clvx_mvm_variant.argtypes = [C.c_int, vp, vp, u64, u64, vp, vp, vp, vp, vp]


def timeit(fn, reps=20, warm=3):
    for _ in range(warm):
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
    return sorted(ts)[len(ts) // 2]


# MB_SIZES="4096x8192,8192x4096" MB_VARIANTS="1,5,6" MB_SKIP_READ=1 narrow the sweep
SIZES = [tuple(int(v) for v in t.split("x")) for t in os.environ["MB_SIZES"].split(",")] if os.environ.get("MB_SIZES") else \
    [(65536, 65536), (32768, 32768), (16384, 16384), (8192, 8192), (131072, 32768), (16384, 131072)]
VARIANTS = [int(v) for v in os.environ["MB_VARIANTS"].split(",")] if os.environ.get("MB_VARIANTS") else list(range(5))
res = {}
big = hip.alloc(2 << 30)
hip.check(lib.clv_fill_random_nibbles(big.ptr, big.nbytes, 1, 0, None))
out = hip.alloc(256)
for nt in (() if os.environ.get("MB_SKIP_READ") else (0, 1)):
    for bpc in (4, 8, 16):
        ms = timeit(lambda: hip.check(lib.clvx_read_bw(big.ptr, big.nbytes, nt, bpc, out.ptr, None)))
        res[f"read_bw_2GiB_nt{nt}_bpc{bpc}"] = round(big.nbytes / ms / 1e6, 1)


def mvm_bytes(rows, cols):
    return rows * cols // 2 + 4 * (rows // 64) * (cols // 64) + (cols // 2 + cols // 16) + (rows // 2 + rows // 16)


for (rows, cols) in SIZES:
    sA = hip.alloc((rows // 64) * (cols // 64) * 4)
    x, sx = hip.alloc(cols // 2), hip.alloc(cols // 16)
    r, sr = hip.alloc(rows // 2), hip.alloc(rows // 16)
    hip.check(lib.clv_fill_random_scales(sA.ptr, sA.nbytes // 4, 2, 0, None))
    hip.check(lib.clv_fill_random_nibbles(x.ptr, x.nbytes, 3, 0, None))
    hip.check(lib.clv_fill_random_scales(sx.ptr, sx.nbytes // 4, 4, 0, None))
    ref = None
    # MB_COLD=1: every call takes the NEXT of the distinct matrices that fit the 2 GiB buffer (>= 768 MiB of them: beyond the Infinity Cache),
    # so the matrix streams from HBM as it does the first time an application touches it
    nmat = max(1, min(big.nbytes // (rows * cols // 2), -(-(768 << 20) // (rows * cols // 2)))) if os.environ.get("MB_COLD") else 1
    turn = [0]
    for v in VARIANTS:
        def fn(v=v):
            off = (turn[0] % nmat) * (rows * cols // 2)
            turn[0] += 1
            hip.check(lib.clvx_mvm_variant(v, big.ptr + off, sA.ptr, rows, cols, x.ptr, sx.ptr, r.ptr, sr.ptr, None))
        ms = timeit(fn, reps=max(20, nmat))
        got = (r.download(np.uint8).tobytes(), sr.download(np.float32).tobytes())
        if ref is None:
            ref = got
        res[f"mvm_{rows}x{cols}_v{v}" + ("_cold" if nmat > 1 else "")] = {"us": round(ms * 1e3, 2), "GB/s": round(mvm_bytes(rows, cols) / ms / 1e6, 1),
                                                                         "same_as_v0": got == ref if nmat == 1 else None}
print(json.dumps(res, indent=1))
