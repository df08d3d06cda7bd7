#!/usr/bin/env python3
"""Launches the 8192^3 GEMM GP_CALLS times (default 3: counter passes; the kernel-trace pass uses 150 so that the average is the
steady-state one, see profiles/r02_gemm_warmup_series.txt)."""
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from clover_amd.lib_binding import CloverHip  # noqa: E402

hip = CloverHip()
lib = hip.lib
G = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
A, B = hip.alloc(G * G // 2), hip.alloc(G * G // 2)
sA, sB = hip.alloc((G // 64) ** 2 * 4), hip.alloc((G // 64) ** 2 * 4)
Cc = hip.alloc(G * G * 4)
for t, sd in ((A, 1), (B, 2)):
    hip.check(lib.clv_fill_random_nibbles(t.ptr, t.nbytes, sd, 0, None))
for t, sd in ((sA, 3), (sB, 4)):
    hip.check(lib.clv_fill_random_scales(t.ptr, t.nbytes // 4, sd, 0, None))
for _ in range(int(os.environ.get("GP_CALLS", "3"))):
    hip.check(lib.clm4_gemm(A.ptr, sA.ptr, G, G, B.ptr, sB.ptr, G, Cc.ptr, None))
hip.sync()
print("gemm probe done")
