#!/usr/bin/env python3
"""clv4_threshold at the one-workgroup sizes (IHT / GD vectors), a few calls per size, for rocprofv3 --kernel-trace --stats."""
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from clover_amd.lib_binding import CloverHip  # noqa: E402

hip = CloverHip(path=os.environ.get("CLV_LIB"))      # CLV_LIB: another build of the library, for same-box A/B runs
lib = hip.lib
for n in [int(v) for v in os.environ.get("TP_N", "8192,32768,131072").split(",")]:
    q, s = hip.alloc(n // 2), hip.alloc(n // 16)
    for _ in range(int(os.environ.get("TP_CALLS", "20"))):
        hip.check(lib.clv_fill_random_nibbles(q.ptr, q.nbytes, 7, 0, None))
        hip.check(lib.clv_fill_random_scales(s.ptr, n // 64, 8, 0, None))
        hip.check(lib.clv4_threshold(q.ptr, s.ptr, n, n, n // 8, None, None))
    hip.sync()
print("threshold small probe done")
