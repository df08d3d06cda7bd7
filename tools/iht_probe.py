#!/usr/bin/env python3
"""A few IHT iterations at N = 8192 (for rocprofv3 passes).  IHT_ITERS (default 20) iterations, IHT_K (default m / 4) survivors."""
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from clover_amd.lib_binding import CloverHip  # noqa: E402

hip = CloverHip()
lib = hip.lib
m, n = 4096, 8192
ITERS = int(os.environ.get("IHT_ITERS", "20"))
KK = int(os.environ.get("IHT_K", str(m // 4)))
Phi, PhiT = hip.alloc(m * n // 2), hip.alloc(m * n // 2)
sPhi, sPhiT = hip.alloc((m // 64) * (n // 64) * 4), hip.alloc((m // 64) * (n // 64) * 4)
hip.check(lib.clv_fill_random_nibbles(Phi.ptr, Phi.nbytes, 31, 0, None))
hip.check(lib.clv_fill_random_scales(sPhi.ptr, sPhi.nbytes // 4, 32, 0, None))
hip.check(lib.clm4_transpose(Phi.ptr, sPhi.ptr, m, n, PhiT.ptr, sPhiT.ptr, None))


def vec(k, sd):
    q, s = hip.alloc(k // 2), hip.alloc(k // 16)
    hip.check(lib.clv_fill_random_nibbles(q.ptr, q.nbytes, sd, 0, None))
    hip.check(lib.clv_fill_random_scales(s.ptr, s.nbytes // 4, sd + 1, 0, None))
    return q, s


x, y, t1, t2, t3 = vec(n, 41), vec(m, 43), vec(m, 45), vec(m, 47), vec(n, 49)
if len(sys.argv) > 1 and sys.argv[1] in ("v8", "v8st"):   # the mixed configuration: 8-bit vectors ("v8st": stochastic rounding)
    def vec8(k, sd):
        q, s = hip.alloc(k), hip.alloc(k // 16)
        hip.check(lib.clv_fill_random_nibbles(q.ptr, q.nbytes, sd, 0, None))
        hip.check(lib.clv_fill_random_scales(s.ptr, s.nbytes // 4, sd + 1, 0, None))
        return q, s
    x, y, t1, t2, t3 = vec8(n, 41), vec8(m, 43), vec8(m, 45), vec8(m, 47), vec8(n, 49)
    rng8 = hip.new_rng(5, 6) if sys.argv[1] == "v8st" else None
    hip.check(lib.clm4_iht_v8(Phi.ptr, sPhi.ptr, PhiT.ptr, sPhiT.ptr, m, n, x[0].ptr, x[1].ptr, n, y[0].ptr, y[1].ptr, t1[0].ptr, t1[1].ptr,
                              t2[0].ptr, t2[1].ptr, t3[0].ptr, t3[1].ptr, ITERS, KK, 1e-3, 1, rng8.ptr if rng8 else None, None))
    hip.sync()
    print("iht v8 probe done")
    sys.exit(0)
rng = hip.new_rng(5, 6) if len(sys.argv) > 1 and sys.argv[1] == "st" else None      # "st": stochastic rounding
hip.check(lib.clm4_iht(Phi.ptr, sPhi.ptr, PhiT.ptr, sPhiT.ptr, m, n, x[0].ptr, x[1].ptr, n, y[0].ptr, y[1].ptr, t1[0].ptr, t1[1].ptr,
                       t2[0].ptr, t2[1].ptr, t3[0].ptr, t3[1].ptr, ITERS, KK, 1e-3, 1, rng.ptr if rng else None, None))
hip.sync()
print("iht probe done")
