#!/usr/bin/env python3
"""The LDS-heavy kernels once each at the IHT size (N = 8192) and a few others, for a rocprofv3 --pmc pass over SQ_LDS_BANK_CONFLICT /
SQ_LDS_IDX_ACTIVE (tools/lds_conflict.sh): which kernel loses LDS cycles to bank conflicts."""
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from clover_amd.lib_binding import CloverHip  # noqa: E402

hip = CloverHip()
lib = hip.lib
m, n = 4096, 8192
Phi, PhiT = hip.alloc(m * n // 2), hip.alloc(m * n // 2)
sPhi, sPhiT = hip.alloc((m // 64) * (n // 64) * 4), hip.alloc((m // 64) * (n // 64) * 4)
hip.check(lib.clv_fill_random_nibbles(Phi.ptr, Phi.nbytes, 31, 0, None))
hip.check(lib.clv_fill_random_scales(sPhi.ptr, sPhi.nbytes // 4, 32, 0, None))
hip.check(lib.clm4_transpose(Phi.ptr, sPhi.ptr, m, n, PhiT.ptr, sPhiT.ptr, None))


def vec(k, sd, bytes_per_elem_x2):
    q, s = hip.alloc(k * bytes_per_elem_x2 // 2), hip.alloc(k // 16)
    hip.check(lib.clv_fill_random_nibbles(q.ptr, q.nbytes, sd, 0, None))
    hip.check(lib.clv_fill_random_scales(s.ptr, s.nbytes // 4, sd + 1, 0, None))
    return q, s


for bits in (4, 8):
    b2 = 1 if bits == 4 else 2
    x, y, t1, t2, t3 = vec(n, 41, b2), vec(m, 43, b2), vec(m, 45, b2), vec(m, 47, b2), vec(n, 49, b2)
    fn = lib.clm4_iht if bits == 4 else lib.clm4_iht_v8
    for persistent in ("1", "0"):
        os.environ["CLV_IHT_PERSISTENT"] = persistent
        for rs in (None, hip.new_rng(5, 6)):
            hip.check(fn(Phi.ptr, sPhi.ptr, PhiT.ptr, sPhiT.ptr, m, n, x[0].ptr, x[1].ptr, n, y[0].ptr, y[1].ptr, t1[0].ptr, t1[1].ptr,
                         t2[0].ptr, t2[1].ptr, t3[0].ptr, t3[1].ptr, 20, n // 4, 1e-3, 1, rs.ptr if rs else None, None))
    # the standalone vector kernels of the loop
    if bits == 4:
        hip.check(lib.clv4_threshold(x[0].ptr, x[1].ptr, n, n, n // 4, None, None))
        hip.check(lib.clv4_scale_and_add(x[0].ptr, x[1].ptr, t3[0].ptr, t3[1].ptr, 1e-3, n, x[0].ptr, x[1].ptr, None, None))
    else:
        hip.check(lib.clv8_threshold(x[0].ptr, x[1].ptr, n, n, n // 4, None, None))
os.environ.pop("CLV_IHT_PERSISTENT", None)
# large kernels with LDS stages
M = N = 16384
qA, sA = hip.alloc(M * N // 2), hip.alloc((M // 64) * (N // 64) * 4)
qT, sT = hip.alloc(M * N // 2), hip.alloc((M // 64) * (N // 64) * 4)
hip.check(lib.clv_fill_random_nibbles(qA.ptr, qA.nbytes, 7, 0, None))
hip.check(lib.clv_fill_random_scales(sA.ptr, sA.nbytes // 4, 8, 0, None))
hip.check(lib.clm4_transpose(qA.ptr, sA.ptr, M, N, qT.ptr, sT.ptr, None))
xq, xs = vec(N, 51, 1)
rq, rs_ = hip.alloc(M // 2), hip.alloc(M // 16)
hip.check(lib.clm4_mvm(qA.ptr, sA.ptr, M, N, xq.ptr, xs.ptr, rq.ptr, rs_.ptr, None, None))
big = 1 << 26
bq, bs = vec(big, 61, 1)
hip.check(lib.clv4_threshold(bq.ptr, bs.ptr, big, big, big // 4, None, None))
G = 4096
gq, gs = hip.alloc(G * G // 2), hip.alloc((G // 64) * (G // 64) * 4)
hip.check(lib.clv_fill_random_nibbles(gq.ptr, gq.nbytes, 9, 0, None))
hip.check(lib.clv_fill_random_scales(gs.ptr, gs.nbytes // 4, 10, 0, None))
C = hip.alloc(G * G * 4)
hip.check(lib.clm4_gemm(gq.ptr, gs.ptr, G, G, gq.ptr, gs.ptr, G, C.ptr, None))
hip.sync()
print("lds conflict probe done")
