#!/usr/bin/env python3
"""A few clm4_mvm_f32 (4-bit matrix x fp32 vector) and stochastic clv4_scale_and_add calls at HBM-resident sizes, for rocprofv3 --pmc passes
(tools/weak_kernels_pmc.sh): the two streaming kernels that sit at ~0.5 of the HBM peak, to show what they are bound by."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from clover_amd.lib_binding import CloverHip  # noqa: E402

hip = CloverHip()
lib = hip.lib
M = N = 32768
qA, sA = hip.alloc(M * N // 2), hip.alloc((M // 64) * (N // 64) * 4)
hip.check(lib.clv_fill_random_nibbles(qA.ptr, qA.nbytes, 1, 0, None))
hip.check(lib.clv_fill_random_scales(sA.ptr, sA.nbytes // 4, 2, 0, None))
x, r = hip.alloc(4 * N), hip.alloc(4 * M)
hip.check(lib.clv_fill_random_ints_f32(x.ptr, N, 10, 3, 0, None))
for _ in range(5):
    hip.check(lib.clm4_mvm_f32(qA.ptr, sA.ptr, M, N, x.ptr, r.ptr, None))
n = 1 << 30
q1, s1, q2, s2, q3, s3 = hip.alloc(n // 2), hip.alloc(n // 16), hip.alloc(n // 2), hip.alloc(n // 16), hip.alloc(n // 2), hip.alloc(n // 16)
for t, sd in ((q1, 4), (q2, 5)):
    hip.check(lib.clv_fill_random_nibbles(t.ptr, t.nbytes, sd, 0, None))
for t, sd in ((s1, 6), (s2, 7)):
    hip.check(lib.clv_fill_random_scales(t.ptr, t.nbytes // 4, sd, 0, None))
rng = hip.new_rng(1, 2)
for _ in range(5):
    hip.check(lib.clv4_scale_and_add(q1.ptr, s1.ptr, q2.ptr, s2.ptr, 0.5, n, q3.ptr, s3.ptr, rng.ptr, None))
for _ in range(5):                                   # round 5: the deterministic kernel beside it
    hip.check(lib.clv4_scale_and_add(q1.ptr, s1.ptr, q2.ptr, s2.ptr, 0.5, n, q3.ptr, s3.ptr, None, None))
hip.sync()
print("weak kernels probe done")
