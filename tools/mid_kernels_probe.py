#!/usr/bin/env python3
"""The streaming kernels between 0.64 and 0.75 of the HBM peak (transpose, stochastic matrix quantize, 4b x 8b mvm, stochastic 8-bit quantize,
large-n threshold) a few times each at HBM-resident sizes, for rocprofv3 --pmc passes (tools/mid_kernels_pmc.sh)."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from clover_amd.lib_binding import CloverHip  # noqa: E402

hip = CloverHip()
lib = hip.lib
M = N = 32768
A = hip.alloc(4 * M * N)
hip.check(lib.clv_fill_random_ints_f32(A.ptr, M * N, 10, 6, 0, None))
qA, sA = hip.alloc(M * N // 2), hip.alloc((M // 64) * (N // 64) * 4)
qT, sT = hip.alloc(M * N // 2), hip.alloc((M // 64) * (N // 64) * 4)
rng = hip.new_rng(3, 4)
for _ in range(3):
    hip.check(lib.clm4_quantize(A.ptr, M, N, qA.ptr, sA.ptr, None, None))
    hip.check(lib.clm4_quantize(A.ptr, M, N, qT.ptr, sT.ptr, rng.ptr, None))
    hip.check(lib.clm4_transpose(qA.ptr, sA.ptr, M, N, qT.ptr, sT.ptr, None))
x8, sx, r8, sr = hip.alloc(N), hip.alloc(N // 16), hip.alloc(M), hip.alloc(M // 16)
hip.check(lib.clv_fill_random_nibbles(x8.ptr, x8.nbytes, 11, 0, None))
hip.check(lib.clv_fill_random_scales(sx.ptr, sx.nbytes // 4, 10, 0, None))
for _ in range(3):
    hip.check(lib.clm4_mvm_v8(qA.ptr, sA.ptr, M, N, x8.ptr, sx.ptr, r8.ptr, sr.ptr, None, None))
del A
n = 1 << 30
x = hip.alloc(4 * n)
hip.check(lib.clv_fill_random_ints_f32(x.ptr, n, 10, 5, 0, None))
q8, s8 = hip.alloc(n), hip.alloc(n // 16)
for _ in range(3):
    hip.check(lib.clv8_quantize(x.ptr, n, q8.ptr, s8.ptr, rng.ptr, None))
    hip.check(lib.clv8_quantize(x.ptr, n, q8.ptr, s8.ptr, None, None))
    hip.check(lib.clv4_quantize(x.ptr, n, q8.ptr, s8.ptr, rng.ptr, None))
hip.sync()
print("mid kernels probe done")
