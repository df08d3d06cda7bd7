#!/usr/bin/env python3
"""Per-kernel achieved bandwidth at HBM-resident sizes (operands >> 256 MiB Infinity Cache), printed as JSON.

Algorithmic bytes per element follow SURVEY 8(d): quantize/restore 4.5625, dot 1.125, scaleAndAdd 1.6875,
threshold 1.125 (nibbles + scales read and written once each), transpose 2 x (1/2 + 4/4096), matrix quantize 4.5625."""
import ctypes as C
import json
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from clover_amd.lib_binding import DOT_EXACT, DOT_FAST, CloverHip  # noqa: E402

hip = CloverHip(path=os.environ.get("CLV_LIB"))      # CLV_LIB: another build of the library, for same-box A/B runs
lib = hip.lib
vp = C.c_void_p


def timeit(fn, reps=10, rounds=5, warm=2):
    for _ in range(warm):
        fn()
    hip.sync()
    a, b = vp(), vp()
    hip.check(lib.clv_event_create(C.byref(a)))
    hip.check(lib.clv_event_create(C.byref(b)))
    ts = []
    for _ in range(rounds):
        hip.check(lib.clv_event_record(a, None))
        for _ in range(reps):
            fn()
        hip.check(lib.clv_event_record(b, None))
        hip.check(lib.clv_event_sync(b))
        ms = C.c_float()
        hip.check(lib.clv_event_elapsed_ms(a, b, C.byref(ms)))
        ts.append(ms.value / reps)
    return sorted(ts)[len(ts) // 2]


res = {}


ONLY = os.environ.get("KB_ONLY")      # substring filter: KB_ONLY=scale_and_add python tools/kernel_bench.py


def rec(name, nbytes, fn, extra=None, reps=10):
    if ONLY and ONLY not in name:
        return
    ms = timeit(fn, reps=reps)
    res[name] = {"ms": round(ms, 5), "GB/s": round(nbytes / ms / 1e6, 1), "frac_of_8TBs": round(nbytes / ms / 1e6 / 8000.0, 4)}
    if extra:
        res[name].update(extra)


# ---- vector ops at n = 2^30 (4 GiB fp32 source, 512 MiB + 64 MiB quantized) and n = 2^24
for logn in (24, 30):
    n = 1 << logn
    x = hip.alloc(4 * n)
    hip.check(lib.clv_fill_random_ints_f32(x.ptr, n, 10, 5, 0, None))
    q, s = hip.alloc(n // 2), hip.alloc(n // 16)
    q2, s2 = hip.alloc(n // 2), hip.alloc(n // 16)
    q3, s3 = hip.alloc(n // 2), hip.alloc(n // 16)
    out = hip.alloc(8)
    rng = hip.new_rng(1, 2)
    hip.check(lib.clv4_quantize(x.ptr, n, q.ptr, s.ptr, None, None))         # operands exist even when KB_ONLY skips the timed calls
    hip.check(lib.clv4_quantize(x.ptr, n, q2.ptr, s2.ptr, None, None))
    rec(f"quantize_n2^{logn}", 4.5625 * n, lambda: hip.check(lib.clv4_quantize(x.ptr, n, q.ptr, s.ptr, None, None)))
    rec(f"quantize_stochastic_n2^{logn}", 4.5625 * n, lambda: hip.check(lib.clv4_quantize(x.ptr, n, q2.ptr, s2.ptr, rng.ptr, None)))
    rec(f"dot_fast_n2^{logn}", 1.125 * n, lambda: hip.check(lib.clv4_dot(q.ptr, s.ptr, q2.ptr, s2.ptr, n, DOT_FAST, out.ptr, None, None)))
    rec(f"scale_and_add_n2^{logn}", 1.6875 * n, lambda: hip.check(lib.clv4_scale_and_add(q.ptr, s.ptr, q2.ptr, s2.ptr, 0.5, n, q3.ptr, s3.ptr, None, None)))
    rec(f"scale_and_add_stochastic_n2^{logn}", 1.6875 * n, lambda: hip.check(lib.clv4_scale_and_add(q.ptr, s.ptr, q2.ptr, s2.ptr, 0.5, n, q3.ptr, s3.ptr, rng.ptr, None)))
    q8 = hip.alloc(n)
    rec(f"v8_quantize_n2^{logn}", 5.0625 * n, lambda: hip.check(lib.clv8_quantize(x.ptr, n, q8.ptr, s3.ptr, None, None)))
    rec(f"v8_quantize_stochastic_n2^{logn}", 5.0625 * n, lambda: hip.check(lib.clv8_quantize(x.ptr, n, q8.ptr, s3.ptr, rng.ptr, None)))
    rec(f"restore_n2^{logn}", 4.5625 * n, lambda: hip.check(lib.clv4_restore(q.ptr, s.ptr, n, x.ptr, None)))
    if logn == 24:
        k = n // 4
        hip.check(lib.clv_memcpy_d2d(q3.ptr, q.ptr, n // 2, None))
        thr_note = ("three launches (round 6): the pass over the nibbles builds per-block magnitude tables and radix level 0, one persistent launch "
                    "runs levels 1-2 and the tie prefixes over the tables, one pass applies on bit planes (threshold4_large, threshold4.hip); bytes = "
                    "nibbles + scales read and written once each (1.125 n; rounds 3-5 counted five passes here), time = calls back to back on one "
                    "stream incl. launch gaps (kernel time alone: profiles/r06_threshold_three_launch.txt)")
        rec(f"threshold_k25pct_n2^{logn}", 1.125 * n, lambda: hip.check(lib.clv4_threshold(q3.ptr, s.ptr, n, n, k, None, None)), reps=3,
            extra={"note": thr_note})
        nb = 1 << 28
        qb, sb = hip.alloc(nb // 2), hip.alloc(nb // 16)
        hip.check(lib.clv_fill_random_nibbles(qb.ptr, qb.nbytes, 7, 0, None))
        hip.check(lib.clv_fill_random_scales(sb.ptr, nb // 64, 8, 0, None))
        rec("threshold_k25pct_n2^28", 1.125 * nb, lambda: hip.check(lib.clv4_threshold(qb.ptr, sb.ptr, nb, nb, nb // 4, None, None)), reps=3,
            extra={"note": thr_note + "; the vector is thresholded in place, so every call after the first finds it already thresholded (same passes, same time)"})
        del qb, sb
        rec(f"dot_exact_n2^{logn}", 1.125 * n, lambda: hip.check(lib.clv4_dot(q.ptr, s.ptr, q2.ptr, s2.ptr, n, DOT_EXACT, out.ptr, None, None)), reps=2)
    q8b = hip.alloc(n)
    hip.check(lib.clv8_quantize(x.ptr, n, q8.ptr, s3.ptr, None, None))
    hip.check(lib.clv8_quantize(x.ptr, n, q8b.ptr, s2.ptr, None, None))
    rec(f"v8_dot_fast_n2^{logn}", 2.125 * n, lambda: hip.check(lib.clv8_dot(q8.ptr, s3.ptr, q8b.ptr, s2.ptr, n, DOT_FAST, out.ptr, None, None)))
    if logn == 24:
        rec(f"v8_dot_exact_n2^{logn}", 2.125 * n, lambda: hip.check(lib.clv8_dot(q8.ptr, s3.ptr, q8b.ptr, s2.ptr, n, DOT_EXACT, out.ptr, None, None)), reps=2)
    q8c = hip.alloc(n)
    rec(f"v8_scale_and_add_n2^{logn}", 3.1875 * n, lambda: hip.check(lib.clv8_scale_and_add(q8.ptr, s3.ptr, q8b.ptr, s2.ptr, 0.5, n, q8c.ptr, s.ptr, None, None)))
    rec(f"v8_scale_and_add_stochastic_n2^{logn}", 3.1875 * n, lambda: hip.check(lib.clv8_scale_and_add(q8.ptr, s3.ptr, q8b.ptr, s2.ptr, 0.5, n, q8c.ptr, s.ptr, rng.ptr, None)))
    del q8c, q8b
    rec(f"v8_restore_n2^{logn}", 5.0625 * n, lambda: hip.check(lib.clv8_restore(q8.ptr, s3.ptr, n, x.ptr, None)))
    del x, q, s, q2, s2, q3, s3, q8

# ---- matrix ops: 32768 x 32768 (4 GiB fp32 source, 512 MiB quantized)
M = N = 32768
A = hip.alloc(4 * M * N)
hip.check(lib.clv_fill_random_ints_f32(A.ptr, M * N, 10, 6, 0, None))
qA, sA = hip.alloc(M * N // 2), hip.alloc((M // 64) * (N // 64) * 4)
qT, sT = hip.alloc(M * N // 2), hip.alloc((M // 64) * (N // 64) * 4)
rngm = hip.new_rng(3, 4)
rec("matrix_quantize_32768^2", 4.5625 * M * N, lambda: hip.check(lib.clm4_quantize(A.ptr, M, N, qA.ptr, sA.ptr, None, None)), reps=3)
rec("matrix_quantize_stochastic_32768^2", 4.5625 * M * N, lambda: hip.check(lib.clm4_quantize(A.ptr, M, N, qT.ptr, sT.ptr, rngm.ptr, None)), reps=3)
rec("transpose_32768^2", 2 * (M * N // 2 + 4 * (M // 64) * (N // 64)), lambda: hip.check(lib.clm4_transpose(qA.ptr, sA.ptr, M, N, qT.ptr, sT.ptr, None)), reps=3)
# shapes whose rows / cols are 128 * odd: edge tiles are masked on the same kernel (r6; the 64-thread kernel they used to fall to is gone)
for (Mr, Nr) in ((32640, 32640), (32768, 32640), (32640, 32768)):
    rec(f"transpose_{Mr}x{Nr}", 2 * (Mr * Nr // 2 + 4 * (Mr // 64) * (Nr // 64)),
        lambda Mr=Mr, Nr=Nr: hip.check(lib.clm4_transpose(qA.ptr, sA.ptr, Mr, Nr, qT.ptr, sT.ptr, None)), reps=3)
x, sx = hip.alloc(N // 2), hip.alloc(N // 16)
r, sr = hip.alloc(M // 2), hip.alloc(M // 16)
hip.check(lib.clv_fill_random_nibbles(x.ptr, x.nbytes, 9, 0, None))
hip.check(lib.clv_fill_random_scales(sx.ptr, sx.nbytes // 4, 10, 0, None))
mvb = M * N // 2 + 4 * (M // 64) * (N // 64) + (N // 2 + N // 16) + (M // 2 + M // 16)
rec("mvm_32768^2", mvb, lambda: hip.check(lib.clm4_mvm(qA.ptr, sA.ptr, M, N, x.ptr, sx.ptr, r.ptr, sr.ptr, None, None)))
rec("mvm_stochastic_32768^2", mvb, lambda: hip.check(lib.clm4_mvm(qA.ptr, sA.ptr, M, N, x.ptr, sx.ptr, r.ptr, sr.ptr, rngm.ptr, None)))
# ---- mixed precision: 4-bit matrix x 8-bit vector
x8, r8 = hip.alloc(N), hip.alloc(M)
hip.check(lib.clv_fill_random_nibbles(x8.ptr, x8.nbytes, 11, 0, None))          # any bytes; 0x80 never occurs (nibbles are in [-7,7])
mvb8 = M * N // 2 + 4 * (M // 64) * (N // 64) + (N + N // 16) + (M + M // 16)
rec("mvm_v8_32768^2", mvb8, lambda: hip.check(lib.clm4_mvm_v8(qA.ptr, sA.ptr, M, N, x8.ptr, sx.ptr, r8.ptr, sr.ptr, None, None)))
rec("mvm_v8_stochastic_32768^2", mvb8, lambda: hip.check(lib.clm4_mvm_v8(qA.ptr, sA.ptr, M, N, x8.ptr, sx.ptr, r8.ptr, sr.ptr, rngm.ptr, None)))
# ---- mixed precision: 4-bit matrix x fp32 vector (fp32 row dots out)
xf32, rf32 = hip.alloc(4 * N), hip.alloc(4 * M)
hip.check(lib.clv_fill_random_ints_f32(xf32.ptr, N, 10, 12, 0, None))
rec("mvm_f32_32768^2", M * N // 2 + 4 * (M // 64) * (N // 64) + 4 * N + 4 * M,
    lambda: hip.check(lib.clm4_mvm_f32(qA.ptr, sA.ptr, M, N, xf32.ptr, rf32.ptr, None)))
# ---- the headline shape, 65536 x 65536 (2 GiB of nibbles): 4-bit and mixed mvm side by side
del A, qT, sT
M2 = N2 = 65536
big, sbig = hip.alloc(M2 * N2 // 2), hip.alloc((M2 // 64) * (N2 // 64) * 4)
hip.check(lib.clv_fill_random_nibbles(big.ptr, big.nbytes, 21, 0, None))
hip.check(lib.clv_fill_random_scales(sbig.ptr, sbig.nbytes // 4, 22, 0, None))
xb4, xb8, sxb = hip.alloc(N2 // 2), hip.alloc(N2), hip.alloc(N2 // 16)
rb, srb = hip.alloc(M2), hip.alloc(M2 // 16)
hip.check(lib.clv_fill_random_nibbles(xb8.ptr, xb8.nbytes, 23, 0, None))
hip.check(lib.clv_fill_random_nibbles(xb4.ptr, xb4.nbytes, 24, 0, None))
hip.check(lib.clv_fill_random_scales(sxb.ptr, sxb.nbytes // 4, 25, 0, None))
base = M2 * N2 // 2 + 4 * (M2 // 64) * (N2 // 64)
rec("mvm_65536^2", base + (N2 // 2 + N2 // 16) + (M2 // 2 + M2 // 16),
    lambda: hip.check(lib.clm4_mvm(big.ptr, sbig.ptr, M2, N2, xb4.ptr, sxb.ptr, rb.ptr, srb.ptr, None, None)))
rec("mvm_v8_65536^2", base + (N2 + N2 // 16) + (M2 + M2 // 16),
    lambda: hip.check(lib.clm4_mvm_v8(big.ptr, sbig.ptr, M2, N2, xb8.ptr, sxb.ptr, rb.ptr, srb.ptr, None, None)))
print(json.dumps(res, indent=1))
