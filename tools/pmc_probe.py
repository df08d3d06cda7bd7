#!/usr/bin/env python3
"""Launches a calibration read (known 2 GiB, 16 B/lane coalesced) and the C3 mvm a few times each, for
rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE need separate passes; see MI355X_MICROARCH.md, HBM)."""
import ctypes as C
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from clover_amd.build import build_probe_library  # noqa: E402
from clover_amd.lib_binding import CloverHip  # noqa: E402

hip = CloverHip(path=build_probe_library(), allow_probe=True)      # clvx_* live in the bench-only probe build
lib = hip.lib
vp, u64 = C.c_void_p, C.c_uint64
lib.clvx_read_bw.argtypes = [vp, u64, C.c_int, C.c_int, vp, vp]
rows = cols = 65536
A = hip.alloc(rows * cols // 2)
sA = hip.alloc((rows // 64) * (cols // 64) * 4)
x, sx = hip.alloc(cols // 2), hip.alloc(cols // 16)
r, sr = hip.alloc(rows // 2), hip.alloc(rows // 16)
out = hip.alloc(256)
hip.check(lib.clv_fill_random_nibbles(A.ptr, A.nbytes, 1, 0, None))
hip.check(lib.clv_fill_random_scales(sA.ptr, sA.nbytes // 4, 2, 0, None))
hip.check(lib.clv_fill_random_nibbles(x.ptr, x.nbytes, 3, 0, None))
hip.check(lib.clv_fill_random_scales(sx.ptr, sx.nbytes // 4, 4, 0, None))
for _ in range(3):
    hip.check(lib.clvx_read_bw(A.ptr, A.nbytes, 1, 16, out.ptr, None))
for _ in range(5):
    hip.check(lib.clm4_mvm(A.ptr, sA.ptr, rows, cols, x.ptr, sx.ptr, r.ptr, sr.ptr, None, None))
# the other streaming kernels at n = 2^30 / 32768^2, three launches each (same passes, traffic vs algorithmic bytes)
n = 1 << 30
xf = hip.alloc(4 * n)
hip.check(lib.clv_fill_random_ints_f32(xf.ptr, n, 10, 5, 0, None))
q4, s4, q8 = hip.alloc(n // 2), hip.alloc(n // 16), hip.alloc(n)
x8 = hip.alloc(cols)
hip.check(lib.clv_fill_random_nibbles(x8.ptr, x8.nbytes, 9, 0, None))
r8 = hip.alloc(rows)
for _ in range(3):
    hip.check(lib.clv4_quantize(xf.ptr, n, q4.ptr, s4.ptr, None, None))
    hip.check(lib.clm4_quantize(xf.ptr, 32768, 32768, q4.ptr, s4.ptr, None, None))
    hip.check(lib.clv8_quantize(xf.ptr, n, q8.ptr, s4.ptr, None, None))
    hip.check(lib.clv4_restore(q4.ptr, s4.ptr, n, xf.ptr, None))
    hip.check(lib.clm4_mvm_v8(A.ptr, sA.ptr, rows, cols, x8.ptr, sx.ptr, r8.ptr, sr.ptr, None, None))
hip.sync()
print("pmc probe done")
