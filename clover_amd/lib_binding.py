"""ctypes binding of libclover_hip.so (include/clover_hip.h) for tests and bench.py.

No fallback: if the library is missing or a call fails this raises -- the HIP path is the only path.
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path

import numpy as np

from .build import hip_library_path

DOT_EXACT = 0
DOT_FAST = 1
THRESHOLD_FAST = 0          # radix select, lowest-index ties
THRESHOLD_REFERENCE = 1     # the reference's survivor set (its min-heap walk, CloverVector4.h:1927-1972)

_vp = C.c_void_p
_u64 = C.c_uint64

# name -> (restype, argtypes); everything include/clover_hip.h declares
SIGNATURES = {
    "clv_version": (C.c_char_p, []),
    "clv_last_error": (C.c_char_p, []),
    "clv_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "clv_set_device": (C.c_int, [C.c_int]),
    "clv_get_device": (C.c_int, [C.POINTER(C.c_int)]),
    "clv_device_info": (C.c_int, [C.c_char_p, C.c_int, C.POINTER(C.c_int), C.POINTER(_u64)]),
    "clv_malloc": (C.c_int, [C.POINTER(_vp), _u64]),
    "clv_free": (C.c_int, [_vp]),
    "clv_memset": (C.c_int, [_vp, C.c_int, _u64, _vp]),
    "clv_memcpy_h2d": (C.c_int, [_vp, _vp, _u64, _vp]),
    "clv_memcpy_d2h": (C.c_int, [_vp, _vp, _u64, _vp]),
    "clv_memcpy_d2d": (C.c_int, [_vp, _vp, _u64, _vp]),
    "clv_host_alloc": (C.c_int, [C.POINTER(_vp), _u64]),
    "clv_host_free": (C.c_int, [_vp]),
    "clv_stream_create": (C.c_int, [C.POINTER(_vp)]),
    "clv_stream_destroy": (C.c_int, [_vp]),
    "clv_stream_sync": (C.c_int, [_vp]),
    "clv_device_sync": (C.c_int, []),
    "clv_event_create": (C.c_int, [C.POINTER(_vp)]),
    "clv_event_destroy": (C.c_int, [_vp]),
    "clv_event_record": (C.c_int, [_vp, _vp]),
    "clv_event_sync": (C.c_int, [_vp]),
    "clv_event_elapsed_ms": (C.c_int, [_vp, _vp, C.POINTER(C.c_float)]),
    "clv_rng_seed": (C.c_int, [_vp, _u64, _u64, _vp]),
    "clv_rng_set": (C.c_int, [_vp, C.POINTER(_u64), C.POINTER(_u64), _vp]),
    "clv_rng_get": (C.c_int, [_vp, C.POINTER(_u64), C.POINTER(_u64), _vp]),
    "clv_rng_graph_mode": (C.c_int, [_vp, C.c_int, _vp]),
    "clv_rng_set_segments": (C.c_int, [C.c_int]),
    "clv4_quantize": (C.c_int, [_vp, _u64, _vp, _vp, _vp, _vp]),
    "clv4_restore": (C.c_int, [_vp, _vp, _u64, _vp, _vp]),
    "clv4_dot_workspace_bytes": (_u64, [_u64]),
    "clv4_dot": (C.c_int, [_vp, _vp, _vp, _vp, _u64, C.c_int, _vp, _vp, _vp]),
    "clv8_dot_workspace_bytes": (_u64, [_u64]),
    "clv8_dot": (C.c_int, [_vp, _vp, _vp, _vp, _u64, C.c_int, _vp, _vp, _vp]),
    "clv4_word_isums": (C.c_int, [_vp, _vp, _u64, _vp, _vp]),
    "clm4_quantize": (C.c_int, [_vp, _u64, _u64, _vp, _vp, _vp, _vp]),
    "clm4_restore": (C.c_int, [_vp, _vp, _u64, _u64, _vp, _vp]),
    "clm4_mvm": (C.c_int, [_vp, _vp, _u64, _u64, _vp, _vp, _vp, _vp, _vp, _vp]),
    "clm4_rowdots": (C.c_int, [_vp, _vp, _u64, _u64, _vp, _vp, _vp, _vp]),
    "clm4_gemm": (C.c_int, [_vp, _vp, _u64, _u64, _vp, _vp, _u64, _vp, _vp]),
    "clm4_gemm_prepare": (C.c_int, [_vp, _u64, _u64, C.POINTER(_vp), _vp]),
    "clm4_gemm_release": (C.c_int, [_vp]),
    "clm4_gemm_prepared": (C.c_int, [_vp, _vp, _vp, _u64, _u64, _vp, _vp, _vp, _u64, _vp, _vp]),
    "clm4_gemm_i32": (C.c_int, [_vp, _u64, _u64, _vp, _u64, _u64, _u64, _vp, _vp]),
    "clm4_gemm_i32_prepared": (C.c_int, [_vp, _vp, _u64, _u64, _vp, _vp, _u64, _u64, _u64, _vp, _vp]),
    "clv4_scale_and_add": (C.c_int, [_vp, _vp, _vp, _vp, C.c_float, _u64, _vp, _vp, _vp, _vp]),
    "clm4_mvm_scale_and_add": (C.c_int, [_vp, _vp, _u64, _u64, _vp, _vp, _vp, _vp, C.c_float, _vp, _vp, _vp, _vp, _vp, _vp]),
    "clv8_quantize": (C.c_int, [_vp, _u64, _vp, _vp, _vp, _vp]),
    "clv8_restore": (C.c_int, [_vp, _vp, _u64, _vp, _vp]),
    "clv8_scale_and_add": (C.c_int, [_vp, _vp, _vp, _vp, C.c_float, _u64, _vp, _vp, _vp, _vp]),
    "clv8_threshold_workspace_bytes": (_u64, [_u64]),
    "clv8_threshold": (C.c_int, [_vp, _vp, _u64, _u64, _u64, _vp, _vp]),
    "clv8_threshold_mode": (C.c_int, [_vp, _vp, _u64, _u64, _u64, C.c_int, _vp, _vp]),
    "clm4_mvm_v8": (C.c_int, [_vp, _vp, _u64, _u64, _vp, _vp, _vp, _vp, _vp, _vp]),
    "clm4_mvm_v8_scale_and_add": (C.c_int, [_vp, _vp, _u64, _u64, _vp, _vp, _vp, _vp, C.c_float, _vp, _vp, _vp, _vp, _vp, _vp]),
    "clv_iht_persistent_launches": (_u64, []),
    "clm4_iht_v8": (C.c_int, [_vp, _vp, _vp, _vp, _u64, _u64, _vp, _vp, _u64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _u64, _u64, C.c_float,
                              C.c_int, _vp, _vp]),
    "clm4_rowdots_v8": (C.c_int, [_vp, _vp, _u64, _u64, _vp, _vp, _vp, _vp]),
    "clv4_threshold_workspace_bytes": (_u64, [_u64]),
    "clv4_threshold": (C.c_int, [_vp, _vp, _u64, _u64, _u64, _vp, _vp]),
    "clv_threshold_reference_workspace_bytes": (_u64, [_u64]),
    "clv_threshold_reference_workspace_bytes_k": (_u64, [_u64, _u64]),
    "clv4_threshold_mode": (C.c_int, [_vp, _vp, _u64, _u64, _u64, C.c_int, _vp, _vp]),
    "clv4_threshold_heap": (C.c_int, [_vp, _vp, _u64, _u64, _u64, _vp, _vp, _vp]),
    "clv8_threshold_heap": (C.c_int, [_vp, _vp, _u64, _u64, _u64, _vp, _vp, _vp]),
    "clm4_transpose": (C.c_int, [_vp, _vp, _u64, _u64, _vp, _vp, _vp]),
    "clm4_mvm_f32": (C.c_int, [_vp, _vp, _u64, _u64, _vp, _vp, _vp]),
    "clm4_iht": (C.c_int, [_vp, _vp, _vp, _vp, _u64, _u64, _vp, _vp, _u64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _u64, _u64,
                          C.c_float, C.c_int, _vp, _vp]),
    "clm4_shard_partition": (C.c_int, [_u64, C.c_int, C.c_int, C.POINTER(_u64), C.POINTER(_u64)]),
    "clm4_sharded_create": (C.c_int, [C.POINTER(_vp), C.c_int, C.POINTER(C.c_int), _u64, _u64]),
    "clm4_sharded_destroy": (C.c_int, [_vp]),
    "clm4_sharded_info": (C.c_int, [_vp, C.c_int, C.POINTER(C.c_int), C.POINTER(_u64), C.POINTER(_u64), C.POINTER(_vp), C.POINTER(_vp)]),
    "clm4_sharded_timing": (C.c_int, [_vp, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "clm4_sharded_comm_info": (C.c_int, [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "clm4_sharded_upload": (C.c_int, [_vp, _vp, _vp]),
    "clm4_sharded_fill_random": (C.c_int, [_vp, _u64]),
    "clm4_sharded_mvm": (C.c_int, [_vp, _vp, _vp, C.c_int, _vp, _vp]),
    "clm4_sharded_result": (C.c_int, [_vp, C.c_int, C.POINTER(_vp), C.POINTER(_vp)]),
    "clm4_sharded_set_x": (C.c_int, [_vp, _vp, _vp, C.c_int]),
    "clm4_sharded_loop_begin": (C.c_int, [_vp, C.c_int]),
    "clm4_sharded_mvm_enqueue": (C.c_int, [_vp, C.c_int, C.c_int]),
    "clm4_sharded_sync": (C.c_int, [_vp]),
    "clm4_sharded_step_timing": (C.c_int, [_vp, C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "clm4_sharded_result_buf": (C.c_int, [_vp, C.c_int, C.c_int, C.POINTER(_vp), C.POINTER(_vp)]),
    "clm4_sharded_gemm": (C.c_int, [_vp, _vp, _vp, _u64, C.c_int, _vp]),
    "clm4_sharded_gemm_result": (C.c_int, [_vp, C.c_int, C.POINTER(_vp)]),
    "clm4_sharded_gemm_begin": (C.c_int, [_vp, _vp, _vp, _u64, C.c_int, C.c_int]),
    "clm4_sharded_gemm_begin_mode": (C.c_int, [_vp, _vp, _vp, _u64, C.c_int, C.c_int, C.c_int]),
    "clm4_sharded_gemm_enqueue": (C.c_int, [_vp, C.c_int, C.c_int]),
    "clm4_sharded_gemm_full": (C.c_int, [_vp, C.c_int, C.c_int, C.POINTER(_vp)]),
    "clv_fill_random_nibbles": (C.c_int, [_vp, _u64, _u64, _u64, _vp]),
    "clv_fill_random_scales": (C.c_int, [_vp, _u64, _u64, _u64, _vp]),
    "clv_fill_random_ints_f32": (C.c_int, [_vp, _u64, C.c_int, _u64, _u64, _vp]),
}


class CloverHipError(RuntimeError):
    pass


def load_library(path: str | Path | None = None, allow_probe: bool = False) -> C.CDLL:
    """dlopen the HIP library and attach prototypes; raises if it is not built.  The bench-only probe build (tools/_build/, GEMM loop variants
    that are wrong by construction) announces itself through clv_version() and is refused unless allow_probe is set."""
    p = Path(path) if path else hip_library_path()
    if not p.exists():
        raise CloverHipError(
            f"{p} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(the HIP extension is mandatory, there is no CPU fallback)")
    lib = C.CDLL(str(p))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError here = header/library mismatch
        fn.restype = res
        fn.argtypes = args
    if lib.clv_version().decode().startswith("clover_hip_probe") and not allow_probe:
        raise CloverHipError(f"{p} is the bench-only probe build ({lib.clv_version().decode()}): not a product library")
    return lib


class DevBuf:
    """A device allocation owned by Python (freed on GC)."""

    def __init__(self, hip: "CloverHip", nbytes: int):
        self.hip = hip
        self.nbytes = int(nbytes)
        p = _vp()
        hip.check(hip.lib.clv_malloc(C.byref(p), self.nbytes))
        self.ptr = p.value or 0

    def __del__(self):
        try:
            if self.ptr:
                self.hip.lib.clv_free(self.ptr)
                self.ptr = 0
        except Exception:
            pass

    def upload(self, arr: np.ndarray, stream=None) -> "DevBuf":
        a = np.ascontiguousarray(arr)
        assert a.nbytes <= self.nbytes
        self.hip.check(self.hip.lib.clv_memcpy_h2d(self.ptr, a.ctypes.data, a.nbytes, stream))
        self.hip.check(self.hip.lib.clv_stream_sync(stream))
        return self

    def download(self, dtype, count: int | None = None, stream=None) -> np.ndarray:
        dt = np.dtype(dtype)
        n = self.nbytes // dt.itemsize if count is None else int(count)
        out = np.empty(n, dtype=dt)
        self.hip.check(self.hip.lib.clv_memcpy_d2h(out.ctypes.data, self.ptr, out.nbytes, stream))
        return out

    def offset(self, nbytes: int) -> int:
        return self.ptr + int(nbytes)


class CloverHip:
    """Thin object wrapper: error checking + numpy convenience around the C ABI."""

    def __init__(self, path: str | Path | None = None, device: int | None = None, allow_probe: bool = False):
        self.lib = load_library(path, allow_probe=allow_probe)
        n = C.c_int(0)
        self.check(self.lib.clv_device_count(C.byref(n)))
        self.device_count = n.value
        if self.device_count < 1:
            raise CloverHipError("no HIP device visible (libclover_hip.so needs an MI355X / gfx950 GPU)")
        if device is not None:
            self.check(self.lib.clv_set_device(device))

    # -- plumbing ------------------------------------------------------------------------------
    def check(self, rc: int) -> None:
        if rc != 0:
            raise CloverHipError(f"clover_hip error {rc}: {self.lib.clv_last_error().decode()}")

    def alloc(self, nbytes: int) -> DevBuf:
        return DevBuf(self, nbytes)

    def to_device(self, arr: np.ndarray) -> DevBuf:
        a = np.ascontiguousarray(arr)
        return DevBuf(self, max(a.nbytes, 1)).upload(a)

    def sync(self) -> None:
        self.check(self.lib.clv_device_sync())

    def device_info(self) -> dict:
        name = C.create_string_buffer(256)
        cu = C.c_int(0)
        mem = _u64(0)
        self.check(self.lib.clv_device_info(name, 256, C.byref(cu), C.byref(mem)))
        return {"name": name.value.decode(), "compute_units": cu.value, "hbm_bytes": mem.value}

    def new_rng(self, key1: int, key2: int) -> DevBuf:
        st = self.alloc(256)      # CLV_RNG_STATE_BYTES
        self.check(self.lib.clv_rng_seed(st.ptr, key1, key2, None))
        return st

    def rng_get(self, st: DevBuf) -> tuple[np.ndarray, np.ndarray]:
        k1 = (_u64 * 4)()
        k2 = (_u64 * 4)()
        self.check(self.lib.clv_rng_get(st.ptr, k1, k2, None))
        return np.array(k1[:], dtype=np.uint64), np.array(k2[:], dtype=np.uint64)

    # -- CloverVector4 ---------------------------------------------------------------------------
    def v4_quantize(self, x: np.ndarray, rng: DevBuf | None = None):
        """numpy fp32 (padded) -> (bytes uint8[n/2], scales fp32[n/64]) through the device."""
        x = np.ascontiguousarray(x, dtype=np.float32)
        n = x.size
        dx = self.to_device(x)
        dq, ds = self.alloc(max(n // 2, 1)), self.alloc(max(n // 16, 4))
        self.check(self.lib.clv4_quantize(dx.ptr, n, dq.ptr, ds.ptr, rng.ptr if rng else None, None))
        return dq.download(np.uint8, n // 2), ds.download(np.float32, n // 64)

    def v4_restore(self, q: np.ndarray, s: np.ndarray) -> np.ndarray:
        n = q.size * 2
        dq, ds = self.to_device(q), self.to_device(s)
        dx = self.alloc(max(4 * n, 4))
        self.check(self.lib.clv4_restore(dq.ptr, ds.ptr, n, dx.ptr, None))
        return dx.download(np.float32, n)

    def v4_dot(self, qu, su, qv, sv, mode: int = DOT_EXACT) -> np.float32:
        n = qu.size * 2
        bufs = [self.to_device(a) for a in (qu, su, qv, sv)]
        out = self.alloc(4)
        self.check(self.lib.clv4_dot(bufs[0].ptr, bufs[1].ptr, bufs[2].ptr, bufs[3].ptr, n, mode, out.ptr, None, None))
        return out.download(np.float32, 1)[0]

    def v4_word_isums(self, qu, qv) -> np.ndarray:
        n = qu.size * 2
        du, dv = self.to_device(qu), self.to_device(qv)
        out = self.alloc(max(n // 2, 4))
        self.check(self.lib.clv4_word_isums(du.ptr, dv.ptr, n, out.ptr, None))
        return out.download(np.int32, n // 8)

    # -- CloverMatrix4 ---------------------------------------------------------------------------
    def m4_quantize(self, A: np.ndarray, rng: DevBuf | None = None):
        A = np.ascontiguousarray(A, dtype=np.float32)
        rows, cols = A.shape
        dA = self.to_device(A)
        dq, ds = self.alloc(max(rows * cols // 2, 1)), self.alloc(max((rows // 64) * (cols // 64) * 4, 4))
        self.check(self.lib.clm4_quantize(dA.ptr, rows, cols, dq.ptr, ds.ptr, rng.ptr if rng else None, None))
        return dq.download(np.uint8, rows * cols // 2), ds.download(np.float32, (rows // 64) * (cols // 64))

    def m4_restore(self, q, s, rows, cols) -> np.ndarray:
        dq, ds, dA = self.to_device(q), self.to_device(s), self.alloc(max(4 * rows * cols, 4))
        self.check(self.lib.clm4_restore(dq.ptr, ds.ptr, rows, cols, dA.ptr, None))
        return dA.download(np.float32, rows * cols).reshape(rows, cols)

    def m4_mvm(self, qA, sA, rows, cols, qx, sx, rng: DevBuf | None = None):
        b = [self.to_device(a) for a in (qA, sA, qx, sx)]
        dr, dsr = self.alloc(max(rows // 2, 1)), self.alloc(max(rows // 16, 4))
        self.check(self.lib.clm4_mvm(b[0].ptr, b[1].ptr, rows, cols, b[2].ptr, b[3].ptr, dr.ptr, dsr.ptr,
                                     rng.ptr if rng else None, None))
        return dr.download(np.uint8, rows // 2), dsr.download(np.float32, rows // 64)

    def m4_rowdots(self, qA, sA, rows, cols, qx, sx) -> np.ndarray:
        b = [self.to_device(a) for a in (qA, sA, qx, sx)]
        d = self.alloc(max(rows * 4, 4))
        self.check(self.lib.clm4_rowdots(b[0].ptr, b[1].ptr, rows, cols, b[2].ptr, b[3].ptr, d.ptr, None))
        return d.download(np.float32, rows)

    def v4_scale_and_add(self, qu, su, qv, sv, a: float, rng: DevBuf | None = None, in_place: bool = False):
        n = qu.size * 2
        b = [self.to_device(x) for x in (qu, su, qv, sv)]
        dr, dsr = (b[0], b[1]) if in_place else (self.alloc(max(n // 2, 1)), self.alloc(max(n // 16, 4)))
        self.check(self.lib.clv4_scale_and_add(b[0].ptr, b[1].ptr, b[2].ptr, b[3].ptr, a, n, dr.ptr, dsr.ptr,
                                               rng.ptr if rng else None, None))
        return dr.download(np.uint8, n // 2), dsr.download(np.float32, n // 64)

    def m4_mvm_scale_and_add(self, qA, sA, rows, cols, qx, sx, qu, su, a: float, rng: DevBuf | None = None, in_place: bool = False,
                             want_t: bool = True):
        """(t, st, r, sr) of clm4_mvm_scale_and_add; t/st are None when want_t is False"""
        b = [self.to_device(v) for v in (qA, sA, qx, sx, qu, su)]
        dt, dst = (self.alloc(rows // 2), self.alloc(rows // 16)) if want_t else (None, None)
        dr, dsr = (b[4], b[5]) if in_place else (self.alloc(rows // 2), self.alloc(rows // 16))
        self.check(self.lib.clm4_mvm_scale_and_add(b[0].ptr, b[1].ptr, rows, cols, b[2].ptr, b[3].ptr, b[4].ptr, b[5].ptr, a,
                                                   dt.ptr if dt else None, dst.ptr if dst else None, dr.ptr, dsr.ptr,
                                                   rng.ptr if rng else None, None))
        t = (dt.download(np.uint8, rows // 2), dst.download(np.float32, rows // 64)) if want_t else (None, None)
        return t[0], t[1], dr.download(np.uint8, rows // 2), dsr.download(np.float32, rows // 64)

    # -- mixed precision (CloverVector8) ----------------------------------------------------------
    def v8_quantize(self, x: np.ndarray, rng: DevBuf | None = None):
        x = np.ascontiguousarray(x, dtype=np.float32)
        n = x.size
        dx, dq, ds = self.to_device(x), self.alloc(max(n, 1)), self.alloc(max(n // 16, 4))
        self.check(self.lib.clv8_quantize(dx.ptr, n, dq.ptr, ds.ptr, rng.ptr if rng else None, None))
        return dq.download(np.int8, n), ds.download(np.float32, n // 64)

    def v8_restore(self, q, s) -> np.ndarray:
        n = q.size
        dq, ds, dx = self.to_device(q), self.to_device(s), self.alloc(max(4 * n, 4))
        self.check(self.lib.clv8_restore(dq.ptr, ds.ptr, n, dx.ptr, None))
        return dx.download(np.float32, n)

    def v8_scale_and_add(self, qu, su, qv, sv, a: float, rng: DevBuf | None = None, in_place: bool = False):
        n = qu.size
        b = [self.to_device(x) for x in (qu, su, qv, sv)]
        dr, dsr = (b[0], b[1]) if in_place else (self.alloc(max(n, 1)), self.alloc(max(n // 16, 4)))
        self.check(self.lib.clv8_scale_and_add(b[0].ptr, b[1].ptr, b[2].ptr, b[3].ptr, a, n, dr.ptr, dsr.ptr, rng.ptr if rng else None, None))
        return dr.download(np.int8, n), dsr.download(np.float32, n // 64)

    def v8_dot(self, qu, su, qv, sv, mode: int = DOT_EXACT) -> np.float32:
        n = qu.size
        bufs = [self.to_device(a) for a in (qu, su, qv, sv)]
        out = self.alloc(4)
        self.check(self.lib.clv8_dot(bufs[0].ptr, bufs[1].ptr, bufs[2].ptr, bufs[3].ptr, n, mode, out.ptr, None, None))
        return out.download(np.float32, 1)[0]

    def v8_threshold(self, q, s, n: int, k: int, mode: int = THRESHOLD_FAST) -> np.ndarray:
        dq, ds = self.to_device(q), self.to_device(s)
        self.check(self.lib.clv8_threshold_mode(dq.ptr, ds.ptr, n, q.size, k, mode, None, None))
        return dq.download(np.int8, q.size)

    def m4_mvm_v8(self, qA, sA, rows, cols, qx, sx, rng: DevBuf | None = None):
        b = [self.to_device(a) for a in (qA, sA, qx, sx)]
        dr, dsr = self.alloc(max(rows, 1)), self.alloc(max(rows // 16, 4))
        self.check(self.lib.clm4_mvm_v8(b[0].ptr, b[1].ptr, rows, cols, b[2].ptr, b[3].ptr, dr.ptr, dsr.ptr, rng.ptr if rng else None, None))
        return dr.download(np.int8, rows), dsr.download(np.float32, rows // 64)

    def m4_rowdots_v8(self, qA, sA, rows, cols, qx, sx) -> np.ndarray:
        b = [self.to_device(a) for a in (qA, sA, qx, sx)]
        d = self.alloc(max(rows * 4, 4))
        self.check(self.lib.clm4_rowdots_v8(b[0].ptr, b[1].ptr, rows, cols, b[2].ptr, b[3].ptr, d.ptr, None))
        return d.download(np.float32, rows)

    def v4_threshold(self, q, s, n: int, k: int, mode: int = THRESHOLD_FAST) -> np.ndarray:
        dq, ds = self.to_device(q), self.to_device(s)
        self.check(self.lib.clv4_threshold_mode(dq.ptr, ds.ptr, n, q.size * 2, k, mode, None, None))
        return dq.download(np.uint8, q.size)

    def m4_transpose(self, q, s, rows, cols):
        dq, ds = self.to_device(q), self.to_device(s)
        dt, dst = self.alloc(max(rows * cols // 2, 1)), self.alloc(max((rows // 64) * (cols // 64) * 4, 4))
        self.check(self.lib.clm4_transpose(dq.ptr, ds.ptr, rows, cols, dt.ptr, dst.ptr, None))
        return dt.download(np.uint8, rows * cols // 2), dst.download(np.float32, (rows // 64) * (cols // 64))

    def m4_mvm_f32(self, qA, sA, rows, cols, x) -> np.ndarray:
        b = [self.to_device(a) for a in (qA, sA, np.ascontiguousarray(x, dtype=np.float32))]
        d = self.alloc(max(rows * 4, 4))
        self.check(self.lib.clm4_mvm_f32(b[0].ptr, b[1].ptr, rows, cols, b[2].ptr, d.ptr, None))
        return d.download(np.float32, rows)

    def m4_gemm_i32(self, qA, M, K, qB, N, kb_begin=0, kb_count=None) -> np.ndarray:
        b = [self.to_device(a) for a in (qA, qB)]
        c = self.alloc(max(M * N * 4, 4))
        self.check(self.lib.clm4_gemm_i32(b[0].ptr, M, K, b[1].ptr, N, kb_begin, K // 64 - kb_begin if kb_count is None else kb_count, c.ptr, None))
        return c.download(np.int32, M * N).reshape(M, N)

    def m4_gemm_i32_prepared(self, qA, M, K, qB, N, kb_begin=0, kb_count=None, prepare=("A", "B")) -> np.ndarray:
        b = [self.to_device(a) for a in (qA, qB)]
        c = self.alloc(max(M * N * 4, 4))
        ops = {"A": C.c_void_p(), "B": C.c_void_p()}
        if "A" in prepare:
            self.check(self.lib.clm4_gemm_prepare(b[0].ptr, M, K, C.byref(ops["A"]), None))
        if "B" in prepare:
            self.check(self.lib.clm4_gemm_prepare(b[1].ptr, N, K, C.byref(ops["B"]), None))
        try:
            self.check(self.lib.clm4_gemm_i32_prepared(ops["A"], None if "A" in prepare else b[0].ptr, M, K, ops["B"], None if "B" in prepare else b[1].ptr, N,
                                                       kb_begin, K // 64 - kb_begin if kb_count is None else kb_count, c.ptr, None))
            return c.download(np.int32, M * N).reshape(M, N)
        finally:
            for o in ops.values():
                self.check(self.lib.clm4_gemm_release(o))

    def m4_gemm_prepared(self, qA, sA, M, K, qB, sB, N, prepare=("A", "B")) -> np.ndarray:
        b = [self.to_device(a) for a in (qA, sA, qB, sB)]
        c = self.alloc(max(M * N * 4, 4))
        ops = {"A": C.c_void_p(), "B": C.c_void_p()}
        if "A" in prepare:
            self.check(self.lib.clm4_gemm_prepare(b[0].ptr, M, K, C.byref(ops["A"]), None))
        if "B" in prepare:
            self.check(self.lib.clm4_gemm_prepare(b[2].ptr, N, K, C.byref(ops["B"]), None))
        try:
            for _ in range(2):          # the second call re-uses the images
                self.check(self.lib.clv_memset(c.ptr, 0xFF, c.nbytes, None))
                self.check(self.lib.clm4_gemm_prepared(ops["A"], None if "A" in prepare else b[0].ptr, b[1].ptr, M, K,
                                                       ops["B"], None if "B" in prepare else b[2].ptr, b[3].ptr, N, c.ptr, None))
            return c.download(np.float32, M * N).reshape(M, N)
        finally:
            for o in ops.values():
                self.check(self.lib.clm4_gemm_release(o))

    def m4_gemm(self, qA, sA, M, K, qB, sB, N) -> np.ndarray:
        b = [self.to_device(a) for a in (qA, sA, qB, sB)]
        c = self.alloc(max(M * N * 4, 4))
        self.check(self.lib.clm4_gemm(b[0].ptr, b[1].ptr, M, K, b[2].ptr, b[3].ptr, N, c.ptr, None))
        return c.download(np.float32, M * N).reshape(M, N)
