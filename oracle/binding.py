"""ctypes wrapper of the CPU oracle (oracle/liboracle.so, oracle/liboracle_fast.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, from __graft_entry__.smoke() and from the cpu_baseline
leg of bench.py -- never from clover_amd/ (the product).  All arrays are numpy; sizes are padded sizes.
"""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_u8p = C.POINTER(C.c_uint8)
_fp = C.POINTER(C.c_float)
_i32p = C.POINTER(C.c_int32)
_u64 = C.c_uint64


class OrcRng(C.Structure):
    _fields_ = [("s0", _u64 * 4), ("s1", _u64 * 4)]


def _p(a: np.ndarray, t):
    return a.ctypes.data_as(t)


def _ensure_built() -> None:
    if not (_HERE / "liboracle.so").exists() or not (_HERE / "liboracle_fast.so").exists():
        subprocess.run(["make", "-C", str(_HERE), "all"], check=True, stdout=subprocess.DEVNULL)


class Oracle:
    """Scalar restatement (the checker).  See oracle/clover4_oracle.h for reference file:line citations."""

    def __init__(self):
        _ensure_built()
        L = C.CDLL(str(_HERE / "liboracle.so"))
        L.orc_v4_dot.restype = C.c_float
        L.orc_v4_dot_scalar.restype = C.c_float
        L.orc_v4_dot_f64.restype = C.c_double
        L.orc_v8_dot.restype = C.c_float
        L.orc_v8_dot_scalar.restype = C.c_float
        L.orc_v8_dot_f64.restype = C.c_double
        L.orc_v4_get.restype = C.c_float
        L.orc_m4_get.restype = C.c_float
        self.L = L

    # -- rng -----------------------------------------------------------------------------------
    def rng(self, key1: int, key2: int) -> OrcRng:
        r = OrcRng()
        self.L.orc_rng_init(C.byref(r), _u64(key1), _u64(key2))
        return r

    @staticmethod
    def rng_keys(r: OrcRng):
        return np.array(r.s0[:], dtype=np.uint64), np.array(r.s1[:], dtype=np.uint64)

    def rng_draw(self, r: OrcRng) -> np.ndarray:
        w = np.zeros(8, np.uint32)
        self.L.orc_rng_draw(C.byref(r), _p(w, C.POINTER(C.c_uint32)))
        return w

    def rng_jump(self, r: OrcRng) -> None:
        self.L.orc_rng_jump(C.byref(r))

    def rng_digest(self, r: OrcRng, count: int) -> tuple[int, int]:
        x, a = _u64(0), _u64(0)
        self.L.orc_rng_digest(C.byref(r), _u64(count), C.byref(x), C.byref(a))
        return x.value, a.value

    # -- vector --------------------------------------------------------------------------------
    def v4_quantize(self, x: np.ndarray, rng: OrcRng | None = None):
        x = np.ascontiguousarray(x, dtype=np.float32)
        n = x.size
        q = np.zeros(n // 2, np.uint8)
        s = np.zeros(n // 64, np.float32)
        self.L.orc_v4_quantize(_p(x, _fp), _u64(n), _p(q, _u8p), _p(s, _fp), C.byref(rng) if rng is not None else None)
        return q, s

    def v4_restore(self, q, s) -> np.ndarray:
        n = q.size * 2
        x = np.zeros(n, np.float32)
        self.L.orc_v4_restore(_p(q, _u8p), _p(s, _fp), _u64(n), _p(x, _fp))
        return x

    def v4_get(self, q, s, pos: int) -> np.float32:
        return np.float32(self.L.orc_v4_get(_p(q, _u8p), _p(s, _fp), _u64(pos)))

    def v4_dot(self, qu, su, qv, sv) -> np.float32:
        return np.float32(self.L.orc_v4_dot(_p(qu, _u8p), _p(su, _fp), _p(qv, _u8p), _p(sv, _fp), _u64(qu.size * 2)))

    def v4_dot_scalar(self, qu, su, qv, sv) -> np.float32:
        return np.float32(self.L.orc_v4_dot_scalar(_p(qu, _u8p), _p(su, _fp), _p(qv, _u8p), _p(sv, _fp), _u64(qu.size * 2)))

    def v4_dot_f64(self, qu, su, qv, sv) -> float:
        return float(self.L.orc_v4_dot_f64(_p(qu, _u8p), _p(su, _fp), _p(qv, _u8p), _p(sv, _fp), _u64(qu.size * 2)))

    def v4_word_isums(self, qu, qv) -> np.ndarray:
        n = qu.size * 2
        out = np.zeros(n // 8, np.int32)
        self.L.orc_v4_word_isums(_p(qu, _u8p), _p(qv, _u8p), _u64(n), _p(out, _i32p))
        return out

    def v4_scale_and_add(self, qu, su, qv, sv, a: float, rng: OrcRng | None = None):
        n = qu.size * 2
        r = np.zeros(n // 2, np.uint8)
        sr = np.zeros(n // 64, np.float32)
        self.L.orc_v4_scale_and_add(_p(qu, _u8p), _p(su, _fp), _p(qv, _u8p), _p(sv, _fp), C.c_float(a), _u64(n),
                                    _p(r, _u8p), _p(sr, _fp), C.byref(rng) if rng is not None else None)
        return r, sr

    def v4_threshold(self, q, s, n: int, k: int) -> np.ndarray:
        out = np.array(q, dtype=np.uint8, copy=True)
        self.L.orc_v4_threshold(_p(out, _u8p), _p(s, _fp), _u64(n), _u64(k))
        return out

    def m4_transpose(self, q, s, rows, cols):
        qt = np.zeros(rows * cols // 2, np.uint8)
        st = np.zeros((rows // 64) * (cols // 64), np.float32)
        self.L.orc_m4_transpose(_p(q, _u8p), _p(s, _fp), _u64(rows), _u64(cols), _p(qt, _u8p), _p(st, _fp))
        return qt, st

    # -- matrix --------------------------------------------------------------------------------
    def m4_quantize(self, A: np.ndarray, rng: OrcRng | None = None):
        A = np.ascontiguousarray(A, dtype=np.float32)
        rows, cols = A.shape
        q = np.zeros(rows * cols // 2, np.uint8)
        s = np.zeros((rows // 64) * (cols // 64), np.float32)
        self.L.orc_m4_quantize(_p(A, _fp), _u64(rows), _u64(cols), _p(q, _u8p), _p(s, _fp),
                               C.byref(rng) if rng is not None else None)
        return q, s

    def m4_restore(self, q, s, rows, cols) -> np.ndarray:
        A = np.zeros(rows * cols, np.float32)
        self.L.orc_m4_restore(_p(q, _u8p), _p(s, _fp), _u64(rows), _u64(cols), _p(A, _fp))
        return A.reshape(rows, cols)

    def m4_get(self, q, s, rows, cols, i, j) -> np.float32:
        return np.float32(self.L.orc_m4_get(_p(q, _u8p), _p(s, _fp), _u64(rows), _u64(cols), _u64(i), _u64(j)))

    def m4_rowdots(self, qA, sA, rows, cols, qx, sx) -> np.ndarray:
        d = np.zeros(rows, np.float32)
        self.L.orc_m4_rowdots(_p(qA, _u8p), _p(sA, _fp), _u64(rows), _u64(cols), _p(qx, _u8p), _p(sx, _fp), _p(d, _fp))
        return d

    def m4_mvm(self, qA, sA, rows, cols, qx, sx, rng: OrcRng | None = None):
        r = np.zeros(rows // 2, np.uint8)
        sr = np.zeros(rows // 64, np.float32)
        self.L.orc_m4_mvm(_p(qA, _u8p), _p(sA, _fp), _u64(rows), _u64(cols), _p(qx, _u8p), _p(sx, _fp),
                          _p(r, _u8p), _p(sr, _fp), C.byref(rng) if rng is not None else None)
        return r, sr

    def m4_mvm_f32(self, qA, sA, rows, cols, x) -> np.ndarray:
        x = np.ascontiguousarray(x, dtype=np.float32)
        r = np.zeros(rows, np.float32)
        self.L.orc_m4_mvm_f32(_p(qA, _u8p), _p(sA, _fp), _u64(rows), _u64(cols), _p(x, _fp), _p(r, _fp))
        return r

    # -- mixed precision: CloverVector8 -------------------------------------------------------------
    def v8_quantize(self, x: np.ndarray, rng: OrcRng | None = None):
        x = np.ascontiguousarray(x, dtype=np.float32)
        n = x.size
        q = np.zeros(n, np.int8)
        s = np.zeros(n // 64, np.float32)
        self.L.orc_v8_quantize(_p(x, _fp), _u64(n), _p(q, C.POINTER(C.c_int8)), _p(s, _fp), C.byref(rng) if rng is not None else None)
        return q, s

    def v8_restore(self, q, s) -> np.ndarray:
        x = np.zeros(q.size, np.float32)
        self.L.orc_v8_restore(_p(q, C.POINTER(C.c_int8)), _p(s, _fp), _u64(q.size), _p(x, _fp))
        return x

    def v8_scale_and_add(self, qu, su, qv, sv, a: float, rng: OrcRng | None = None):
        n = qu.size
        i8 = C.POINTER(C.c_int8)
        r = np.zeros(n, np.int8)
        sr = np.zeros(n // 64, np.float32)
        self.L.orc_v8_scale_and_add(_p(qu, i8), _p(su, _fp), _p(qv, i8), _p(sv, _fp), C.c_float(a), _u64(n), _p(r, i8), _p(sr, _fp),
                                    C.byref(rng) if rng is not None else None)
        return r, sr

    def v8_dot(self, qu, su, qv, sv) -> np.float32:
        i8 = C.POINTER(C.c_int8)
        return np.float32(self.L.orc_v8_dot(_p(qu, i8), _p(su, _fp), _p(qv, i8), _p(sv, _fp), _u64(qu.size)))

    def v8_dot_scalar(self, qu, su, qv, sv) -> np.float32:
        i8 = C.POINTER(C.c_int8)
        return np.float32(self.L.orc_v8_dot_scalar(_p(qu, i8), _p(su, _fp), _p(qv, i8), _p(sv, _fp), _u64(qu.size)))

    def v8_dot_f64(self, qu, su, qv, sv) -> float:
        i8 = C.POINTER(C.c_int8)
        return float(self.L.orc_v8_dot_f64(_p(qu, i8), _p(su, _fp), _p(qv, i8), _p(sv, _fp), _u64(qu.size)))

    def v8_threshold(self, q, s, n: int, k: int) -> np.ndarray:
        out = np.array(q, dtype=np.int8, copy=True)
        self.L.orc_v8_threshold(_p(out, C.POINTER(C.c_int8)), _p(s, _fp), _u64(n), _u64(k))
        return out

    def m4_rowdots_v8(self, qA, sA, rows, cols, qx, sx, f64: bool = False) -> np.ndarray:
        d = np.zeros(rows, np.float32)
        fn = self.L.orc_m4_rowdots_v8_f64 if f64 else self.L.orc_m4_rowdots_v8
        fn(_p(qA, _u8p), _p(sA, _fp), _u64(rows), _u64(cols), _p(qx, C.POINTER(C.c_int8)), _p(sx, _fp), _p(d, _fp))
        return d

    def m4_mvm_v8(self, qA, sA, rows, cols, qx, sx, rng: OrcRng | None = None):
        r = np.zeros(rows, np.int8)
        sr = np.zeros(rows // 64, np.float32)
        self.L.orc_m4_mvm_v8(_p(qA, _u8p), _p(sA, _fp), _u64(rows), _u64(cols), _p(qx, C.POINTER(C.c_int8)), _p(sx, _fp),
                             _p(r, C.POINTER(C.c_int8)), _p(sr, _fp), C.byref(rng) if rng is not None else None)
        return r, sr

    def m4_gemm(self, qA, sA, M, K, qB, sB, N) -> np.ndarray:
        c = np.zeros(M * N, np.float32)
        self.L.orc_m4_gemm(_p(qA, _u8p), _p(sA, _fp), _u64(M), _u64(K), _p(qB, _u8p), _p(sB, _fp), _u64(N), _p(c, _fp))
        return c.reshape(M, N)

    def m4_gemm_isums(self, qA, M, K, qB, N) -> np.ndarray:
        s = np.zeros(M * N * (K // 64), np.int32)
        self.L.orc_m4_gemm_isums(_p(qA, _u8p), _u64(M), _u64(K), _p(qB, _u8p), _u64(N), _p(s, _i32p))
        return s.reshape(M, N, K // 64)


class FastOracle:
    """AVX2 + OpenMP restatement (the timed CPU baseline); checked against Oracle in tests/."""

    def __init__(self, threads: int | None = None):
        _ensure_built()
        L = C.CDLL(str(_HERE / "liboracle_fast.so"))
        L.orcf_v4_dot.restype = C.c_float
        L.orcf_v4_dot_parallel.restype = C.c_float
        L.orcf_max_threads.restype = C.c_int
        self.L = L
        if threads:
            L.orcf_set_threads(C.c_int(threads))

    def max_threads(self) -> int:
        return int(self.L.orcf_max_threads())

    def set_threads(self, n: int) -> None:
        self.L.orcf_set_threads(C.c_int(n))

    def set_kernel(self, name: str) -> None:
        """"maddubs" (default): the reference's instruction mix (CloverVector4.h:1136-1180); "planes": int16 planes + vpmaddwd"""
        self.L.orcf_set_kernel(C.c_int({"planes": 0, "maddubs": 1}[name]))

    def first_touch_copy(self, A: np.ndarray, rows: int) -> np.ndarray:
        """a copy of the packed matrix whose pages were first touched by the threads that will read them in m4_mvm"""
        out = np.empty(A.size, np.uint8)                 # np.empty does not touch the pages
        self.L.orcf_first_touch_copy(_p(out, _u8p), _p(A, _u8p), _u64(rows), _u64(A.size // rows))
        return out

    def v4_quantize(self, x):
        x = np.ascontiguousarray(x, dtype=np.float32)
        n = x.size
        q = np.zeros(n // 2, np.uint8)
        s = np.zeros(n // 64, np.float32)
        self.L.orcf_v4_quantize(_p(x, _fp), _u64(n), _p(q, _u8p), _p(s, _fp))
        return q, s

    def v4_dot(self, qu, su, qv, sv) -> np.float32:
        return np.float32(self.L.orcf_v4_dot(_p(qu, _u8p), _p(su, _fp), _p(qv, _u8p), _p(sv, _fp), _u64(qu.size * 2)))

    def v4_dot_parallel(self, qu, su, qv, sv) -> np.float32:
        """CloverVector4::dot_parallel's decomposition (block pairs over threads + reduction): tolerance-only, as in the reference"""
        return np.float32(self.L.orcf_v4_dot_parallel(_p(qu, _u8p), _p(su, _fp), _p(qv, _u8p), _p(sv, _fp), _u64(qu.size * 2)))

    def v4_quantize_into(self, x, q, s) -> None:
        """quantize without allocating (timed loops)"""
        self.L.orcf_v4_quantize(_p(x, _fp), _u64(x.size), _p(q, _u8p), _p(s, _fp))

    def m4_gemm(self, qA, sA, M, K, qB, sB, N, out=None) -> np.ndarray:
        """the build-defined GEMM (one fma chain over the K-blocks per element), AVX2 + OpenMP: whole-result checks at the timed sizes"""
        c = np.empty(M * N, np.float32) if out is None else out
        self.L.orcf_m4_gemm(_p(qA, _u8p), _p(sA, _fp), _u64(M), _u64(K), _p(qB, _u8p), _p(sB, _fp), _u64(N), _p(c, _fp))
        return c.reshape(M, N)

    def m4_mvm(self, qA, sA, rows, cols, qx, sx, out=None):
        r = np.zeros(rows // 2, np.uint8) if out is None else out[0]
        sr = np.zeros(rows // 64, np.float32) if out is None else out[1]
        self.L.orcf_m4_mvm(_p(qA, _u8p), _p(sA, _fp), _u64(rows), _u64(cols), _p(qx, _u8p), _p(sx, _fp), _p(r, _u8p), _p(sr, _fp))
        return r, sr


class RefXorshift:
    """The REFERENCE's generator itself (oracle/_ref/libxorshift_ref.so = oracle/ref_xorshift.cpp compiled over
    /root/reference/include/simdxorshift128plus.h).  Built by `make -C oracle ref` where the reference tree exists; the
    prebuilt .so travels to the GPU box.  available() is False where neither holds (then only the committed fixture pins)."""

    PATH = _HERE / "_ref" / "libxorshift_ref.so"

    @classmethod
    def available(cls) -> bool:
        if not cls.PATH.exists():
            subprocess.run(["make", "-C", str(_HERE), "ref"], check=False, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        return cls.PATH.exists()

    def __init__(self):
        if not self.available():
            raise FileNotFoundError(str(self.PATH))
        self.L = C.CDLL(str(self.PATH))

    def init(self, key1: int, key2: int):
        s0, s1 = np.zeros(4, np.uint64), np.zeros(4, np.uint64)
        self.L.ref_xs_init(_u64(key1), _u64(key2), _p(s0, C.POINTER(_u64)), _p(s1, C.POINTER(_u64)))
        return s0, s1

    def draw(self, s0, s1, count: int) -> np.ndarray:
        W = np.zeros(8 * count, np.uint32)
        self.L.ref_xs_draw(_p(s0, C.POINTER(_u64)), _p(s1, C.POINTER(_u64)), _u64(count), _p(W, C.POINTER(C.c_uint32)))
        return W.reshape(count, 8)

    def digest(self, s0, s1, count: int) -> tuple[int, int]:
        x, a = _u64(0), _u64(0)
        self.L.ref_xs_digest(_p(s0, C.POINTER(_u64)), _p(s1, C.POINTER(_u64)), _u64(count), C.byref(x), C.byref(a))
        return x.value, a.value

    def jump(self, s0, s1) -> None:
        self.L.ref_xs_jump(_p(s0, C.POINTER(_u64)), _p(s1, C.POINTER(_u64)))
