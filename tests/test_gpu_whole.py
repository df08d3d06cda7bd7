"""WHOLE-result parity at the sizes bench.py times: every output byte / float of the timed call is compared with the
CPU restatement, not a sample of it.

  C3  mvm 65536 x 65536        : the 2 GiB matrix is downloaded once and the entire 65536-row result (32 KiB + 4 KiB) compared with the
                                 AVX2 + OpenMP port (bit-identical to the scalar oracle: tests/test_oracle_properties.py);
  C4  gemm 4096^3 and 8192^3   : all of C (64 MiB / 256 MiB of fp32) against orcf_m4_gemm -- the definition's one fma chain per element,
                                 vectorised on the host, itself == the scalar orc_m4_gemm on small shapes; FP6 kernel, both tiles' default;
      + the int32 K-block sums of the whole product against integer numpy on a 2048^3 product;
  stochastic                   : one 2^24-element vector quantize, and an 8192 x 8192 matrix quantize followed by mvm, against the scalar
                                 oracle consuming the same XORShift stream (reference: CloverVector4.h:690-734, CloverMatrix4.h:925-932).
Reference relations these mirror: test/validate/03_matrix.cpp:248-326 (mvm SIMD == scalar == parallel, exact)."""
import numpy as np
import pytest

from conftest import bits

pytestmark = pytest.mark.gpu


def same(a, b):
    return np.array_equal(np.asarray(a).view(np.uint8), np.asarray(b).view(np.uint8))


def _fill(hip, rows, cols, seed):
    A = hip.alloc(rows * cols // 2)
    sA = hip.alloc((rows // 64) * (cols // 64) * 4)
    hip.check(hip.lib.clv_fill_random_nibbles(A.ptr, A.nbytes, seed, 0, None))
    hip.check(hip.lib.clv_fill_random_scales(sA.ptr, sA.nbytes // 4, seed + 1, 0, None))
    return A, sA


def test_c3_mvm_65536_whole_result(hip, fast_oracle):
    """the matrix and x of bench.py's default step (same seeds), every one of the 1024 output blocks"""
    rows = cols = 65536
    seed = 0xC10FE4
    A, sA = _fill(hip, rows, cols, seed)
    x, sx = hip.alloc(cols // 2), hip.alloc(cols // 64 * 4)
    hip.check(hip.lib.clv_fill_random_nibbles(x.ptr, x.nbytes, seed + 2, 0, None))
    hip.check(hip.lib.clv_fill_random_scales(sx.ptr, cols // 64, seed + 3, 0, None))
    r, sr = hip.alloc(rows // 2), hip.alloc(rows // 64 * 4)
    hip.check(hip.lib.clm4_mvm(A.ptr, sA.ptr, rows, cols, x.ptr, sx.ptr, r.ptr, sr.ptr, None, None))
    r_h, sr_h = r.download(np.uint8), sr.download(np.float32)
    qA, sAh, qx, sxh = A.download(np.uint8), sA.download(np.float32), x.download(np.uint8), sx.download(np.float32)
    ro, sro = fast_oracle.m4_mvm(qA, sAh, rows, cols, qx, sxh)
    assert same(r_h, ro) and same(sr_h, sro)
    assert len(set(r_h.tolist())) > 100 and float(sr_h.min()) > 0


@pytest.mark.parametrize("G", [4096, 8192])
def test_gemm_whole_result_against_the_definition(hip, fast_oracle, G):
    """8192^3 = BASELINE configs[3], the call bench.py's `gemm` object times (same seeds): all 6.7e7 elements"""
    M = N = K = G
    A, sA = _fill(hip, M, K, 21)
    B, sB = _fill(hip, N, K, 22)
    hip.check(hip.lib.clv_fill_random_scales(sA.ptr, sA.nbytes // 4, 23, 0, None))
    hip.check(hip.lib.clv_fill_random_scales(sB.ptr, sB.nbytes // 4, 24, 0, None))
    Cd = hip.alloc(M * N * 4)
    hip.check(hip.lib.clv_memset(Cd.ptr, 0xFF, Cd.nbytes, None))
    hip.check(hip.lib.clm4_gemm(A.ptr, sA.ptr, M, K, B.ptr, sB.ptr, N, Cd.ptr, None))
    Ch = Cd.download(np.float32)
    Co = fast_oracle.m4_gemm(A.download(np.uint8), sA.download(np.float32), M, K, B.download(np.uint8), sB.download(np.float32), N)
    diff = np.flatnonzero(Ch.view(np.uint32) != Co.reshape(-1).view(np.uint32))
    assert diff.size == 0, (diff.size, diff[:8], Ch[diff[:8]], Co.reshape(-1)[diff[:8]])
    assert np.isfinite(Ch).all() and np.unique(Ch[:4096]).size > 1000


def test_gemm_i32_whole_result_2048(hip):
    """the exact integer GEMM (SURVEY a8 output (1)) over all K-blocks, every element, against integer numpy"""
    G = 2048
    A, _ = _fill(hip, G, G, 31)
    B, _ = _fill(hip, G, G, 33)
    qA, qB = A.download(np.uint8), B.download(np.uint8)

    def unpack(q):
        hi = (q.view(np.int8) >> 4).astype(np.float32)
        lo = ((q << 4).view(np.int8) >> 4).astype(np.float32)
        return np.stack([hi, lo], axis=1).reshape(G, G)
    S_ref = (unpack(qA).astype(np.float64) @ unpack(qB).astype(np.float64).T)          # |sum| <= 49 * 2048: exact in fp64
    S = hip.m4_gemm_i32(qA, G, G, qB, G)
    assert np.array_equal(S.reshape(G, G).astype(np.float64), S_ref)


def test_stochastic_vector_quantize_2p24_whole(hip, oracle):
    n = 1 << 24
    x = (np.random.default_rng(7).standard_normal(n) * 3).astype(np.float32)
    st = hip.new_rng(12345, 67890)
    q, s = hip.v4_quantize(x, st)
    orng = oracle.rng(12345, 67890)
    qo, so = oracle.v4_quantize(x, orng)
    assert same(q, qo) and same(s, so)
    k1, k2 = hip.rng_get(st)
    ok1, ok2 = oracle.rng_keys(orng)
    assert np.array_equal(k1, ok1) and np.array_equal(k2, ok2)          # 2^19 draws later: the same place in the stream


@pytest.mark.parametrize("width", [4, 8])
def test_stochastic_vector_quantize_large_form_whole(hip, oracle, width):
    """beyond 2^18 blocks the stochastic vector quantizers take their large-vector form on their own (lane = one float4, 16 segments per
    wave: k_v4_quantize_st / k_v8_quantize_st, NSEG == 16): the whole result of a ragged n just above 2^25, both element widths, and
    the place the stream is left at"""
    n = (1 << 25) + 128 * 3
    x = (np.random.default_rng(70 + width).standard_normal(n) * 3).astype(np.float32)
    st, orng = hip.new_rng(424242, 171717), oracle.rng(424242, 171717)
    q, s = (hip.v4_quantize if width == 4 else hip.v8_quantize)(x, st)
    qo, so = (oracle.v4_quantize if width == 4 else oracle.v8_quantize)(x, orng)
    assert same(q, qo) and same(s, so)
    k1, k2 = hip.rng_get(st)
    ok1, ok2 = oracle.rng_keys(orng)
    assert np.array_equal(k1, ok1) and np.array_equal(k2, ok2)


def test_stochastic_matrix_quantize_and_mvm_8192_whole(hip, oracle):
    """8192 x 8192: the quantize consumes 2 draws per tile row in column-block-outer order (CloverMatrix4.h:524-525), the mvm
    2 draws per output block with the 8j+g lane map (:925-932) -- one stream through both calls"""
    rows = cols = 8192
    rng = np.random.default_rng(11)
    A = (rng.standard_normal((rows, cols)) * 2).astype(np.float32)
    xv = rng.integers(-10, 11, cols).astype(np.float32)
    st = hip.new_rng(445560390295639063, 2935984234003016713)               # the reference's own seed pair (test/random/00_random.cpp:42)
    orng = oracle.rng(445560390295639063, 2935984234003016713)
    qA, sA = hip.m4_quantize(A, st)
    oA, osA = oracle.m4_quantize(A, orng)
    assert same(qA, oA) and same(sA, osA)
    qx, sx = hip.v4_quantize(xv, st)
    ox, osx = oracle.v4_quantize(xv, orng)
    assert same(qx, ox) and same(sx, osx)
    r, sr = hip.m4_mvm(qA, sA, rows, cols, qx, sx, st)
    ro, sro = oracle.m4_mvm(oA, osA, rows, cols, ox, osx, orng)
    assert same(r, ro) and same(sr, sro)
    k1, k2 = hip.rng_get(st)
    ok1, ok2 = oracle.rng_keys(orng)
    assert np.array_equal(k1, ok1) and np.array_equal(k2, ok2)
