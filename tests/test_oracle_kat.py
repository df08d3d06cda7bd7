"""Pins the CPU oracle on the reference's known-answer vectors.

These are the only outputs of the reference's SIMD path available in this image (the reference needs
IPP/MKL headers to build), so they are the parity anchor: bytes, scale bits, dot bits (SIMD order AND
dot_scalar), restore bits, matrix quantize + mvm bytes/scales.
"""
import json
from pathlib import Path

import numpy as np

from conftest import bits, kat2_inputs, kat3_inputs

KAT = json.loads((Path(__file__).parent / "golden" / "kat_reference.json").read_text())


def test_kat1_readme_example(oracle):
    k = KAT["KAT1"]
    a = oracle.v4_quantize(np.full(k["n"], k["a_const"], np.float32))
    b = oracle.v4_quantize(np.full(k["n"], k["b_const"], np.float32))
    assert set(a[0].tolist()) == {int(k["bytes_all"], 16)} and set(b[0].tolist()) == {int(k["bytes_all"], 16)}
    assert [hex(v) for v in bits(a[1])] == [k["scale_a_bits"]] * 2
    assert [hex(v) for v in bits(b[1])] == [k["scale_b_bits"]] * 2
    assert hex(bits(oracle.v4_dot(*a, *b))) == k["dot_bits"]
    assert hex(bits(oracle.v4_dot_scalar(*a, *b))) == k["dot_scalar_bits"]


def test_kat2_quantize_dot_restore(oracle):
    k = KAT["KAT2"]
    x, y = kat2_inputs()
    qx, qy = oracle.v4_quantize(x), oracle.v4_quantize(y)
    assert [hex(v) for v in bits(qx[1])] == [k["qx_scale_bits"]] * 4
    assert [hex(v) for v in bits(qy[1])] == [k["qy_scale_bits"]] * 4
    assert qx[0][:32].tobytes().hex() == k["qx_bytes_0_31"]
    assert qy[0][:32].tobytes().hex() == k["qy_bytes_0_31"]
    assert hex(bits(oracle.v4_dot(*qx, *qy))) == k["dot_bits"]
    assert hex(bits(oracle.v4_dot_scalar(*qx, *qy))) == k["dot_scalar_bits"]
    r = oracle.v4_restore(*qx)
    assert [hex(v) for v in bits(r[:4])] == k["restore_qx_0_3_bits"]


def test_kat3_matrix_quantize_mvm(oracle):
    k = KAT["KAT3"]
    A, x = kat3_inputs()
    M, N = A.shape
    qA, sA = oracle.m4_quantize(A)
    qx = oracle.v4_quantize(x)
    r, sr = oracle.m4_mvm(qA, sA, M, N, *qx)
    assert r.tobytes().hex() == k["r_bytes_0_63"]
    assert [hex(v) for v in bits(sr)] == k["r_scale_bits"]
    got = [float(oracle.m4_get(qA, sA, M, N, 0, j)) for j in range(4)]
    np.testing.assert_allclose(got, k["qA_get_0_0_3"], rtol=0, atol=1e-5)
