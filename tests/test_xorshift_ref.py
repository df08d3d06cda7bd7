"""Pins the generator on the REFERENCE itself.

tests/golden/xorshift_ref.json is produced by oracle/_ref/xorshift_gen = oracle/ref_xorshift.cpp compiled over
/root/reference/include/simdxorshift128plus.h (the one reference header that builds here without stand-ins;
recipe: `make -C oracle ref-fixtures`).  It holds, for the keys (12345, 67890) and the reference's own deterministic
seed pair (test/random/00_random.cpp:42): the four lanes after avx_xorshift128plus_init (:81-92), the first 256 draws of
avx_xorshift128plus (:97-109), the state after them, a digest of the next 2^20 draws, and the state after
avx_xorshift128plus_jump (:115-127).

CPU tests: the oracle's restatement == fixture (and == the live reference build when oracle/_ref is present).
GPU tests: clv_rng_seed == fixture lanes; the state a stochastic kernel leaves behind after consuming 256 (then 2^20 more)
draws == the fixture's states, i.e. the device jump-ahead walks the reference's stream.
"""
import json
from pathlib import Path

import numpy as np
import pytest

FIX = json.loads((Path(__file__).parent / "golden" / "xorshift_ref.json").read_text())
STREAMS = FIX["streams"]
IDS = [s["key1"] for s in STREAMS]


def _u64s(v):
    return np.array([int(x, 16) for x in v], dtype=np.uint64)


def _words(s):
    h = s["draw_words_hex"]
    return np.array([int(h[i:i + 8], 16) for i in range(0, len(h), 8)], dtype=np.uint32).reshape(-1, 8)


@pytest.mark.parametrize("s", STREAMS, ids=IDS)
def test_oracle_init_draws_digest_jump_equal_reference_fixture(oracle, s):
    r = oracle.rng(int(s["key1"]), int(s["key2"]))
    k0, k1 = oracle.rng_keys(r)
    assert np.array_equal(k0, _u64s(s["init_s0"])) and np.array_equal(k1, _u64s(s["init_s1"]))
    want = _words(s)
    assert want.shape == (FIX["draws"], 8)
    got = np.stack([oracle.rng_draw(r) for _ in range(FIX["draws"])])
    assert np.array_equal(got, want)
    k0, k1 = oracle.rng_keys(r)
    assert np.array_equal(k0, _u64s(s["after_draws_s0"])) and np.array_equal(k1, _u64s(s["after_draws_s1"]))
    x, a = oracle.rng_digest(r, FIX["long_draws"])
    assert (x, a) == (int(s["long_xor_fold"], 16), int(s["long_sum"], 16))
    k0, k1 = oracle.rng_keys(r)
    assert np.array_equal(k0, _u64s(s["after_long_s0"])) and np.array_equal(k1, _u64s(s["after_long_s1"]))
    oracle.rng_jump(r)
    k0, k1 = oracle.rng_keys(r)
    assert np.array_equal(k0, _u64s(s["after_jump_s0"])) and np.array_equal(k1, _u64s(s["after_jump_s1"]))


def test_fixture_matches_live_reference_build_when_present(oracle):
    """Where oracle/_ref/libxorshift_ref.so exists (builder container: compiled from /root/reference; GPU box: the prebuilt
    file), the committed fixture and the oracle are re-checked against the reference code itself, on other keys too."""
    from oracle.binding import RefXorshift
    if not RefXorshift.available():
        pytest.skip("no oracle/_ref build and no /root/reference here: the committed fixture is the pin")
    ref = RefXorshift()
    for s in STREAMS:
        s0, s1 = ref.init(int(s["key1"]), int(s["key2"]))
        assert np.array_equal(s0, _u64s(s["init_s0"])) and np.array_equal(s1, _u64s(s["init_s1"]))
        assert np.array_equal(ref.draw(s0, s1, FIX["draws"]), _words(s))
    rs = np.random.default_rng(7)
    for _ in range(16):
        k1, k2 = (int(v) for v in rs.integers(1, 2**63, size=2))
        s0, s1 = ref.init(k1, k2)
        r = oracle.rng(k1, k2)
        o0, o1 = oracle.rng_keys(r)
        assert np.array_equal(o0, s0) and np.array_equal(o1, s1)
        n = int(rs.integers(1, 2000))
        assert np.array_equal(ref.draw(s0, s1, n), np.stack([oracle.rng_draw(r) for _ in range(n)]))
        assert ref.digest(s0, s1, 50000) == oracle.rng_digest(r, 50000)
        ref.jump(s0, s1)
        oracle.rng_jump(r)
        o0, o1 = oracle.rng_keys(r)
        assert np.array_equal(o0, s0) and np.array_equal(o1, s1)


# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("s", STREAMS, ids=IDS)
def test_gpu_seed_and_consumed_stream_equal_reference_fixture(hip, s):
    st = hip.new_rng(int(s["key1"]), int(s["key2"]))
    k0, k1 = hip.rng_get(st)
    assert np.array_equal(k0, _u64s(s["init_s0"])) and np.array_equal(k1, _u64s(s["init_s1"]))
    # CloverVector4::quantize consumes two draws per 64-block (CloverVector4.h:690-734): 128 blocks = the fixture's 256 draws
    rs = np.random.default_rng(3)
    x = rs.integers(-10, 11, size=64 * FIX["draws"] // 2).astype(np.float32)
    hip.v4_quantize(x, rng=st)
    k0, k1 = hip.rng_get(st)
    assert np.array_equal(k0, _u64s(s["after_draws_s0"])) and np.array_equal(k1, _u64s(s["after_draws_s1"]))
    # ... and 2^19 more blocks = the fixture's 2^20 further draws (exercises the in-kernel GF(2) jump-ahead at scale)
    x = rs.integers(-10, 11, size=64 * FIX["long_draws"] // 2).astype(np.float32)
    hip.v4_quantize(x, rng=st)
    k0, k1 = hip.rng_get(st)
    assert np.array_equal(k0, _u64s(s["after_long_s0"])) and np.array_equal(k1, _u64s(s["after_long_s1"]))


@pytest.mark.gpu
def test_gpu_other_stochastic_kernels_walk_the_same_reference_stream(hip, oracle):
    """scaleAndAdd (2 draws per block) and mvm's re-quantise (2 per 64 output rows) land on the fixture's state after 256
    draws; matrix quantize (2 draws per tile row; 128x128 = 512 draws) lands where the fixture-pinned oracle lands."""
    s = STREAMS[1]
    key = (int(s["key1"]), int(s["key2"]))
    rs = np.random.default_rng(5)
    st = hip.new_rng(*key)
    x = rs.integers(-10, 11, size=64 * 128).astype(np.float32)
    q, sc = hip.v4_quantize(x)
    hip.v4_scale_and_add(q, sc, q, sc, 0.5, rng=st)
    k0, k1 = hip.rng_get(st)
    assert np.array_equal(k0, _u64s(s["after_draws_s0"])) and np.array_equal(k1, _u64s(s["after_draws_s1"]))
    st = hip.new_rng(*key)
    M, N = 64 * 128, 128
    qA, sA = hip.m4_quantize(rs.integers(-10, 11, size=(M, N)).astype(np.float32))
    qx, sx = hip.v4_quantize(rs.integers(-10, 11, size=N).astype(np.float32))
    hip.m4_mvm(qA, sA, M, N, qx, sx, rng=st)
    k0, k1 = hip.rng_get(st)
    assert np.array_equal(k0, _u64s(s["after_draws_s0"])) and np.array_equal(k1, _u64s(s["after_draws_s1"]))
    st = hip.new_rng(*key)
    r = oracle.rng(*key)
    A = rs.integers(-10, 11, size=(128, 128)).astype(np.float32)
    gq, gs = hip.m4_quantize(A, rng=st)
    oq, os_ = oracle.m4_quantize(A, rng=r)
    assert np.array_equal(gq, oq) and np.array_equal(gs.view(np.uint32), os_.view(np.uint32))
    k0, k1 = hip.rng_get(st)
    o0, o1 = oracle.rng_keys(r)
    assert np.array_equal(k0, o0) and np.array_equal(k1, o1)
