"""Stochastic calls inside a hipGraph (clv_rng_graph_mode).  A state in graph mode keeps its launch stamps on the device (a one-thread
tick kernel in front of every stochastic kernel), so the calls can be captured once and replayed: every replay consumes the NEXT part
of the XORShift stream -- the same nibbles as the same calls issued one after another, i.e. as the reference's sequential methods on one
object (CloverVector4.h:690-734, 1196-1478) -- checked against the oracle walking the stream."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def same(a, b):
    return np.array_equal(np.asarray(a).view(np.uint8), np.asarray(b).view(np.uint8))


@pytest.mark.parametrize("n", [4096, 1 << 18])
def test_captured_stochastic_calls_walk_the_stream_on_every_replay(hip, oracle, n):
    lib = hip.lib
    rng = np.random.default_rng(n)
    x = (rng.standard_normal(n) * 3).astype(np.float32)
    xd = hip.to_device(x)
    q, s = hip.alloc(n // 2), hip.alloc(n // 16)
    q2, s2 = hip.alloc(n // 2), hip.alloc(n // 16)
    st = hip.new_rng(12345, 67890)
    orng = oracle.rng(12345, 67890)
    # an ordinary (host-stamped) call first: builds the generator tables outside the capture and moves the stream on
    hip.check(lib.clv4_quantize(xd.ptr, n, q.ptr, s.ptr, st.ptr, None))
    hip.sync()
    oq, osc = oracle.v4_quantize(x, orng)
    assert same(q.download(np.uint8), oq)
    hip.check(lib.clv_rng_graph_mode(st.ptr, 1, None))
    hip.sync()
    # the HIP graph API straight from the runtime the library is linked against (no torch: a second HIP runtime in the process is not needed)
    rt = C.CDLL("libamdhip64.so")
    stream, graph, gexec = C.c_void_p(), C.c_void_p(), C.c_void_p()

    def ok(rc):
        assert rc == 0, f"HIP runtime call failed: {rc}"
    ok(rt.hipStreamCreate(C.byref(stream)))
    ok(rt.hipStreamBeginCapture(stream, 0))                                       # hipStreamCaptureModeGlobal
    hip.check(lib.clv4_quantize(xd.ptr, n, q.ptr, s.ptr, st.ptr, stream))
    hip.check(lib.clv4_scale_and_add(q.ptr, s.ptr, q.ptr, s.ptr, 0.5, n, q2.ptr, s2.ptr, st.ptr, stream))
    ok(rt.hipStreamEndCapture(stream, C.byref(graph)))
    ok(rt.hipGraphInstantiate(C.byref(gexec), graph, None, None, 0))
    for rep in range(3):
        ok(rt.hipGraphLaunch(gexec, stream))
        ok(rt.hipStreamSynchronize(stream))
        oq, osc = oracle.v4_quantize(x, orng)
        o2, os2 = oracle.v4_scale_and_add(oq, osc, oq, osc, 0.5, orng)
        assert same(q.download(np.uint8), oq) and same(s.download(np.float32), osc), rep
        assert same(q2.download(np.uint8), o2) and same(s2.download(np.float32), os2), rep
    ok(rt.hipGraphExecDestroy(gexec))
    ok(rt.hipGraphDestroy(graph))
    ok(rt.hipStreamDestroy(stream))
    # the state is where 1 + 3 x 2 sequential calls leave it, and ordinary calls continue from there after leaving graph mode
    k1, k2 = hip.rng_get(st)
    ok1, ok2 = oracle.rng_keys(orng)
    assert np.array_equal(k1, ok1) and np.array_equal(k2, ok2)
    hip.check(lib.clv_rng_graph_mode(st.ptr, 0, None))
    hip.check(lib.clv4_quantize(xd.ptr, n, q.ptr, s.ptr, st.ptr, None))
    hip.sync()
    oq, _ = oracle.v4_quantize(x, orng)
    assert same(q.download(np.uint8), oq)


@pytest.mark.parametrize("n", [128, 8192, 1 << 20, (1 << 24) + 256])
def test_captured_dot_fast_replays(hip, oracle, n):
    """round 5: clv4_dot FAST is ONE launch whose workgroups hand their partials to a collector workgroup through slots that must be zero
    when the kernel starts (clv_internal_sync_slots) -- the collector clears what it read.  Captured once, replayed on changing operands:
    every replay must find clean slots and give the bits of an ordinary call (and of the two-launch form: same fixed tree)."""
    from clover_amd.lib_binding import DOT_FAST
    lib = hip.lib
    rng = np.random.default_rng(n)
    rt = C.CDLL("libamdhip64.so")
    stream, graph, gexec = C.c_void_p(), C.c_void_p(), C.c_void_p()

    def ok(rc):
        assert rc == 0, f"HIP runtime call failed: {rc}"
    ok(rt.hipStreamCreate(C.byref(stream)))
    qa, sa, qb, sb = hip.alloc(n // 2), hip.alloc(n // 16), hip.alloc(n // 2), hip.alloc(n // 16)
    out = hip.alloc(8)

    def fill(seed):
        hip.check(lib.clv_fill_random_nibbles(qa.ptr, qa.nbytes, seed, 0, stream))
        hip.check(lib.clv_fill_random_nibbles(qb.ptr, qb.nbytes, seed + 1, 0, stream))
        hip.check(lib.clv_fill_random_scales(sa.ptr, sa.nbytes // 4, seed + 2, 0, stream))
        hip.check(lib.clv_fill_random_scales(sb.ptr, sb.nbytes // 4, seed + 3, 0, stream))
    fill(1)
    # one ordinary call on this stream first: the slots and the scratch of (device, stream) are allocated outside the capture
    hip.check(lib.clv4_dot(qa.ptr, sa.ptr, qb.ptr, sb.ptr, n, DOT_FAST, out.ptr, None, stream))
    ok(rt.hipStreamSynchronize(stream))
    ok(rt.hipStreamBeginCapture(stream, 0))
    hip.check(lib.clv4_dot(qa.ptr, sa.ptr, qb.ptr, sb.ptr, n, DOT_FAST, out.ptr, None, stream))
    ok(rt.hipStreamEndCapture(stream, C.byref(graph)))
    ok(rt.hipGraphInstantiate(C.byref(gexec), graph, None, None, 0))
    for rep in range(4):
        fill(10 * rep + int(rng.integers(1, 1000)))
        ok(rt.hipGraphLaunch(gexec, stream))
        ok(rt.hipStreamSynchronize(stream))
        got = out.download(np.uint32)[0]
        hip.check(lib.clv4_dot(qa.ptr, sa.ptr, qb.ptr, sb.ptr, n, DOT_FAST, out.ptr + 4, None, stream))
        ok(rt.hipStreamSynchronize(stream))
        plain = out.download(np.uint32)[1]
        assert got == plain, (rep, hex(got), hex(plain))
        d64 = oracle.v4_dot_f64(qa.download(np.uint8), sa.download(np.float32), qb.download(np.uint8), sb.download(np.float32))
        mag = float(np.abs(np.float32(0)) + abs(d64))
        assert abs(float(np.array([got], np.uint32).view(np.float32)[0]) - d64) <= 2e-5 * max(mag, 1.0) + 2e-6 * n
    ok(rt.hipGraphExecDestroy(gexec))
    ok(rt.hipGraphDestroy(graph))
    ok(rt.hipStreamDestroy(stream))
