#!/usr/bin/env python3
"""Quantized IHT on a synthetic compressed-sensing problem, the reference's generator (test/performance/03_iht_gd_util.cpp:449-495):
Phi uniform(-1,1), K-sparse x of ones, y = Phi x in fp32; then Q_IHT with a 4-bit Phi and 8-bit vectors, everything on the GPU.
Prints the recovered support overlap per step size mu (the reference grid-searches mu: 03_iht_gd_util.h:300-335)."""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from clover_amd.lib_binding import CloverHip  # noqa: E402

hip = CloverHip()
lib = hip.lib
m, n, K, iters = 1024, 2048, 32, 60
rng = np.random.default_rng(7)
Phi = rng.uniform(-1, 1, size=(m, n)).astype(np.float32)
x_true = np.zeros(n, np.float32)
x_true[rng.permutation(n)[:K]] = 1.0
y = Phi @ x_true

dPhi32 = hip.to_device(Phi)
qPhi, sPhi = hip.alloc(m * n // 2), hip.alloc((m // 64) * (n // 64) * 4)
hip.check(lib.clm4_quantize(dPhi32.ptr, m, n, qPhi.ptr, sPhi.ptr, None, None))
qPhiT, sPhiT = hip.alloc(m * n // 2), hip.alloc((m // 64) * (n // 64) * 4)
hip.check(lib.clm4_transpose(qPhi.ptr, sPhi.ptr, m, n, qPhiT.ptr, sPhiT.ptr, None))
dy32 = hip.to_device(y)
qy, sy = hip.alloc(m), hip.alloc(m // 16)
hip.check(lib.clv8_quantize(dy32.ptr, m, qy.ptr, sy.ptr, None, None))
bufs = [hip.alloc(k) for k in (n, n // 16, m, m // 16, m, m // 16, n, n // 16)]
xr = hip.alloc(4 * n)
for mu in (0.5 / m, 1.0 / m, 2.0 / m, 3.0 / m):
    hip.check(lib.clm4_iht_v8(qPhi.ptr, sPhi.ptr, qPhiT.ptr, sPhiT.ptr, m, n, bufs[0].ptr, bufs[1].ptr, n, qy.ptr, sy.ptr, bufs[2].ptr,
                              bufs[3].ptr, bufs[4].ptr, bufs[5].ptr, bufs[6].ptr, bufs[7].ptr, iters, K, mu, 1, None, None))
    hip.check(lib.clv8_restore(bufs[0].ptr, bufs[1].ptr, n, xr.ptr, None))
    x = xr.download(np.float32, n)
    support = set(np.argsort(-np.abs(x))[:K].tolist()) if np.any(x) else set()
    hit = len(support & set(np.flatnonzero(x_true).tolist()))
    err = float(np.linalg.norm(x - x_true) / np.linalg.norm(x_true))
    print(f"mu = {mu:.5f}: support recovered {hit}/{K}, relative error {err:.3f}, nonzeros {int(np.count_nonzero(x))}")
