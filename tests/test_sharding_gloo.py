"""Multi-rank path on CPU: world_size-2 (and 3, unequal shards) gloo runs of the sharding logic bench.py
uses on the GPUs.  Each rank computes its row shard with the CPU oracle standing in for the HIP kernel (test
infrastructure only), gathers the packed result, and every rank must hold the unsharded answer byte for byte."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

from clover_amd.sharding import (exchange_gemm_rows, gather_gemm_rows, gather_packed, packed_bytes, partition_gemm_rows, partition_rows,  # noqa: E402
                                 unpack_gathered)


def test_partition_matches_c_abi_and_covers_rows():
    import ctypes as C
    from clover_amd.lib_binding import load_library
    lib = load_library()
    for rows in (128, 640, 65536, 1 << 20):
        for n in (1, 2, 3, 8):
            if rows // 64 < n:
                continue
            got = [partition_rows(rows, n, k) for k in range(n)]
            assert got[0][0] == 0 and sum(c for _, c in got) == rows
            for k in range(n):
                b, c = C.c_uint64(), C.c_uint64()
                assert lib.clm4_shard_partition(rows, n, k, C.byref(b), C.byref(c)) == 0
                assert (b.value, c.value) == got[k] and c.value % 64 == 0
                if k:
                    assert got[k][0] == got[k - 1][0] + got[k - 1][1]


def _worker(rank, world, port, rows, cols, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle.binding import Oracle
        orc = Oracle()
        rng = np.random.default_rng(123)                      # same data on every rank
        q = rng.integers(-7, 8, size=rows * cols).astype(np.int8)
        qA = (((q[0::2].astype(np.uint8) & 0xF) << 4) | (q[1::2].astype(np.uint8) & 0xF)).astype(np.uint8)
        sA = rng.uniform(0.5, 2, size=(rows // 64) * (cols // 64)).astype(np.float32)
        qx = rng.integers(0, 256, size=cols // 2, dtype=np.uint8) & 0x77
        sx = rng.uniform(0.5, 2, size=cols // 64).astype(np.float32)
        b, c = partition_rows(rows, world, rank)
        hb = cols // 64
        r, sr = orc.m4_mvm(qA[b * cols // 2:(b + c) * cols // 2], sA[(b // 64) * hb:((b + c) // 64) * hb], c, cols, qx, sx)
        local = torch.from_numpy(np.concatenate([r, sr.view(np.uint8)]))
        assert local.numel() == packed_bytes(c)
        full = gather_packed(local, rows, None)
        nib, scales = unpack_gathered(full, rows, world)
        r_ref, sr_ref = orc.m4_mvm(qA, sA, rows, cols, qx, sx)
        ok = np.array_equal(nib.numpy(), r_ref) and np.array_equal(scales.numpy().view(np.uint32), sr_ref.view(np.uint32))
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world,rows", [(2, 256), (3, 640)])
def test_sharded_mvm_gloo(world, rows):
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(k, world, port, rows, 256, ret)) for k in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert all(ret.get(k) for k in range(world)), dict(ret)


def _gemm_worker(rank, world, port, M, N, K, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle.binding import Oracle
        orc = Oracle()
        rng = np.random.default_rng(321)                      # same data on every rank
        qA = rng.integers(0, 256, size=M * K // 2, dtype=np.uint8) & 0x77
        qB = rng.integers(0, 256, size=N * K // 2, dtype=np.uint8) & 0x77
        sA = rng.uniform(0.5, 2, size=(M // 64) * (K // 64)).astype(np.float32)
        sB = rng.uniform(0.5, 2, size=(N // 64) * (K // 64)).astype(np.float32)
        b, c = partition_gemm_rows(M, world, rank)
        kb = K // 64
        C_loc = orc.m4_gemm(qA[b * K // 2:(b + c) * K // 2], sA[(b // 64) * kb:((b + c) // 64) * kb], c, K, qB, sB, N)
        full = gather_gemm_rows(torch.from_numpy(np.ascontiguousarray(C_loc)).reshape(c, N), M, N)
        ref = orc.m4_gemm(qA, sA, M, K, qB, sB, N).reshape(M, N)
        ok = bool(np.array_equal(full.numpy().view(np.uint32), ref.view(np.uint32)))
        # the other two exchange modes (clm4_sharded_gemm_begin_mode): the root ends with the whole C, everybody else with its own panel
        loc = torch.from_numpy(np.ascontiguousarray(C_loc)).reshape(c, N)
        for mode in ("gather_root", "sharded"):
            got = exchange_gemm_rows(loc, M, N, mode).numpy().view(np.uint32)
            want = ref.view(np.uint32) if (mode == "gather_root" and rank == 0) else ref.view(np.uint32)[b:b + c]
            ok = ok and got.shape == want.shape and bool(np.array_equal(got, want))
        ret[rank] = ok
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,M", [(2, 256), (3, 512)])
def test_sharded_gemm_gloo(world, M):
    """row-sharded GEMM (equal and unequal shards): the gathered C equals the unsharded one bit for bit"""
    assert [partition_gemm_rows(M, world, k) for k in range(world)][-1][0] + partition_gemm_rows(M, world, world - 1)[1] == M
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_gemm_worker, args=(k, world, port, M, 128, 256, ret)) for k in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert all(ret.get(k) for k in range(world)), dict(ret)
