"""Row sharding of CloverMatrix4::mvm across ranks (one process per GPU, torch.distributed).

Host-side logic shared by bench.py and the tests: contiguous shards in units of 64 rows (the same split
clm4_shard_partition / the reference's mvm_parallel use, CloverMatrix4.h:1700-1705) and the packed-result
exchange.  The exchange is an all-gather of [rows_k/2 nibble bytes | rows_k/64 fp32 scales] per rank -- never
an all-reduce, which would change the fp32 summation order.  Works on any backend (nccl = RCCL on the GPU
box, gloo in the CPU tests).
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def partition_rows(rows: int, nparts: int, part: int) -> tuple[int, int]:
    """(row_begin, row_count) of shard `part`: multiples of 64, remainder spread over the first ranks."""
    assert rows % 64 == 0 and 0 <= part < nparts
    blocks = rows // 64
    base, extra = divmod(blocks, nparts)
    b0 = part * base + min(part, extra)
    return b0 * 64, (base + (1 if part < extra else 0)) * 64


def packed_bytes(rows: int) -> int:
    """bytes of a packed CloverVector4 result of `rows` rows: nibbles + scales."""
    return rows // 2 + 4 * (rows // 64)


class PendingGather:
    """An all-gather in flight (async_op=True).  wait() orders the caller's stream behind it and returns the gathered
    bytes in rank order; until then neither the input nor the output buffer may be reused."""

    def __init__(self, work, out: torch.Tensor, sizes: list[int], keep_alive):
        self._work, self._out, self._sizes, self._keep = work, out, sizes, keep_alive

    def wait(self) -> torch.Tensor:
        if self._work is not None:
            self._work.wait()
            self._work = None
        mx = max(self._sizes)
        if len(set(self._sizes)) == 1:
            return self._out
        return torch.cat([self._out[k * mx: k * mx + self._sizes[k]] for k in range(len(self._sizes))])


def gather_packed_async(local: torch.Tensor, rows_total: int, group=None) -> PendingGather:
    """Start the all-gather of the per-rank packed results (uint8 tensors, possibly of different length).  On the NCCL (RCCL)
    backend the collective runs on the communicator's own stream behind the work already queued on the current stream, so
    the next kernel on the current stream overlaps it."""
    world = dist.get_world_size(group)
    sizes = [packed_bytes(partition_rows(rows_total, world, k)[1]) for k in range(world)]
    assert local.numel() == sizes[dist.get_rank(group)]
    mx = max(sizes)
    src = local
    if len(set(sizes)) != 1:
        # unequal shards (rows/64 not divisible by the world size): pad to the largest, gather, drop the pads in wait()
        src = torch.zeros(mx, dtype=torch.uint8, device=local.device)
        src[: local.numel()] = local
    out = torch.empty(mx * world, dtype=torch.uint8, device=local.device)
    work = dist.all_gather_into_tensor(out, src, group=group, async_op=True)
    return PendingGather(work, out, sizes, src)


def gather_packed(local: torch.Tensor, rows_total: int, group=None) -> torch.Tensor:
    """All-gather the per-rank packed results in rank order (blocking form of gather_packed_async)."""
    return gather_packed_async(local, rows_total, group).wait()


def unpack_gathered(buf: torch.Tensor, rows_total: int, world: int) -> tuple[torch.Tensor, torch.Tensor]:
    """[r_0|s_0|r_1|s_1|...] -> (all nibble bytes, all scales as fp32) of the full result vector."""
    nib, sc, off = [], [], 0
    for k in range(world):
        rk = partition_rows(rows_total, world, k)[1]
        nib.append(buf[off: off + rk // 2])
        sc.append(buf[off + rk // 2: off + packed_bytes(rk)])
        off += packed_bytes(rk)
    scales = torch.cat(sc).contiguous().view(torch.float32)
    return torch.cat(nib).contiguous(), scales


# ---- GEMM: C = A * B^T sharded by rows of A (B replicated) ----------------------------------------------------------------
# Every element of C is one fma chain over the K-blocks of its own row of A and row of B, so a rank that holds
# rows [b, b + c) of A computes rows [b, b + c) of C bit for bit as the unsharded call would.  Shards are multiples of 128
# rows (clm4_gemm's unit); the only exchange, if the whole C is wanted in one place, is an all-gather of fp32 rows.

def partition_gemm_rows(rows: int, nparts: int, part: int) -> tuple[int, int]:
    """(row_begin, row_count) of shard `part` of A / C: multiples of 128, remainder spread over the first ranks."""
    assert rows % 128 == 0 and 0 <= part < nparts
    blocks = rows // 128
    base, extra = divmod(blocks, nparts)
    b0 = part * base + min(part, extra)
    return b0 * 128, (base + (1 if part < extra else 0)) * 128


def exchange_gemm_rows(c_local: torch.Tensor, rows_total: int, ncols: int, mode: str = "all_gather", group=None):
    """What happens to the per-rank row shards of C after a step, the three modes of clm4_sharded_gemm_begin_mode (multi.hip):
    "all_gather" -> every rank returns the whole C; "gather_root" -> rank 0 returns the whole C, the others their own panel (they send
    it and receive nothing); "sharded" -> no exchange, every rank returns its own panel."""
    if mode == "all_gather":
        return gather_gemm_rows(c_local, rows_total, ncols, group)
    if mode == "sharded":
        return c_local
    assert mode == "gather_root", mode
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    counts = [partition_gemm_rows(rows_total, world, k)[1] for k in range(world)]
    if rank == 0:
        parts = [c_local.reshape(-1)] + [torch.empty(counts[k] * ncols, dtype=torch.float32, device=c_local.device) for k in range(1, world)]
        for k in range(1, world):
            dist.recv(parts[k], src=k, group=group)
        return torch.cat(parts).view(rows_total, ncols)
    dist.send(c_local.reshape(-1).contiguous(), dst=0, group=group)
    return c_local


def gather_gemm_rows(c_local: torch.Tensor, rows_total: int, ncols: int, group=None) -> torch.Tensor:
    """All-gather the per-rank row shards of C (fp32, [rows_k, ncols]) in rank order -> [rows_total, ncols]."""
    world = dist.get_world_size(group)
    counts = [partition_gemm_rows(rows_total, world, k)[1] for k in range(world)]
    assert c_local.dtype == torch.float32 and c_local.numel() == counts[dist.get_rank(group)] * ncols
    mx = max(counts)
    src = c_local.reshape(-1)
    if len(set(counts)) != 1:                      # unequal shards: pad to the largest, drop the pads afterwards
        src = torch.zeros(mx * ncols, dtype=torch.float32, device=c_local.device)
        src[: c_local.numel()] = c_local.reshape(-1)
    out = torch.empty(mx * ncols * world, dtype=torch.float32, device=c_local.device)
    dist.all_gather_into_tensor(out, src, group=group)
    if len(set(counts)) == 1:
        return out.view(rows_total, ncols)
    return torch.cat([out[k * mx * ncols: k * mx * ncols + counts[k] * ncols] for k in range(world)]).view(rows_total, ncols)
