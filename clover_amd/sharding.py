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


def gather_packed(local: torch.Tensor, rows_total: int, group=None) -> torch.Tensor:
    """All-gather the per-rank packed results (uint8 tensors, possibly of different length) in rank order."""
    world = dist.get_world_size(group)
    sizes = [packed_bytes(partition_rows(rows_total, world, k)[1]) for k in range(world)]
    assert local.numel() == sizes[dist.get_rank(group)]
    if len(set(sizes)) == 1:
        out = torch.empty(sizes[0] * world, dtype=torch.uint8, device=local.device)
        dist.all_gather_into_tensor(out, local, group=group)
        return out
    # unequal shards (rows/64 not divisible by the world size): pad to the largest, gather, drop the pads
    mx = max(sizes)
    padded = torch.zeros(mx, dtype=torch.uint8, device=local.device)
    padded[: local.numel()] = local
    out = torch.empty(mx * world, dtype=torch.uint8, device=local.device)
    dist.all_gather_into_tensor(out, padded, group=group)
    return torch.cat([out[k * mx: k * mx + sizes[k]] for k in range(world)])


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
