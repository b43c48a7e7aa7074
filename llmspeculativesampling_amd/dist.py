"""Multi-GPU sharding of independent prompt streams.

Every stream is a self-contained batch-1 decode (reference speculative_sampling.py:1905, 1911-1912), so
ranks own disjoint streams and never talk inside the decode loop.  The single collective is the
throughput-mode gather of the generated ids at the end (RCCL over xGMI on the GPU box, gloo in the CPU tests).
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import torch


def shard_streams(n_streams: int, rank: int, world: int) -> List[int]:
    """Round-robin: stream s runs on rank s mod world."""
    return list(range(rank, n_streams, world))


def pack_outputs(outs: Sequence[torch.Tensor], width: int, device) -> torch.Tensor:
    """[n_local, width] int32, -1 padded; each row is one stream's (1, len) output."""
    buf = torch.full((len(outs), width), -1, dtype=torch.int32, device=device)
    for i, o in enumerate(outs):
        n = int(o.shape[-1])
        assert n <= width, (n, width)
        buf[i, :n] = o.reshape(-1).to(device=device, dtype=torch.int32)
    return buf


def gather_streams(outs: Sequence[torch.Tensor], n_streams: int, width: int, device=None,
                   group=None) -> List[torch.Tensor]:
    """all_gather of every rank's packed outputs; returns the streams in global order (stream s at index s),
    each trimmed of its padding.  Ranks must hold len(shard_streams(n_streams, rank, world)) outputs."""
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    device = device if device is not None else (outs[0].device if len(outs) else "cpu")
    per_rank = (n_streams + world - 1) // world
    mine = pack_outputs(outs, width, device)
    if mine.shape[0] < per_rank:                      # ragged tail: pad with an all -1 row
        pad = torch.full((per_rank - mine.shape[0], width), -1, dtype=torch.int32, device=device)
        mine = torch.cat([mine, pad], 0)
    gathered = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(gathered, mine, group=group)
    result: List[torch.Tensor] = []
    for s in range(n_streams):
        row = gathered[s % world][s // world]
        n = int((row >= 0).sum())
        result.append(row[:n].to(torch.int64))
    return result
