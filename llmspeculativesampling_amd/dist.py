"""Multi-GPU sharding of independent prompt streams.

Every stream is a self-contained batch-1 decode (reference speculative_sampling.py:1905, 1911-1912), so
ranks own disjoint streams and never talk inside the decode loop.  The single collective is the
throughput-mode gather of the generated ids at the end (RCCL over xGMI on the GPU box, gloo in the CPU tests).
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

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


class TokenComm:
    """One rank's membership in the RCCL communicator of the throughput-mode gather (sd_comm_* of include/specdec.h).
    The 128-byte ncclUniqueId travels from rank 0 to the others over the existing torch.distributed group (its store /
    broadcast is the side channel); the gather itself is ONE ncclAllGather issued by libspecdec on the current stream."""

    def __init__(self, group=None, device=None):
        import ctypes as C
        import torch.distributed as dist
        from ._lib import lib, check
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        ident = [None]
        if self.rank == 0:
            buf = (C.c_char * 128)()
            check(lib.sd_comm_unique_id(buf), "sd_comm_unique_id")
            ident[0] = bytes(buf)
        dist.broadcast_object_list(ident, src=0, group=group)
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            check(lib.sd_comm_init(self.rank, self.world, ident[0], C.byref(h)), "sd_comm_init")
        self.handle = h

    def all_gather_tokens(self, mine: torch.Tensor) -> torch.Tensor:
        """mine: [rows][width] int32 on this rank's GPU -> [world][rows][width]."""
        from ._lib import lib, check
        assert mine.dtype == torch.int32 and mine.is_cuda and mine.is_contiguous()
        out = torch.empty((self.world,) + tuple(mine.shape), dtype=torch.int32, device=mine.device)
        check(lib.sd_comm_all_gather_tokens(self.handle, mine.data_ptr(), out.data_ptr(), mine.shape[0], mine.shape[1],
                                            torch.cuda.current_stream(mine.device).cuda_stream), "sd_comm_all_gather_tokens")
        return out

    def close(self):
        from ._lib import lib
        if getattr(self, "handle", None):
            lib.sd_comm_destroy(self.handle)
            self.handle = None

    __del__ = close


_COMMS = {}


def _token_comm(group, device) -> Optional["TokenComm"]:
    """The rank's TokenComm for this process group, or None when it could not be created on EVERY rank (the ranks agree
    through one all_reduce, so they all take the same collective; the reason goes to stderr)."""
    import sys
    import torch.distributed as dist
    key = (id(group), str(device))
    if key not in _COMMS:
        comm, err = None, None
        try:
            comm = TokenComm(group, device)
        except Exception as e:                                   # e.g. librccl not resolvable from libspecdec
            err = e
        ok = torch.tensor([1 if comm is not None else 0], dtype=torch.int32, device=device)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
        if int(ok) == 0:
            if err is not None:
                print(f"[dist] sd_comm_init failed on this rank ({type(err).__name__}: {err}); "
                      "gathering through torch.distributed's RCCL all_gather instead", file=sys.stderr, flush=True)
            if comm is not None:
                comm.close()
            comm = None
        _COMMS[key] = comm
    return _COMMS[key]


def gather_streams(outs: Sequence[torch.Tensor], n_streams: int, width: int, device=None,
                   group=None) -> List[torch.Tensor]:
    """all_gather of every rank's packed outputs; returns the streams in global order (stream s at index s),
    each trimmed of its padding.  Ranks must hold len(shard_streams(n_streams, rank, world)) outputs.
    With an RCCL ("nccl") process group and GPU buffers the collective is libspecdec's own ncclAllGather
    (sd_comm_all_gather_tokens); otherwise (gloo, CPU tests) torch.distributed's all_gather."""
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    device = device if device is not None else (outs[0].device if len(outs) else "cpu")
    per_rank = (n_streams + world - 1) // world
    mine = pack_outputs(outs, width, device)
    if mine.shape[0] < per_rank:                      # ragged tail: pad with an all -1 row
        pad = torch.full((per_rank - mine.shape[0], width), -1, dtype=torch.int32, device=device)
        mine = torch.cat([mine, pad], 0)
    comm = _token_comm(group, mine.device) if dist.get_backend(group) == "nccl" and mine.is_cuda else None
    if comm is not None:
        gathered = comm.all_gather_tokens(mine.contiguous())
    else:
        gathered = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(gathered, mine, group=group)
    result: List[torch.Tensor] = []
    for s in range(n_streams):
        row = gathered[s % world][s // world]
        n = int((row >= 0).sum())
        result.append(row[:n].to(torch.int64))
    return result
