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
    """One rank's membership in the RCCL communicator of the throughput-mode gather (sd_comm_* of include/specdec.h),
    created from the 128-byte ncclUniqueId every rank was handed (see _token_comm); the gather itself is ONE ncclAllGather
    issued by libspecdec on the current stream."""

    def __init__(self, rank: int, world: int, ident: bytes, device):
        import ctypes as C
        from ._lib import lib, check
        self.rank, self.world = rank, world
        self.device = torch.device(device)
        h = C.c_void_p()
        if self.device.type == "cuda":
            with torch.cuda.device(self.device):
                check(lib.sd_comm_init(self.rank, self.world, ident, C.byref(h)), "sd_comm_init")
        else:
            check(lib.sd_comm_init(self.rank, self.world, ident, C.byref(h)), "sd_comm_init")
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


def _agree(ok: bool, group, device) -> bool:
    """True iff `ok` holds on EVERY rank (one all_reduce that every rank always takes part in)."""
    import torch.distributed as dist
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
    return int(flag) == 1


def _token_comm(group, device) -> Optional["TokenComm"]:
    """The rank's TokenComm for this process group, or None when it could not be created on EVERY rank - in which case
    the caller gathers through torch.distributed instead.

    The SEQUENCE OF COLLECTIVES is the same on every rank whatever fails where (with a conditional broadcast a
    rank that failed early went on to the next collective while the others still sat in the broadcast - a hang in the
    very case the fallback exists for):
      1. every rank probes RCCL locally (dlopen + symbol lookup, no communicator) -> all_reduce(MIN) of the result;
         not resolvable somewhere: every rank returns None, no further collective;
      2. rank 0 asks for the unique id; the broadcast ALWAYS runs and carries None when that failed;
      3. ranks holding an id call sd_comm_init (ncclCommInitRank, itself collective: every rank is in it or none is);
         all_reduce(MIN) of its outcome; any failure: every rank destroys what it created and returns None."""
    import ctypes as C
    import sys
    import torch.distributed as dist
    from ._lib import lib
    key = (id(group), str(device))
    if key in _COMMS:
        return _COMMS[key]
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    comm, why = None, None
    try:
        probe_ok = lib.sd_comm_probe() == 0
        if not probe_ok:
            why = lib.sd_last_error().decode(errors="replace")
    except Exception as e:                                       # noqa: BLE001 - e.g. a library built without the entry
        probe_ok, why = False, repr(e)
    if _agree(probe_ok, group, device):
        ident = [None]
        if rank == 0:
            try:
                buf = (C.c_char * 128)()
                if lib.sd_comm_unique_id(buf) == 0:
                    ident[0] = bytes(buf)
                else:
                    why = lib.sd_last_error().decode(errors="replace")
            except Exception as e:                               # noqa: BLE001
                why = repr(e)
        dist.broadcast_object_list(ident, src=0, group=group)    # unconditional: None tells the others to skip the init
        if ident[0] is not None:
            try:
                comm = TokenComm(rank, world, ident[0], device)
            except Exception as e:                               # noqa: BLE001
                why = repr(e)
        elif why is None:
            why = "rank 0 could not create the unique id"
        if not _agree(comm is not None, group, device):
            if comm is not None:
                comm.close()
            comm = None
    if comm is None:
        print(f"[dist] rank {rank}: libspecdec's RCCL gather is unavailable ({why or 'failed on another rank'}); "
              "gathering through torch.distributed's all_gather instead", file=sys.stderr, flush=True)
    _COMMS[key] = comm
    return comm


def _use_own_collective(group, mine: torch.Tensor) -> bool:
    """libspecdec's ncclAllGather needs an RCCL process group and GPU buffers (gloo / CPU tests take torch's all_gather)."""
    import torch.distributed as dist
    return dist.get_backend(group) == "nccl" and mine.is_cuda


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
    comm = _token_comm(group, mine.device) if _use_own_collective(group, mine) else None
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
