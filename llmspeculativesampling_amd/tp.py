"""Tensor parallelism of the TARGET model over the GPUs of one node (BASELINE config 5: Llama-2-70b, TP = 8 over xGMI).

The reference is a single process; its only multi-GPU mode is accelerate's sequential layer placement
(evaluation.py:186).  Here every rank holds a Megatron slice of each decoder layer and runs the SAME decode loop on the
same tokens: q / k / v and gate / up projections are split by output rows (whole heads per rank: Llama-2-70b's 8 KV heads
give one per rank at TP = 8), o_proj and down_proj by input columns, and their fp32 partial outputs are all-reduced
(csrc/engine.hip: tp_reduce, two per layer).  Embedding, norms and lm_head are replicated; so are the draft model and all
of the sampling, which is why the ranks stay token-identical without exchanging tokens.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import replace
from typing import Callable, List

import torch

from ._lib import lib, check
from .config import ModelConfig


def shard_config(cfg: ModelConfig, world: int) -> ModelConfig:
    """The LOCAL geometry of one of `world` shards: hidden stays, heads / KV heads / MLP width are divided."""
    if cfg.arch != "llama":
        raise NotImplementedError("tensor parallelism is implemented for the Llama family (config 5)")
    if cfg.num_attention_heads % world or cfg.num_key_value_heads % world or cfg.intermediate_size % (world * 32):
        raise ValueError(f"{cfg.num_attention_heads} heads / {cfg.num_key_value_heads} KV heads / "
                         f"{cfg.intermediate_size} MLP columns do not split over {world} ranks")
    return replace(cfg, num_attention_heads=cfg.num_attention_heads // world,
                   num_key_value_heads=cfg.num_key_value_heads // world,
                   intermediate_size=cfg.intermediate_size // world, head_dim_=cfg.head_dim,
                   name=f"{cfg.name}[tp{world}]")


def shard_tensor(cfg: ModelConfig, name: str, t: torch.Tensor, rank: int, world: int) -> torch.Tensor:
    """Rank `rank`'s slice of a full HF-named Llama tensor."""
    def rows(x):
        n = x.shape[0] // world
        return x[rank * n:(rank + 1) * n]

    def cols(x):
        n = x.shape[1] // world
        return x[:, rank * n:(rank + 1) * n]
    if name.endswith(("q_proj.weight", "k_proj.weight", "v_proj.weight", "gate_proj.weight", "up_proj.weight")):
        return rows(t).contiguous()                 # heads (resp. MLP columns) are contiguous row blocks
    if name.endswith(("o_proj.weight", "down_proj.weight")):
        return cols(t).contiguous()
    return t


def sharded_getter(cfg: ModelConfig, get: Callable[[str], torch.Tensor], rank: int, world: int):
    return lambda name: shard_tensor(cfg, name, get(name), rank, world)


class TPGroup:
    """Handle of one rank's membership in a tensor-parallel group (sd_tp)."""

    def __init__(self, handle, rank: int, world: int):
        self.handle, self.rank, self.world = handle, rank, world

    @classmethod
    def rccl(cls, rank: int, world: int, broadcast: Callable[[bytes], bytes]) -> "TPGroup":
        """`broadcast(id_or_empty)` must return rank 0's 128-byte id on every rank (e.g. torch.distributed
        broadcast_object_list over the bench's process group)."""
        buf = (C.c_char * 128)()
        if rank == 0:
            check(lib.sd_tp_unique_id(buf), "sd_tp_unique_id")
        uid = broadcast(bytes(buf) if rank == 0 else b"")
        assert len(uid) == 128
        h = C.c_void_p()
        check(lib.sd_tp_create_rccl(rank, world, uid, C.byref(h)), "sd_tp_create_rccl")
        return cls(h, rank, world)

    @classmethod
    def loopback(cls, world: int) -> List["TPGroup"]:
        arr = (C.c_void_p * world)()
        check(lib.sd_tp_create_loopback(world, arr), "sd_tp_create_loopback")
        return [cls(C.c_void_p(arr[r]), r, world) for r in range(world)]

    def bind(self, session) -> None:
        check(lib.sd_session_set_tp(session.handle, self.handle), "sd_session_set_tp")
        session._tp = self                           # keeps the group alive as long as the session

    def __del__(self):
        h = getattr(self, "handle", None)
        if h and lib is not None:
            lib.sd_tp_destroy(h)
            self.handle = None


_SHARDED = ("q_proj.weight", "k_proj.weight", "v_proj.weight", "o_proj.weight", "gate_proj.weight", "up_proj.weight",
            "down_proj.weight")


def synthetic_shard(cfg: ModelConfig, rank: int, world: int, seed: int, group: "TPGroup" = None, **kw):
    """Random-init shard `rank` of `cfg` generated directly at the local shapes (no full-size tensor ever exists: the
    70b model is 140 GB in bf16).  Replicated tensors (embedding, norms, lm_head) use the same seed on every rank, sliced
    ones a per-rank seed - statistically a slice of one random model."""
    from .engine import SpecDecModel
    local = shard_config(cfg, world)
    m = SpecDecModel.synthetic(local, seed=seed, seed_of=lambda n: seed * 131 + 1 + rank if n.endswith(_SHARDED) else seed, **kw)
    m.full_cfg = cfg
    m.tp_group = group
    return m


def shard_model(cfg: ModelConfig, sd, rank: int, world: int, group: "TPGroup" = None, **kw):
    """Shard `rank` of a full HF-named Llama state dict."""
    from .engine import SpecDecModel
    m = SpecDecModel(shard_config(cfg, world), sharded_getter(cfg, lambda n: sd[n], rank, world), **kw)
    m.full_cfg = cfg
    m.tp_group = group
    return m
