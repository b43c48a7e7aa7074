"""Oracle for the tensor-parallel target (test infrastructure, see oracle/__init__.py).

The reference has no tensor parallelism; this restates its Llama forward
(sampling/models/modeling_llama.py:292-393, 405-457 with pretraining_tp = 1) for ONE Megatron shard: local heads, local MLP
columns, and an all-reduce of the o_proj / down_proj partial outputs.  Summing the shards' partials is algebraically the
unsharded forward, which is what the tests compare it with (oracle.models_ref.llama_forward on the full weights).
"""
from __future__ import annotations

import math
from typing import Callable, Dict, Optional

import torch
import torch.nn.functional as F

from .models_ref import KV, _causal_bias, _rms_norm, _rope_tables, _rot_half


def llama_forward_tp(cfg_local, sd: Dict[str, torch.Tensor], ids: torch.Tensor, past: Optional[KV],
                     all_reduce: Callable[[torch.Tensor], torch.Tensor]):
    """cfg_local: tp.shard_config(cfg, world); sd: this rank's slices (tp.shard_tensor).  all_reduce(t) returns the sum of
    t over the ranks (fp32 partials, rounded to the model dtype afterwards - one rounding, like an unsharded Linear)."""
    dt = sd["model.embed_tokens.weight"].dtype
    H, Hkv, D = cfg_local.num_attention_heads, cfg_local.num_key_value_heads, cfg_local.head_dim
    B, q_len = ids.shape
    n_past = past[0][0].shape[2] if past else 0
    x = F.embedding(ids, sd["model.embed_tokens.weight"])
    cos, sin = _rope_tables(D, n_past + q_len, cfg_local.rope_theta, dt)
    cos, sin = cos[n_past:n_past + q_len][None, None], sin[n_past:n_past + q_len][None, None]
    bias = _causal_bias(q_len, n_past, dt)[None, None]
    new_past: KV = []

    def row_parallel(inp, w):
        return all_reduce(F.linear(inp.float(), w.float())).to(dt)
    for li in range(cfg_local.num_hidden_layers):
        p = f"model.layers.{li}."
        h = _rms_norm(x, sd[p + "input_layernorm.weight"], cfg_local.rms_norm_eps)
        q = F.linear(h, sd[p + "self_attn.q_proj.weight"]).view(B, q_len, H, D).transpose(1, 2)
        k = F.linear(h, sd[p + "self_attn.k_proj.weight"]).view(B, q_len, Hkv, D).transpose(1, 2)
        v = F.linear(h, sd[p + "self_attn.v_proj.weight"]).view(B, q_len, Hkv, D).transpose(1, 2)
        q = q * cos + _rot_half(q) * sin
        k = k * cos + _rot_half(k) * sin
        if past:
            k = torch.cat([past[li][0], k], dim=2)
            v = torch.cat([past[li][1], v], dim=2)
        new_past.append((k, v))
        if H != Hkv:
            rep = H // Hkv
            k = k[:, :, None].expand(B, Hkv, rep, k.shape[2], D).reshape(B, H, -1, D)
            v = v[:, :, None].expand(B, Hkv, rep, v.shape[2], D).reshape(B, H, -1, D)
        s = torch.matmul(q, k.transpose(2, 3)) / math.sqrt(D) + bias
        a = F.softmax(s, dim=-1, dtype=torch.float32).to(dt)
        o = torch.matmul(a, v).transpose(1, 2).reshape(B, q_len, H * D)
        x = x + row_parallel(o, sd[p + "self_attn.o_proj.weight"])
        h = _rms_norm(x, sd[p + "post_attention_layernorm.weight"], cfg_local.rms_norm_eps)
        g = F.silu(F.linear(h, sd[p + "mlp.gate_proj.weight"])) * F.linear(h, sd[p + "mlp.up_proj.weight"])
        x = x + row_parallel(g, sd[p + "mlp.down_proj.weight"])
    x = _rms_norm(x, sd["model.norm.weight"], cfg_local.rms_norm_eps)
    return F.linear(x, sd["lm_head.weight"]).float(), new_past
