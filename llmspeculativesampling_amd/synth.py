"""Synthetic (random-init) weights in the HF state-dict naming.

No checkpoints exist offline, so benches and tests
build weights from a ModelConfig and a seed.  Two generators:

* ``method="numpy"``: numpy PCG64 streams keyed by (seed, tensor index).  numpy
  guarantees the bit stream across platforms, so the same tensors come out in
  this container and on the GPU box; the golden fixtures rely on that.
* ``method="torch"``: torch's generator on ``device`` (fast for 13 B parameters
  on the GPU).  Only reproducible on the same device type.
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import numpy as np
import torch

from .config import ModelConfig


def param_shapes(cfg: ModelConfig) -> List[Tuple[str, Tuple[int, ...], str]]:
    """(name, shape, kind) in a fixed order.  kind: mat | norm_w | bias | emb | pos."""
    h, L = cfg.hidden_size, cfg.num_hidden_layers
    out: List[Tuple[str, Tuple[int, ...], str]] = []
    if cfg.arch == "llama":
        kv = cfg.num_key_value_heads * cfg.head_dim
        qd = cfg.num_attention_heads * cfg.head_dim       # == h, except for a tensor-parallel shard's local geometry (tp.shard_config)
        out.append(("model.embed_tokens.weight", (cfg.vocab_size, h), "emb"))
        for i in range(L):
            p = f"model.layers.{i}."
            out += [
                (p + "self_attn.q_proj.weight", (qd, h), "mat"),
                (p + "self_attn.k_proj.weight", (kv, h), "mat"),
                (p + "self_attn.v_proj.weight", (kv, h), "mat"),
                (p + "self_attn.o_proj.weight", (h, qd), "mat"),
                (p + "mlp.gate_proj.weight", (cfg.intermediate_size, h), "mat"),
                (p + "mlp.up_proj.weight", (cfg.intermediate_size, h), "mat"),
                (p + "mlp.down_proj.weight", (h, cfg.intermediate_size), "mat"),
                (p + "input_layernorm.weight", (h,), "norm_w"),
                (p + "post_attention_layernorm.weight", (h,), "norm_w"),
            ]
        out.append(("model.norm.weight", (h,), "norm_w"))
        out.append(("lm_head.weight", (cfg.vocab_size, h), "head"))
        return out
    pd = cfg.word_embed_proj_dim
    d = "model.decoder."
    out.append((d + "embed_tokens.weight", (cfg.vocab_size, pd), "emb"))
    out.append((d + "embed_positions.weight", (cfg.max_position_embeddings + 2, h), "pos"))
    if pd != h:
        out.append((d + "project_in.weight", (h, pd), "mat"))
        out.append((d + "project_out.weight", (pd, h), "mat"))
    if cfg.do_layer_norm_before:
        out.append((d + "final_layer_norm.weight", (h,), "norm_w"))
        out.append((d + "final_layer_norm.bias", (h,), "bias"))
    for i in range(L):
        p = d + f"layers.{i}."
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            out.append((p + f"self_attn.{n}.weight", (h, h), "mat"))
            out.append((p + f"self_attn.{n}.bias", (h,), "bias"))
        out += [
            (p + "self_attn_layer_norm.weight", (h,), "norm_w"),
            (p + "self_attn_layer_norm.bias", (h,), "bias"),
            (p + "fc1.weight", (cfg.ffn_dim, h), "mat"),
            (p + "fc1.bias", (cfg.ffn_dim,), "bias"),
            (p + "fc2.weight", (h, cfg.ffn_dim), "mat"),
            (p + "fc2.bias", (h,), "bias"),
            (p + "final_layer_norm.weight", (h,), "norm_w"),
            (p + "final_layer_norm.bias", (h,), "bias"),
        ]
    return out


def _scale(kind: str, shape, gain: float, head_gain: float) -> Tuple[float, float]:
    """(mean, std) per tensor kind.  Matrices are fan-in scaled so activations
    stay O(1) through the stack; head_gain widens the logit spread so that the
    sampled distributions are not uniform over the vocabulary."""
    if kind == "mat":
        return 0.0, gain / np.sqrt(shape[-1])
    if kind == "head":
        return 0.0, head_gain / np.sqrt(shape[-1])
    if kind == "emb":
        return 0.0, 1.0
    if kind == "pos":
        return 0.0, 0.1
    if kind == "norm_w":
        return 1.0, 0.1
    if kind == "bias":
        return 0.0, 0.02
    raise ValueError(kind)


def make_state_dict(cfg: ModelConfig, seed: int, dtype: torch.dtype = torch.float32,
                    device: str = "cpu", method: str = "numpy", gain: float = 1.0,
                    head_gain: float = 4.0) -> Dict[str, torch.Tensor]:
    sd: Dict[str, torch.Tensor] = {}
    gen = None
    if method == "torch":
        gen = torch.Generator(device=device)
        gen.manual_seed(int(seed))
    for idx, (name, shape, kind) in enumerate(param_shapes(cfg)):
        mean, std = _scale(kind, shape, gain, head_gain)
        if method == "numpy":
            rng = np.random.default_rng([int(seed), idx])
            a = rng.standard_normal(size=shape, dtype=np.float32) * np.float32(std) + np.float32(mean)
            t = torch.from_numpy(a).to(device=device, dtype=dtype)
        else:
            t = torch.empty(shape, dtype=torch.float32 if max(shape) < 1 << 20 else dtype, device=device)
            t.normal_(mean, std, generator=gen)
            t = t.to(dtype)
        sd[name] = t
    if cfg.arch == "opt":
        # tied head (reference modeling_opt.py:833,840)
        sd["lm_head.weight"] = sd["model.decoder.embed_tokens.weight"]
    return sd


def perturb_state_dict(sd: Dict[str, torch.Tensor], seed: int, sigma: float) -> Dict[str, torch.Tensor]:
    """target = draft + sigma * noise (relative to each tensor's std): gives a
    model pair whose acceptance length spreads over 0..gamma."""
    out: Dict[str, torch.Tensor] = {}
    for idx, (name, t) in enumerate(sd.items()):
        if name == "lm_head.weight" and "model.decoder.embed_tokens.weight" in sd:
            continue
        rng = np.random.default_rng([int(seed), 7919, idx])
        n = torch.from_numpy(rng.standard_normal(size=tuple(t.shape), dtype=np.float32))
        s = float(t.float().std()) if t.numel() > 1 else 1.0
        out[name] = (t.float() + sigma * s * n).to(t.dtype)
    if "model.decoder.embed_tokens.weight" in out:
        out["lm_head.weight"] = out["model.decoder.embed_tokens.weight"]
    return out


# ------------------------------------------------------------------------------------------------------------------
# Acceptance dial: a draft / target pair of the NAMED architectures whose next-token
# distributions agree up to a knob, built without any checkpoint.  Every weight is still a dense random tensor that the
# forward streams and multiplies (same bytes, same flops as a real checkpoint of that shape).
#   * both models damp their residual branches (o_proj / down_proj scaled by eps), so the hidden state stays close to
#     the token embedding and the distribution is close to lm_head . rmsnorm(embed[token]);
#   * the target carries the base draft's embedding, final-norm weight and lm_head in its first `hd` hidden dims (the
#     head scaled by sqrt(hd / H) to undo RMSNorm's 1/sqrt(H) over the wider row) and zeros in the others;
#   * the knob sigma adds sigma * std * noise to the DRAFT's lm_head: sigma = 0 -> (almost) every draft accepted,
#     growing sigma -> acceptance falls to ~0.  One target serves every sigma; drafts are 68 M parameters each.
# ------------------------------------------------------------------------------------------------------------------
_RESID = ("self_attn.o_proj.weight", "mlp.down_proj.weight")


def dial_draft_transform(sigma: float, seed: int, resid_eps: float = 0.01):
    """transform for SpecDecModel.synthetic / make_state_dict-style generators of the llama DRAFT."""
    import torch as _t

    def tf(name: str, t):
        if name.endswith(_RESID):
            return t * resid_eps
        if name == "lm_head.weight" and sigma:
            g = _t.Generator(device=t.device).manual_seed(int(seed) * 7919 + 17)
            n = _t.empty(t.shape, dtype=_t.float32, device=t.device).normal_(0.0, 1.0, generator=g)
            return (t.float() + float(sigma) * float(t.float().std()) * n).to(t.dtype)
        return t
    return tf


def dial_target_transform(draft_get, draft_hidden: int, target_hidden: int, resid_eps: float = 0.002):
    """transform for the llama TARGET: ``draft_get(name)`` regenerates the sigma = 0 draft's tensors."""
    hd, H = int(draft_hidden), int(target_hidden)
    scale = float(np.sqrt(hd / H))

    def tf(name: str, t):
        if name.endswith(_RESID):
            return t * resid_eps
        if name == "model.embed_tokens.weight":
            t[:, hd:] = 0
            t[:, :hd] = draft_get(name).to(device=t.device, dtype=t.dtype)
        elif name == "model.norm.weight":
            t[:hd] = draft_get(name).to(device=t.device, dtype=t.dtype)
        elif name == "lm_head.weight":
            t[:, hd:] = 0
            t[:, :hd] = (draft_get(name).to(device=t.device).float() * scale).to(t.dtype)
        return t
    return tf
