"""Oracle decoder-only forwards (test infrastructure, see oracle/__init__.py).

Functional torch-CPU restatements of the two model families the reference runs
(its modeling files are HF-4.35.2 copies):

  Llama: reference sampling/models/modeling_llama.py:803-895 -> 624-768 -> 405-457 -> 292-393
  OPT:   reference sampling/models/modeling_opt.py:864-997 -> 561-759 -> 303-378 -> 160-278

Weights come in as an HF-named state dict; the KV cache is the reference's
tuple layout, one (k, v) pair of shape (B, H_kv, S, D) per layer (B = 1 on the
north-star path).  Every
rounding point of the reference (per-op results in the weight dtype, fp32
RMSNorm / softmax islands) is kept so that a bf16 run rounds where the
reference's would.
"""
from __future__ import annotations

import math
from types import SimpleNamespace
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

KV = List[Tuple[torch.Tensor, torch.Tensor]]


def _causal_bias(q_len: int, past: int, dtype: torch.dtype) -> torch.Tensor:
    """Additive mask, finfo.min above the diagonal (modeling_llama.py:43-57, modeling_opt.py:67-82)."""
    neg = torch.finfo(dtype).min
    i = torch.arange(q_len).view(q_len, 1) + past
    j = torch.arange(past + q_len).view(1, past + q_len)
    return torch.where(j <= i, torch.zeros((), dtype=dtype), torch.full((), neg, dtype=dtype))


# --------------------------------------------------------------------------- Llama

def _rms_norm(x, w, eps):
    # modeling_llama.py:84-89: fp32 statistics, cast back, then the weight multiply in x's dtype
    xf = x.to(torch.float32)
    xf = xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps)
    return w * xf.to(x.dtype)


def _rope_tables(head_dim: int, n_pos: int, theta: float, dtype):
    # modeling_llama.py:107-125: fp32 table, cast to the activation dtype on use
    inv = 1.0 / (theta ** (torch.arange(0, head_dim, 2).float() / head_dim))
    ang = torch.outer(torch.arange(n_pos, dtype=torch.float32), inv)
    ang = torch.cat((ang, ang), dim=-1)
    return ang.cos().to(dtype), ang.sin().to(dtype)


def _rot_half(x):
    h = x.shape[-1] // 2
    return torch.cat((-x[..., h:], x[..., :h]), dim=-1)


def _tree_bias(bias, extra_attention_mask, dtype):
    """Tree attention (modeling_llama.py:684-689, modeling_opt.py:660-667): finfo.min is ADDED to the causal bias of the
    last extra_cnt query rows wherever the extra mask is False."""
    if extra_attention_mask is None:
        return bias
    add = torch.where(extra_attention_mask, torch.zeros((), dtype=dtype), torch.full((), torch.finfo(dtype).min, dtype=dtype))
    n = extra_attention_mask.size(1)
    bias = bias.expand(extra_attention_mask.size(0), -1, -1, -1).clone()
    bias[:, :, -n:, :] += add[:, None, :, :]
    return bias


def _kv_fp8(t: torch.Tensor) -> torch.Tensor:
    """fp8 (OCP e4m3, scale 1) KV arena of BASELINE config 5 - NO reference counterpart (the reference keeps K / V in the
    model dtype): a new K / V row is stored as e4m3 (round to nearest even, clamped to +-448) and read back exactly.  The
    oracle applies the same quantisation where the row joins the cache, so that a 16-bit forward with an fp8 arena can be
    held to the same 'no further from the fp32 truth than the reference arithmetic' rule as one with a 16-bit arena."""
    return t.float().clamp(-448.0, 448.0).to(torch.float8_e4m3fn).to(t.dtype)


def llama_forward(cfg, sd: Dict[str, torch.Tensor], ids: torch.Tensor, past: Optional[KV],
                  extra_attention_mask=None, position_ids=None, kv_quant=None):
    dt = sd["model.embed_tokens.weight"].dtype
    H, Hkv, D = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim
    B, q_len = ids.shape          # B > 1 only for the width-w drafts of multi_speculative_sampling (oracle/multi_ref.py)
    n_past = past[0][0].shape[2] if past else 0
    x = F.embedding(ids, sd["model.embed_tokens.weight"])
    if position_ids is None:
        cos, sin = _rope_tables(D, n_past + q_len, cfg.rope_theta, dt)
        cos, sin = cos[n_past:n_past + q_len][None, None], sin[n_past:n_past + q_len][None, None]
    else:                                            # tree nodes: the position is the node's depth (modeling_llama.py:180-188)
        cos, sin = _rope_tables(D, int(position_ids.max()) + 1, cfg.rope_theta, dt)
        cos, sin = cos[position_ids][:, None], sin[position_ids][:, None]
    bias = _tree_bias(_causal_bias(q_len, n_past, dt)[None, None], extra_attention_mask, dt)
    new_past: KV = []
    for li in range(cfg.num_hidden_layers):
        p = f"model.layers.{li}."
        h = _rms_norm(x, sd[p + "input_layernorm.weight"], cfg.rms_norm_eps)
        q = F.linear(h, sd[p + "self_attn.q_proj.weight"]).view(B, q_len, H, D).transpose(1, 2)
        k = F.linear(h, sd[p + "self_attn.k_proj.weight"]).view(B, q_len, Hkv, D).transpose(1, 2)
        v = F.linear(h, sd[p + "self_attn.v_proj.weight"]).view(B, q_len, Hkv, D).transpose(1, 2)
        q = q * cos + _rot_half(q) * sin
        k = k * cos + _rot_half(k) * sin
        if kv_quant == "fp8":
            k, v = _kv_fp8(k), _kv_fp8(v)
        if past:
            k = torch.cat([past[li][0], k], dim=2)
            v = torch.cat([past[li][1], v], dim=2)
        new_past.append((k, v))
        if H != Hkv:                                   # repeat_kv, modeling_llama.py:225-234
            rep = H // Hkv
            k = k[:, :, None].expand(B, Hkv, rep, k.shape[2], D).reshape(B, H, -1, D)
            v = v[:, :, None].expand(B, Hkv, rep, v.shape[2], D).reshape(B, H, -1, D)
        s = torch.matmul(q, k.transpose(2, 3)) / math.sqrt(D)      # scale after the matmul (:346)
        s = s + bias
        a = F.softmax(s, dim=-1, dtype=torch.float32).to(dt)        # fp32 softmax, cast back (:371)
        o = torch.matmul(a, v).transpose(1, 2).reshape(B, q_len, H * D)
        x = x + F.linear(o, sd[p + "self_attn.o_proj.weight"])
        h = _rms_norm(x, sd[p + "post_attention_layernorm.weight"], cfg.rms_norm_eps)
        g = F.silu(F.linear(h, sd[p + "mlp.gate_proj.weight"])) * F.linear(h, sd[p + "mlp.up_proj.weight"])
        x = x + F.linear(g, sd[p + "mlp.down_proj.weight"])
    x = _rms_norm(x, sd["model.norm.weight"], cfg.rms_norm_eps)
    logits = F.linear(x, sd["lm_head.weight"]).float()             # always fp32 (:870)
    return logits, new_past


# --------------------------------------------------------------------------- OPT

def opt_forward(cfg, sd: Dict[str, torch.Tensor], ids: torch.Tensor, past: Optional[KV],
                extra_attention_mask=None, position_ids=None):
    d = "model.decoder."
    dt = sd[d + "embed_tokens.weight"].dtype
    H, D, hid = cfg.num_attention_heads, cfg.head_dim, cfg.hidden_size
    B, q_len = ids.shape
    n_past = past[0][0].shape[2] if past else 0
    x = F.embedding(ids, sd[d + "embed_tokens.weight"])
    # learned positions, offset 2 (modeling_opt.py:98-124); an all-ones mask makes them n_past..n_past+q-1
    pos = (torch.arange(n_past, n_past + q_len)[None] if position_ids is None else position_ids) + 2
    pe = F.embedding(pos, sd[d + "embed_positions.weight"])
    if d + "project_in.weight" in sd:
        x = F.linear(x, sd[d + "project_in.weight"])
    x = x + pe
    bias = _tree_bias(_causal_bias(q_len, n_past, dt)[None, None], extra_attention_mask, dt)
    pre = cfg.do_layer_norm_before
    eps = cfg.layer_norm_eps
    scaling = D ** -0.5
    new_past: KV = []
    for li in range(cfg.num_hidden_layers):
        p = d + f"layers.{li}."
        res = x
        h = F.layer_norm(x, (hid,), sd[p + "self_attn_layer_norm.weight"], sd[p + "self_attn_layer_norm.bias"], eps) if pre else x
        q = F.linear(h, sd[p + "self_attn.q_proj.weight"], sd[p + "self_attn.q_proj.bias"]) * scaling   # pre-scaled q (:178)
        k = F.linear(h, sd[p + "self_attn.k_proj.weight"], sd[p + "self_attn.k_proj.bias"])
        v = F.linear(h, sd[p + "self_attn.v_proj.weight"], sd[p + "self_attn.v_proj.bias"])
        q = q.view(B, q_len, H, D).transpose(1, 2)
        k = k.view(B, q_len, H, D).transpose(1, 2)
        v = v.view(B, q_len, H, D).transpose(1, 2)
        if past:
            k = torch.cat([past[li][0], k], dim=2)
            v = torch.cat([past[li][1], v], dim=2)
        new_past.append((k, v))
        s = torch.bmm(q.reshape(B * H, q_len, D), k.reshape(B * H, -1, D).transpose(1, 2)).view(B, H, q_len, -1)
        s = torch.max(s + bias, torch.tensor(torch.finfo(dt).min))  # mask then clamp (:228-231)
        if dt == torch.float16:                                     # fp32 softmax only for fp16 (:235-238)
            a = F.softmax(s, dim=-1, dtype=torch.float32).to(dt)
        else:
            a = F.softmax(s, dim=-1)
        o = torch.bmm(a.view(B * H, q_len, -1), v.reshape(B * H, -1, D)).view(B, H, q_len, D)
        o = o.transpose(1, 2).reshape(B, q_len, hid)
        x = res + F.linear(o, sd[p + "self_attn.out_proj.weight"], sd[p + "self_attn.out_proj.bias"])
        if not pre:
            x = F.layer_norm(x, (hid,), sd[p + "self_attn_layer_norm.weight"], sd[p + "self_attn_layer_norm.bias"], eps)
        res = x
        h = F.layer_norm(x, (hid,), sd[p + "final_layer_norm.weight"], sd[p + "final_layer_norm.bias"], eps) if pre else x
        h = F.relu(F.linear(h, sd[p + "fc1.weight"], sd[p + "fc1.bias"]))
        x = res + F.linear(h, sd[p + "fc2.weight"], sd[p + "fc2.bias"])
        if not pre:
            x = F.layer_norm(x, (hid,), sd[p + "final_layer_norm.weight"], sd[p + "final_layer_norm.bias"], eps)
    if d + "final_layer_norm.weight" in sd:
        x = F.layer_norm(x, (hid,), sd[d + "final_layer_norm.weight"], sd[d + "final_layer_norm.bias"], eps)
    if d + "project_out.weight" in sd:
        x = F.linear(x, sd[d + "project_out.weight"])
    logits = F.linear(x, sd["lm_head.weight"]).contiguous()        # stays in the weight dtype (:974)
    return logits, new_past


class RefCausalLM:
    """Callable with the surface the reference touches: ``model(ids, past_key_values=, use_cache=)``
    -> ``.logits`` / ``.past_key_values``; ``.config.is_encoder_decoder``; ``.device``."""

    def __init__(self, cfg, state_dict: Dict[str, torch.Tensor], kv_quant: Optional[str] = None):
        assert kv_quant in (None, "fp8") and (kv_quant is None or cfg.arch == "llama")
        self.kv_quant = kv_quant                      # "fp8": the e4m3 KV arena of config 5 (_kv_fp8; not in the reference)
        self.cfg = cfg
        self.config = SimpleNamespace(is_encoder_decoder=False, vocab_size=cfg.vocab_size)
        self.sd = state_dict
        self.device = torch.device("cpu")
        self.n_calls = 0

    @torch.no_grad()
    def __call__(self, input_ids, past_key_values=None, use_cache=True, extra_attention_mask=None, position_ids=None, **_):
        fwd = llama_forward if self.cfg.arch == "llama" else opt_forward
        kw = {"kv_quant": self.kv_quant} if self.kv_quant else {}
        logits, kv = fwd(self.cfg, self.sd, input_ids, past_key_values, extra_attention_mask, position_ids, **kw)
        self.n_calls += 1
        return SimpleNamespace(logits=logits, past_key_values=kv)
