"""Model geometry for the decode engine.

The reference never stores model geometry: it loads by hub name
(reference evaluation.py:183-253).  SURVEY.md section 8 lists the constants of
the model cards the BASELINE configs name; they live as JSON next to this file
so nothing has to be fetched.
"""
from __future__ import annotations

import json
import os
from dataclasses import dataclass, asdict
from typing import Optional

_CFG_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "configs")

ARCH_LLAMA = 0
ARCH_OPT = 1


@dataclass
class ModelConfig:
    arch: str                      # "llama" | "opt"
    vocab_size: int
    hidden_size: int
    num_hidden_layers: int
    num_attention_heads: int
    max_position_embeddings: int
    # llama
    intermediate_size: int = 0
    num_key_value_heads: int = 0
    rms_norm_eps: float = 1e-6
    rope_theta: float = 10000.0
    # opt
    ffn_dim: int = 0
    do_layer_norm_before: bool = True
    word_embed_proj_dim: int = 0
    layer_norm_eps: float = 1e-5   # nn.LayerNorm default, reference modeling_opt.py:296-301
    # the reference branches on this attribute (speculative_sampling.py:1942,1955)
    is_encoder_decoder: bool = False
    name: str = ""
    # width of one attention head when it is not hidden_size / num_attention_heads: a tensor-parallel shard keeps the
    # model's hidden size but only a slice of its heads (tp.py)
    head_dim_: int = 0

    def __post_init__(self):
        if self.arch == "llama":
            if not self.num_key_value_heads:
                self.num_key_value_heads = self.num_attention_heads
        elif self.arch == "opt":
            self.num_key_value_heads = self.num_attention_heads
            if not self.word_embed_proj_dim:
                self.word_embed_proj_dim = self.hidden_size
            self.intermediate_size = self.ffn_dim
        else:
            raise ValueError(f"unknown arch {self.arch!r}")
        if not self.head_dim_ and self.hidden_size % self.num_attention_heads:
            raise ValueError("hidden_size must be divisible by num_attention_heads")

    @property
    def head_dim(self) -> int:
        return self.head_dim_ or self.hidden_size // self.num_attention_heads

    @property
    def arch_id(self) -> int:
        return ARCH_LLAMA if self.arch == "llama" else ARCH_OPT

    def to_dict(self):
        return asdict(self)

    def n_params(self, streamed_only: bool = False) -> int:
        """Parameter count; streamed_only drops the input-embedding gather
        tables (W_stream)."""
        h, L, V = self.hidden_size, self.num_hidden_layers, self.vocab_size
        if self.arch == "llama":
            kv = self.num_key_value_heads * self.head_dim
            qd = self.num_attention_heads * self.head_dim           # == h unless this is a tensor-parallel shard
            per = h * qd * 2 + 2 * h * kv + 3 * h * self.intermediate_size + 2 * h
            n = per * L + h + V * h          # final norm + lm_head
            if not streamed_only:
                n += V * h                   # embed_tokens (untied)
            return n
        pd = self.word_embed_proj_dim
        per = 4 * (h * h + h) + 2 * h * self.ffn_dim + self.ffn_dim + h + 4 * h
        n = per * L + V * pd                 # tied lm_head is streamed once
        if self.do_layer_norm_before:
            n += 2 * h
        if pd != h:
            n += 2 * pd * h
        if not streamed_only:
            n += (self.max_position_embeddings + 2) * h
        return n


def load_config(name_or_path: str) -> ModelConfig:
    path = name_or_path
    if not os.path.exists(path):
        path = os.path.join(_CFG_DIR, name_or_path + ".json")
    with open(path) as f:
        d = json.load(f)
    d.setdefault("name", os.path.splitext(os.path.basename(path))[0])
    return ModelConfig(**d)


def config_from_hf(hf_config) -> ModelConfig:
    """Map a transformers LlamaConfig / OPTConfig onto ModelConfig."""
    mt = getattr(hf_config, "model_type", "")
    if mt == "llama":
        return ModelConfig(
            arch="llama", vocab_size=hf_config.vocab_size, hidden_size=hf_config.hidden_size,
            num_hidden_layers=hf_config.num_hidden_layers,
            num_attention_heads=hf_config.num_attention_heads,
            max_position_embeddings=hf_config.max_position_embeddings,
            intermediate_size=hf_config.intermediate_size,
            num_key_value_heads=getattr(hf_config, "num_key_value_heads", None)
            or hf_config.num_attention_heads,
            rms_norm_eps=hf_config.rms_norm_eps,
            rope_theta=float(getattr(hf_config, "rope_theta", 10000.0) or 10000.0),
            name=getattr(hf_config, "_name_or_path", "") or "llama")
    if mt == "opt":
        return ModelConfig(
            arch="opt", vocab_size=hf_config.vocab_size, hidden_size=hf_config.hidden_size,
            num_hidden_layers=hf_config.num_hidden_layers,
            num_attention_heads=hf_config.num_attention_heads,
            max_position_embeddings=hf_config.max_position_embeddings,
            ffn_dim=hf_config.ffn_dim, do_layer_norm_before=hf_config.do_layer_norm_before,
            word_embed_proj_dim=hf_config.word_embed_proj_dim,
            name=getattr(hf_config, "_name_or_path", "") or "opt")
    raise NotImplementedError(
        f"model_type {mt!r}: only decoder-only llama / opt are on the hot path "
        "(encoder-decoder branches of reference speculative_sampling.py:1946,1958 are out of scope)")
