"""Oracle for the sampling primitives (test infrastructure, see oracle/__init__.py).

Restates reference sampling/utils.py:152-245 with the randomness made explicit.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from .noise import TorchGlobalNoise

# Tie order inside runs of equal logits after the descending sort of top_k_top_p_filter.  The reference calls
# torch.sort(descending=True) without stable=True (utils.py:170): on CPU that is an unstable std::sort whose order among
# equal keys is unspecified (observed: neither ascending nor descending ids).  It only matters when the top-p cut falls
# inside such a run, which fp32 logits practically never produce and bf16 / fp16 logits often do.  False = exactly the
# reference's call; True = ties in ascending token id, the rule the HIP kernels implement - the fixtures generator uses
# it to mark the low-precision cases whose result depends on the unspecified order.
STABLE_TIES = False


def top_k_top_p_filter(logits: torch.Tensor, top_k: int = 0, top_p: float = 0.0) -> torch.Tensor:
    """reference utils.py:152-179.  Works on a copy (the reference mutates the
    temporary ``logits / temperature`` it is handed, utils.py:197-198)."""
    z = logits.clone()
    if top_k is not None and top_k > 0:
        # k-th largest value; everything strictly below it goes (ties at the k-th value stay)
        kth = torch.topk(z, min(top_k, z.size(-1)))[0][:, -1:]
        z = z.masked_fill(z < kth, float("-inf"))
    if top_p is not None and top_p > 0.0:
        srt, order = torch.sort(z, descending=True, stable=True) if STABLE_TIES else torch.sort(z, descending=True)
        cum = torch.cumsum(F.softmax(srt, dim=-1), dim=-1)
        over = cum > top_p
        # shift right by one: the first token that crosses top_p is kept (utils.py:174-176)
        drop_sorted = torch.zeros_like(over)
        drop_sorted[..., 1:] = over[..., :-1]
        drop = torch.zeros_like(over).scatter(1, order, drop_sorted)
        z = z.masked_fill(drop, float("-inf"))
    return z


def norm_logits(logits: torch.Tensor, temperature: float, top_k: float, top_p: float) -> torch.Tensor:
    """reference utils.py:182-210."""
    assert logits.dim() == 2
    z = top_k_top_p_filter(logits / temperature, top_k=top_k, top_p=top_p)
    probs = torch.log_softmax(z, dim=1).exp()
    if probs.isnan().any() or probs.isinf().any() or (probs < 0).any():
        raise RuntimeError("norm logits error")
    return probs


def sample(probs: torch.Tensor, noise=None) -> torch.Tensor:
    """reference utils.py:213-233 with num_samples == 1.

    ``torch.multinomial(p, 1)`` on CPU: validity checks, then argmax(p / q) per row with
    q ~ Exp(1)^(rows x V) drawn from the generator in ONE call.  Invalid input raises *before*
    any draw, so the noise stream is untouched when the reference's residual fallback fires
    (speculative_sampling.py:2007-2010).  Rows > 1 only occur in multi_speculative_sampling."""
    noise = noise or TorchGlobalNoise()
    assert probs.dim() == 2
    pmax, pmin = probs.max(), probs.min()
    if not bool((pmax < float("inf")) & (pmin >= 0)) or bool((probs.sum(1) == 0).any()):
        raise RuntimeError("prob error")
    q = noise.exponential(probs)
    idx = torch.argmax(probs / q, dim=-1, keepdim=True)
    # utils.py:228-230: a draw that landed on a (near-)zero entry is replaced by the mode - of the FLATTENED
    # tensor (torch.argmax without dim), which for one row is that row's mode
    mask = torch.gather(probs, -1, idx) < 1e-9
    if bool(mask.any()):
        idx[mask] = torch.argmax(probs).item()
    return idx


def max_fn(x: torch.Tensor) -> torch.Tensor:
    """reference utils.py:236-245: max(x, 0) / (sum(max(x, 0)) + 1e-6)."""
    pos = torch.where(x > 0, x, torch.zeros_like(x))
    tot = pos.sum(dim=1, keepdim=True) if x.dim() > 1 else pos.sum()
    return pos / (tot + 1e-6)
