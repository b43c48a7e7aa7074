"""Drop-in for reference sampling/utils.py:152-245 on device tensors, backed by the HIP kernels.

Same names, argument meaning and error behaviour as the reference; tensors must live on the GPU.
"""
from __future__ import annotations

import ctypes as C

import torch

from .. import _lib
from .._lib import lib, check
from ..engine import _stream
from ..noise import HostTorchNoise


def _rows_f32(t: torch.Tensor) -> torch.Tensor:
    assert t.is_cuda, "the HIP path has no CPU fallback: move the tensor to the GPU"
    return t.float().contiguous()


def _mode(t: torch.Tensor) -> int:
    """dtype_mode for a tensor of the caller's dtype: a bf16 / fp16 row is processed with the reference's rounding points
    (every intermediate tensor of utils.py:182-245 rounded to that dtype); the values travel as exact fp32 copies."""
    return {torch.bfloat16: _lib.SD_NORM_DT_BF16, torch.float16: _lib.SD_NORM_DT_F16}.get(t.dtype, 0)


def norm_logits(logits: torch.Tensor, temperature: float, top_k: float, top_p: float) -> torch.Tensor:
    """reference utils.py:182-210."""
    assert logits.dim() == 2
    x = _rows_f32(logits)
    rows, V = x.shape
    out = torch.empty_like(x)
    err = torch.zeros(rows, dtype=torch.int32, device=x.device)
    ws = torch.empty(lib.sd_norm_workspace_bytes(rows), dtype=torch.uint8, device=x.device)
    check(lib.sd_norm_probs(x.data_ptr(), rows, V, x.stride(0), float(temperature), int(top_k or 0),
                            float(top_p or 0.0), _mode(logits), out.data_ptr(), out.stride(0), err.data_ptr(), ws.data_ptr(),
                            _stream()), "sd_norm_probs")
    if bool(err.any()):
        raise RuntimeError("norm logits error")
    return out.to(logits.dtype)


def top_k_top_p_filter(logits: torch.Tensor, top_k: int = 0, top_p: float = 0.0) -> torch.Tensor:
    """reference utils.py:152-179: dropped tokens become -inf IN the argument, which is returned (utils.py:167, 177);
    top_k == 0 and top_p == 0 leave it untouched.  The kept set comes from the kernel's own top-k / top-p decision,
    not from "probability > 0" (a kept logit whose probability underflows stays kept)."""
    assert logits.dim() == 2
    x = _rows_f32(logits)
    rows, V = x.shape
    out = torch.empty_like(x)
    check(lib.sd_topk_topp_filter(x.data_ptr(), rows, V, x.stride(0), int(top_k or 0), float(top_p or 0.0),
                                  out.data_ptr(), out.stride(0), _stream()), "sd_topk_topp_filter")
    logits.masked_fill_(out == float("-inf"), float("-inf"))
    return logits


def sample(probs: torch.Tensor, num_samples: int = 1, noise=None) -> torch.Tensor:
    """reference utils.py:213-233 (num_samples == 1).  Noise defaults to torch's global CPU generator,
    consumed exactly as ``torch.multinomial`` on CPU consumes it."""
    if num_samples != 1:
        raise NotImplementedError("only num_samples == 1 is on the hot path (reference kvcache_model.py:283)")
    p = _rows_f32(probs)
    assert p.dim() == 2 and p.size(0) == 1
    V = p.size(1)
    tok = torch.zeros(1, dtype=torch.int32, device=p.device)
    err = torch.zeros(1, dtype=torch.int32, device=p.device)
    noise = noise or HostTorchNoise(p.device)
    mode = _mode(probs)
    if getattr(noise, "on_device", False):
        check(lib.sd_sample(p.data_ptr(), V, None, noise.seed, noise.next_draws(1), tok.data_ptr(), err.data_ptr(),
                            mode, _stream()), "sd_sample")
    else:
        # validity is checked before any draw, like multinomial does: peek with a zero-cost dry run
        check(lib.sd_sample(p.data_ptr(), V, p.data_ptr(), 0, 0, tok.data_ptr(), err.data_ptr(), mode, _stream()), "sd_sample")
        if int(err) != 0:
            raise RuntimeError("prob error")
        e = noise.exponential(V, probs.dtype if probs.dtype in (torch.bfloat16, torch.float16) else torch.float32)
        check(lib.sd_sample(p.data_ptr(), V, e.data_ptr(), 0, 0, tok.data_ptr(), err.data_ptr(), mode, _stream()), "sd_sample")
    if int(err) != 0:
        raise RuntimeError("prob error")
    return tok.to(torch.int64).view(1, 1)


def max_fn(x: torch.Tensor) -> torch.Tensor:
    """reference utils.py:236-245."""
    p = _rows_f32(x)
    flat = p.reshape(-1) if p.dim() == 1 else p
    out = torch.empty_like(flat)
    rows = 1 if flat.dim() == 1 else flat.size(0)
    V = flat.size(-1)
    for r in range(rows):
        src = flat if flat.dim() == 1 else flat[r]
        dst = out if out.dim() == 1 else out[r]
        check(lib.sd_max_fn(src.data_ptr(), None, V, dst.data_ptr(), _mode(x), _stream()), "sd_max_fn")
    return out.to(x.dtype).reshape(x.shape)
