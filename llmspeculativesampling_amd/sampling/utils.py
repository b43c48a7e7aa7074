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
    # a 16-bit tensor is sorted / softmaxed / cumsummed in its own dtype by the reference (utils.py:170-172): same mode
    # switch as norm_logits, so the kept set at the top-p cut is decided on the same 16-bit sums
    check(lib.sd_topk_topp_filter(x.data_ptr(), rows, V, x.stride(0), int(top_k or 0), float(top_p or 0.0), _mode(logits),
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


# ------------------------------------------------------------------------------------------------------------------
# Host-side helpers of the tree-attention variant.  Pure index / scalar arithmetic on a few
# dozen numbers per iteration: they stay on the host, like in the reference.
# ------------------------------------------------------------------------------------------------------------------
def get_seq_att_mask(input_cnt, all_input_idx, all_beam_idx, all_next_token, input_len, pad_token_id, device="cpu"):
    """reference utils.py:95-148.  Flattens the beam-sampled draft tree into per-input token rows: beam j of level l
    appends all_next_token[l][j] to input all_input_idx[l][j] and attends to what its parent beam
    (all_beam_idx[l][j], a beam of level l-1) attended to, plus itself.  Returns
    (tokens (input_cnt, n), mask (input_cnt, n, input_len + n) with the prefix part all True, pos (input_cnt + nodes, 2)
    = [input, slot] with leading [i, -1] rows, position_ids (input_cnt, n))."""
    toks = [[] for _ in range(input_cnt)]
    rows = [[] for _ in range(input_cnt)]
    depth_of = [[] for _ in range(input_cnt)]
    parent_masks = [[] for _ in range(all_input_idx[0].numel())]
    pos = [[i, -1] for i in range(input_cnt)]
    for level, (inps, beams, nxt) in enumerate(zip(all_input_idx, all_beam_idx, all_next_token)):
        this_level = []
        for j in range(inps.numel()):
            i, b = int(inps[j]), int(beams[j])
            slot = len(toks[i])
            pos.append([i, slot])
            toks[i].append(int(nxt[j]))
            depth_of[i].append(input_len + level)
            inherited = parent_masks[b]
            m = inherited + [False] * (slot - len(inherited)) + [True]
            rows[i].append(m)
            this_level.append(m)
        parent_masks = this_level
    n = max(len(t) for t in toks)
    mask = torch.zeros((input_cnt, n, input_len + n), dtype=torch.bool)
    mask[:, :, :input_len] = True
    seq = torch.full((input_cnt, n), int(pad_token_id), dtype=torch.long)
    pids = torch.zeros((input_cnt, n), dtype=torch.long)
    for i in range(input_cnt):
        k = len(toks[i])
        seq[i, :k] = torch.tensor(toks[i], dtype=torch.long)
        pids[i, :k] = torch.tensor(depth_of[i], dtype=torch.long)
        for r, m in enumerate(rows[i]):
            mask[i, r, input_len:input_len + len(m)] = torch.tensor(m, dtype=torch.bool)
    return seq.to(device), mask.to(device), torch.tensor(pos, dtype=torch.long, device=device), pids.to(device)


def get_accept_prob(p, q):
    """reference utils.py:247-250: sum_i min(1, p_i / (q_i + 1e-6)) * q_i."""
    r = p / (q + 1e-6)
    return torch.sum(torch.clamp(r, max=1.0) * q)


def update_large_prob(p, q):
    """reference utils.py:252-255: norm(max(p - q, 0)) with the +1e-6 denominator."""
    d = torch.clamp(p - q, min=0.0)
    return d / (d.sum() + 1e-6)


def get_num_acc_prob(p, q, m):
    """reference utils.py:316-337 (with :257-314): distribution of how many of m drafts drawn from q get accepted
    against p when every rejection replaces p by its residual.  alpha_i is the acceptance probability after i
    rejections; the count recursion is P(n, k) = sum_i alpha_i prod_{j<i}(1 - alpha_j) P(n - i, k - 1), restarting from
    alpha_0 as the reference's does.  Returns (prob, expect); index quirk kept: prob[k - 1] = P(m, k), prob[m] = P(m, 0)."""
    alpha = []
    cur = p.clone()
    for _ in range(m):
        alpha.append(get_accept_prob(cur, q))
        cur = update_large_prob(cur, q)
    none = [1.0]                                     # none[i] = prod_{j<i} (1 - alpha_j)
    for a in alpha:
        none.append(none[-1] * (1 - a))
    table = {}

    def count(n, k):
        if n < k:
            return 0
        if n == 0 and k == 0:
            return 1
        if k == 0:
            return none[n]
        if (n, k) not in table:
            table[(n, k)] = sum(none[i - 1] * alpha[i - 1] * count(n - i, k - 1) for i in range(1, n + 1))
        return table[(n, k)]
    prob = torch.zeros(m + 1, dtype=torch.float32)
    expect = 0.0
    for k in range(m + 1):
        v = count(m, k)
        prob[k - 1] = v
        expect = expect + v * k
    return prob, expect


def get_expect_cnt_by_thres(p_width, expect_thres):
    """reference utils.py:339-350."""
    n, mass = p_width.numel(), 0
    while mass < expect_thres and n > 0:
        n -= 1
        mass += p_width[n]
    return int(n)
