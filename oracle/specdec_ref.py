"""Oracle decoding loops (test infrastructure, see oracle/__init__.py).

speculative_sampling restates reference sampling/speculative_sampling.py:1876-2076
(decoder-only branches); autoregressive_sampling restates
sampling/autoregressive_sampling.py:8-61.  The draw order is the RNG contract
(SURVEY.md section 8(a) A1): gamma draft samples, one target sample whose result
is thrown away, up to gamma uniforms (each preceded by a reseed when random_seed
is truthy), then one residual-or-bonus sample.
"""
from __future__ import annotations

from time import process_time_ns

import numpy as np
import torch

from .kvcache_ref import RefKVCacheModel
from .noise import TorchGlobalNoise
from .sampling_ref import max_fn, norm_logits, sample


@torch.no_grad()
def speculative_sampling(prefix, approx_model, target_model, eos_token_id, pad_token_id, max_len,
                         gamma=4, temperature=1, top_k=0, top_p=0, verbose=False, random_seed=None,
                         details=False, noise=None):
    noise = noise or TorchGlobalNoise()
    eos_in_prompt = int((prefix == eos_token_id).sum())
    T = prefix.shape[1] + max_len
    assert prefix.shape[0] == 1, "input batch size must be 1"

    draft = RefKVCacheModel(approx_model, temperature, top_k, top_p, noise)
    target = RefKVCacheModel(target_model, temperature, top_k, top_p, noise)
    t_draft = t_target = t_other = 0
    n_draft_calls = n_target_calls = 0
    acc_rate, acc_len = [], []
    out = prefix
    try:
        while prefix.shape[1] < T:
            tick = process_time_ns()
            x = draft.generate(prefix, gamma)
            L = prefix.shape[1]
            n_draft_calls += 1
            t_draft += process_time_ns() - tick
            tick = process_time_ns()
            target.generate(x, 1)          # the sample inside is drawn and discarded (kvcache_model.py:283)
            n_target_calls += 1
            t_target += process_time_ns() - tick
            tick = process_time_ns()

            p_hist, q_hist = target._prob_history, draft._prob_history
            for i in range(gamma):         # stats over all gamma drafted positions (:1966-1971)
                j = int(x[0, L + i])
                acc_rate.append(min(1.0, p_hist[0, L + i - 1, j].item() / q_hist[0, L + i - 1, j].item()))
            n = L + gamma - 1
            accepted = 0
            for i in range(gamma):         # (:1975-1990)
                if random_seed:
                    noise.reseed(random_seed)
                r = noise.uniform()
                j = int(x[0, L + i])
                ratio = p_hist[0, L + i - 1, j].item() / q_hist[0, L + i - 1, j].item()   # python double
                if bool(r > ratio):        # float32 compare: the double is rounded to fp32 by torch
                    n = L + i - 1
                    break
                accepted += 1
            acc_len.append(accepted)
            assert n >= L - 1
            prefix = x[:, :n + 1]
            draft.rollback(n + 1)
            assert draft._prob_history.shape[-2] <= n + 1
            if n < L + gamma - 1:
                try:                       # residual resample (:2007-2010)
                    t = sample(max_fn(p_hist[:, n, :] - q_hist[:, n, :]), noise)
                except Exception:
                    t = sample(max_fn(p_hist[:, n, :]), noise)
                target.rollback(n + 1)
            else:
                assert n == p_hist.shape[1] - 1
                t = sample(p_hist[:, -1, :], noise)
                target.rollback(n + 2)
            prefix = torch.cat((prefix, t), dim=1)
            out = prefix
            hit = out == eos_token_id      # EOS rule over the whole sequence (:2033-2041)
            if int(hit.sum()) > eos_in_prompt:
                keep = torch.cumsum(hit.float(), dim=1) < eos_in_prompt + 1
                end = int(keep.sum())
                if end < keep.size(1):
                    keep[:, end] = True
                out = out[keep][None, :]
                break
            t_other += process_time_ns() - tick
    except Exception as e:                 # (:2044-2046)
        print(e)
        raise RuntimeError("s")
    if details:
        return out, {
            "approx_time": t_draft, "target_time": t_target, "other_time": t_other,
            "acc_len": acc_len, "acc_rate": np.mean(acc_rate),
            "target_call_times": n_target_calls, "approx_call_times": n_draft_calls,
            "target_model_time": target.forward_time_dict["_model_time"],
            "target_pre_cache_time": target.forward_time_dict["prepare_cache_time"],
            "target_post_prob_time": target.forward_time_dict["norm_prob_time"],
            # extras for the tests (not in the reference dict)
            "_rows_fed_draft": draft.rows_fed, "_rows_fed_target": target.rows_fed,
        }
    return out


@torch.no_grad()
def autoregressive_sampling(x, model, N, eos_token_id, temperature=1, top_k=0, top_p=0,
                            pad_token_id=None, noise=None):
    noise = noise or TorchGlobalNoise()
    n, T = len(x), len(x) + N              # len() is the batch dim == 1 (:13-14): exactly N tokens unless EOS
    past = None
    while n < T:
        if past:
            out = model(x[:, -1:], past_key_values=past, use_cache=True)
        else:
            out = model(x)
        p = norm_logits(out.logits[:, -1, :], temperature, top_k, top_p)
        past = out.past_key_values
        nxt = sample(p, noise)
        x = torch.cat((x, nxt), dim=1)
        n += 1
        if int(nxt) == eos_token_id:
            break
    return x
