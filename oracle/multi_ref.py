"""Oracle for the width-w i.i.d.-draft variant (test infrastructure, see oracle/__init__.py).

Restates reference sampling/speculative_sampling.py:1379-1716 (``multi_speculative_sampling``),
decoder-only, ``strategy="iid"``: every iteration the draft
cache is replicated ``width`` times and each replica samples its own gamma tokens
(kvcache_model.py:273-276), the target scores all of them in one batched forward, the
replica with the longest accepted run wins and both caches are cut back to that replica
(``rollback(end, choice)``, kvcache_model.py:390-396, 433-436).

Draw order per iteration (the RNG contract of this variant): gamma draft samples, each ONE
Exp(1) draw of shape (width, V); one discarded target sample of shape (width, V) (:1560 via
kvcache_model.py:283); for w = 0.. the uniforms of replica w up to and including its first
reject (each preceded by a reseed when ``random_seed`` is truthy), stopping after the first
replica that accepts everything; then one (1, V) residual-or-bonus sample.

The ``beam`` / ``acc_beam`` strategies need ``beam_sample_with_kv_cache`` (SURVEY.md
section 2 row 10, out of scope) and raise NotImplementedError.
"""
from __future__ import annotations

import numpy as np
import torch

from .kvcache_ref import RefKVCacheModel
from .noise import TorchGlobalNoise
from .sampling_ref import max_fn, sample


@torch.no_grad()
def multi_speculative_sampling(prefix, approx_model, target_model, eos_token_id, pad_token_id, max_len,
                               gamma=4, width=8, num_beams=None, strategy="beam", acc_rate_head=None,
                               acc_rate_thres=0.4, temperature=1, top_k=0, top_p=0, verbose=False,
                               random_seed=None, details=False, noise=None):
    noise = noise or TorchGlobalNoise()
    if strategy in ("beam", "acc_beam", "diverse"):
        raise NotImplementedError(f"strategy {strategy!r} needs beam_sample_with_kv_cache (out of scope)")
    if strategy != "iid":
        raise RuntimeError("Strategy not implemented " + strategy)
    eos_in_prompt = int((prefix == eos_token_id).sum())
    T = prefix.shape[1] + max_len
    acc_len, acc_rate = [], []
    draft = RefKVCacheModel(approx_model, temperature, top_k, top_p, noise)
    target = RefKVCacheModel(target_model, temperature, top_k, top_p, noise)
    assert prefix.shape[0] == 1, "input batch size must be 1"
    n_target_calls = n_draft_calls = 0
    out = prefix
    try:
        while out.shape[1] < T:
            L = out.shape[1]
            x = draft.generate(out, gamma, multi=width, strategy="iid")          # (width, L + gamma)   (:1531-1535)
            q = draft._prob_history[:, L - 1:, :]                                # (:1543)
            inc = x.shape[1] - L
            n_draft_calls += 1
            target.generate(x, 1)                                               # sample discarded (:1560)
            n_target_calls += 1
            p = target._prob_history
            for w in range(width):                                              # statistics (:1592-1601)
                for i in range(gamma):
                    j = x[w, L + i]
                    a = (p[w, L + i - 1, j] / q[w, i, j]).item()
                    if a > 1:
                        a = 1
                    if q[w, i, j] == 0:
                        a = 0
                    acc_rate.append(a)
            all_accept = False
            max_n, max_l, choice = L - 1, 0, 0
            for w in range(width):                                              # (:1611-1638)
                cur_n, cur_l, cur_all = L - 1, 0, True
                for i in range(inc):
                    if random_seed:
                        noise.reseed(random_seed)
                    r = noise.uniform()
                    j = x[w, L + i]
                    if r < torch.min(torch.tensor([1]), p[w, L + i - 1, j] / q[w, i, j]):
                        cur_l += 1
                        cur_n += 1
                    else:
                        cur_all = False
                        break
                if cur_l > max_l:
                    max_n, max_l, choice = cur_n, cur_l, w
                    if cur_all:
                        all_accept = True
                        break
            acc_len.append(max_l)
            n = max_n
            out = x[choice:choice + 1, :n + 1]
            draft.rollback(n + 1, choice)
            if all_accept:
                t = sample(p[choice:choice + 1, -1, :], noise)                  # (:1646)
                target.rollback(n + 2, choice)
            else:
                new_p = max_fn(p[choice:choice + 1, n, :] - q[choice:choice + 1, max_l, :])
                try:
                    t = sample(new_p, noise)
                except Exception:
                    t = sample(p[choice:choice + 1, n, :], noise)               # no max_fn here (:1666-1668)
                target.rollback(n + 1, choice)
            out = torch.cat((out, t), dim=1)
            mask = (out == eos_token_id)                                        # EOS rule (:1688-1695)
            if int(mask.int().sum()) > eos_in_prompt:
                keep = torch.cumsum(mask.float(), dim=1) < eos_in_prompt + 1
                end = int(keep.int().sum())
                if end < keep.size(1):
                    keep[:, end] = True
                out = out[keep][None, :]
                break
    except Exception as e:                                                      # swallowed, like the reference (:1696-1697)
        print(e)
    if details:
        return out, {"approx_time": 0, "target_time": 0, "other_time": 0, "acc_len": acc_len,
                     "acc_rate": np.mean(acc_rate), "target_call_times": n_target_calls,
                     "approx_call_times": n_draft_calls,
                     "_rows_fed_draft": draft.rows_fed, "_rows_fed_target": target.rows_fed}
    return out
