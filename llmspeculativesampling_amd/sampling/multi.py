"""Drop-in for reference sampling/speculative_sampling.py:1379-1716 (``multi_speculative_sampling``), decoder-only,
``strategy="iid"``.

Each iteration ``width`` replicas of the draft sample their own gamma tokens, the target scores all of them, the replica
with the longest accepted run wins and both models continue from it.  Mechanism here: one KV arena, probability arena and
token buffer per replica; the width draft rows of a step and the width*(gamma+1) verify rows go through ONE pass over the
weights each (``sd_batch_forward`` + ``sd_norm_batch``), the replica scan is ``sd_accept_multi``, the residual / bonus
sample ``sd_multi_resample``, and where the reference re-materialises the cache with ``val.repeat(width, ...)``
(kvcache_model.py:180-192) and slices it with ``rollback(end, choice)`` (:390-396, 433-436) only the few positions
written in this iteration are copied from the winner's arenas to the others.

Same signature, return value, ``details`` keys, RNG draw order and EOS rule.  ``strategy="beam"`` (the reference's
default) and ``"acc_beam"`` rest on ``beam_sample_with_kv_cache`` (out of scope).
"""
from __future__ import annotations

import ctypes as C
from time import process_time_ns

import numpy as np
import torch

from .._lib import lib, check, SdMultiItem, SdMultiResult, SdNormRow, SpecDecError
from ..engine import as_specdec_model, batch_forward, _stream, MAX_ROWS_PER_FORWARD, check_token_ids, same_device
from .kvcache_model import KVCacheModel
from .speculative_sampling import _make_noise


def _copy_positions(dst_ses, src_ses, lo: int, hi: int) -> None:
    """KV rows [lo, hi) of every layer / head from one replica's arena to another's."""
    if hi > lo:
        dst_ses.kv[:, :, :, lo:hi].copy_(src_ses.kv[:, :, :, lo:hi])


@torch.no_grad()
def multi_speculative_sampling(prefix: torch.Tensor, approx_model, target_model, eos_token_id, pad_token_id,
                               max_len: int, gamma: int = 4, width: int = 8, num_beams=None, strategy: str = "beam",
                               acc_rate_head=None, acc_rate_thres=0.4, temperature: float = 1, top_k: int = 0,
                               top_p: float = 0, verbose: bool = False, random_seed: int = None,
                               details: bool = False, *, rng=None):
    """reference speculative_sampling.py:1379-1716.  ``rng`` as in ``speculative_sampling``."""
    if strategy in ("beam", "acc_beam", "diverse"):
        raise NotImplementedError(f"strategy {strategy!r} needs beam_sample_with_kv_cache "
                                  "(reference kvcache_model.py:439-567), out of scope; use strategy='iid'")
    if strategy != "iid":
        raise RuntimeError("Strategy not implemented " + strategy)         # reference :1548
    assert prefix.shape[0] == 1, "input batch size must be 1"
    draft_m, target_m = as_specdec_model(approx_model), as_specdec_model(target_model)
    same_device(draft_m, target_m)
    dev = target_m.device
    V = target_m.cfg.vocab_size
    assert draft_m.cfg.vocab_size == V, "draft and target must share a vocabulary"
    W = int(width)
    assert 1 <= W <= 16 and 1 <= gamma <= 16
    assert W * 2 <= MAX_ROWS_PER_FORWARD

    L0 = prefix.shape[1]
    T = L0 + max_len
    cap = T + gamma + 2
    host = [int(t) for t in prefix[0].tolist()]
    ori_eos = sum(1 for t in host if t == eos_token_id)
    noise = _make_noise(rng, dev)
    on_dev = getattr(noise, "on_device", False)

    drafts = [KVCacheModel(draft_m, temperature, top_k, top_p, max_seq=cap, full_history=False) for _ in range(W)]
    targets = [KVCacheModel(target_m, temperature, top_k, top_p, max_seq=cap, full_history=False) for _ in range(W)]
    for m in drafts + targets:
        m._ensure(cap)
    d_ses, t_ses = [m._session for m in drafts], [m._session for m in targets]
    seqs = [torch.zeros(cap + 1, dtype=torch.int32, device=dev) for _ in range(W)]
    for s in seqs:
        s[:L0] = prefix[0].to(device=dev, dtype=torch.int32)
    # the prompt but its last token goes through each model once; the other replicas get copies of those KV rows
    # (the reference runs the same rows width times, kvcache_model.py:156 with a (width, L) batch)
    if L0 > 1:
        d_ses[0].forward(seqs[0][:L0 - 1], 0)
        t_ses[0].forward(seqs[0][:L0 - 1], 0)
        for w in range(1, W):
            _copy_positions(d_ses[w], d_ses[0], 0, L0 - 1)
            _copy_positions(t_ses[w], t_ses[0], 0, L0 - 1)
    draft_len = target_len = L0 - 1

    n_err = 2 * gamma + (gamma + 1)                  # draft norm / draft sample / target norm words per replica
    err = torch.zeros((W, n_err), dtype=torch.int32, device=dev)
    res_dev = torch.zeros(C.sizeof(SdMultiResult), dtype=torch.uint8, device=dev)
    res_host = torch.zeros(C.sizeof(SdMultiResult), dtype=torch.uint8).pin_memory()
    err_host = torch.zeros((W, n_err), dtype=torch.int32).pin_memory()
    norm_ws = torch.empty(lib.sd_norm_workspace_bytes(MAX_ROWS_PER_FORWARD), dtype=torch.uint8, device=dev)
    q_ptr = [m._probs.data_ptr() for m in drafts]
    p_ptr = [m._probs.data_ptr() for m in targets]
    seq_ptr = [s.data_ptr() for s in seqs]
    ld = drafts[0]._probs.stride(0)
    ld_bytes = ld * 4
    err_ptr = err.data_ptr()
    cu = _stream()
    r_const = None
    if random_seed and on_dev:
        g = torch.Generator().manual_seed(int(random_seed))
        r_const = torch.rand(1, generator=g).repeat(W * gamma).to(dev)
    items = (SdMultiItem * W)()
    for w in range(W):
        items[w].p_hist, items[w].q_hist, items[w].seq = p_ptr[w], q_ptr[w], seq_ptr[w]
    Tk, Kk, Pk = float(temperature), int(top_k or 0), float(top_p or 0.0)
    res_mode = target_m.norm_mode if target_m.norm_mode == draft_m.norm_mode else 0

    acc_len, acc_rate = [], []
    approx_time = target_time = other_time = 0
    approx_calls = target_calls = 0
    out = host
    try:
        if len(host) < T:
            check_token_ids(host, V)                                   # nn.Embedding's IndexError, swallowed below
        while len(host) < T:
            L = len(host)
            tt = process_time_ns()
            d_lo, t_lo = draft_len, target_len
            # ---- gamma draft steps, all replicas per pass (kvcache_model.py:279-293 with multi = width)
            for i in range(gamma):
                n_new = L + i - draft_len
                for s in d_ses:
                    s.cache_len = draft_len
                logits = batch_forward(d_ses, seqs, [n_new] * W, [1] * W)
                if on_dev:
                    e_base, seed, draw0 = 0, noise.seed, noise.next_draws(W)
                else:
                    e = noise.exponential_rows(W, V, draft_m.probs_dtype)   # ONE (width, V) draw, like torch.multinomial
                    e_base, seed, draw0 = e.data_ptr(), 0, 0
                rows = (SdNormRow * W)()
                for w in range(W):
                    rows[w].probs_out = q_ptr[w] + (L + i - 1) * ld_bytes
                    rows[w].err = err_ptr + 4 * (w * n_err + i)
                    rows[w].exp_noise = (e_base + w * V * 4) if e_base else None
                    rows[w].philox_seed = seed
                    rows[w].draw_index = draw0 + w
                    rows[w].tok_out = seq_ptr[w] + 4 * (L + i)
                    rows[w].sample_err = err_ptr + 4 * (w * n_err + gamma + i)
                check(lib.sd_norm_batch(logits.data_ptr(), W, V, logits.stride(0), Tk, Kk, Pk, draft_m.norm_mode, rows, 1,
                                        norm_ws.data_ptr(), cu), "sd_norm_batch")
                draft_len = L + i
            approx_calls += 1
            approx_time += process_time_ns() - tt
            tt = process_time_ns()
            # ---- target over every replica's uncached rows (speculative_sampling.py:1560), <= 64 rows per pass
            n_new = L + gamma - target_len
            per_pass = max(1, MAX_ROWS_PER_FORWARD // n_new)
            for a in range(0, W, per_pass):
                grp = list(range(a, min(W, a + per_pass)))
                for w in grp:
                    t_ses[w].cache_len = target_len
                logits = batch_forward([t_ses[w] for w in grp], [seqs[w] for w in grp], [n_new] * len(grp),
                                       [n_new] * len(grp))
                rows = (SdNormRow * (len(grp) * n_new))()
                k = 0
                for w in grp:
                    for r_ in range(n_new):
                        pos = L + gamma - n_new + r_
                        rows[k].probs_out = p_ptr[w] + pos * ld_bytes
                        rows[k].err = err_ptr + 4 * (w * n_err + 2 * gamma + min(r_, gamma))
                        k += 1
                check(lib.sd_norm_batch(logits.data_ptr(), k, V, logits.stride(0), Tk, Kk, Pk, target_m.norm_mode, rows, 0,
                                        norm_ws.data_ptr(), cu), "sd_norm_batch")
            # the target's own sample, drawn and thrown away by the reference (kvcache_model.py:283)
            if on_dev:
                noise.next_draws(W)
            else:
                noise.skip_exponential_rows(W, V, target_m.probs_dtype)
            target_calls += 1
            target_time += process_time_ns() - tt
            tt = process_time_ns()
            # ---- replica scan
            token = None
            if on_dev:
                if random_seed:
                    noise.reseed(random_seed)
                r_ptr, seed, draw = (r_const.data_ptr() if r_const is not None else None), noise.seed, \
                    noise.next_draws(W * gamma)
            else:
                r, token = noise.uniforms(W * gamma, random_seed)
                r_ptr, seed, draw = r.data_ptr(), 0, 0
            check(lib.sd_accept_multi(items, W, ld, L, gamma, r_ptr, seed, draw, res_dev.data_ptr(), cu),
                  "sd_accept_multi")
            res_host.copy_(res_dev, non_blocking=True)
            err_host.copy_(err, non_blocking=True)
            torch.cuda.current_stream().synchronize()
            eh = err_host.numpy()
            if eh.any():
                if eh[:, gamma:2 * gamma].any():
                    raise RuntimeError("prob error")                   # reference utils.py:224
                raise RuntimeError("norm logits error")                # reference utils.py:207
            res = SdMultiResult.from_buffer_copy(res_host.numpy().tobytes())
            if not on_dev:
                noise.realign(token, res.n_uniform)
            choice, l, n = res.choice, res.chosen.n_accepted, res.chosen.n
            all_accept = bool(res.chosen.flags & 4)
            pa = np.ctypeslib.as_array(res.p_at).reshape(16, 16)[:W, :gamma].astype(np.float32)
            qa = np.ctypeslib.as_array(res.q_at).reshape(16, 16)[:W, :gamma].astype(np.float32)
            with np.errstate(divide="ignore", invalid="ignore"):
                ratio = pa / qa                                        # fp32 division, as p[...] / q[...] (:1597)
            for w in range(W):
                for i in range(gamma):
                    a = float(ratio[w, i])
                    if a > 1:
                        a = 1
                    if qa[w, i] == 0:
                        a = 0
                    acc_rate.append(a)
            acc_len.append(l)
            # ---- residual / bonus sample on the winner's rows (:1645-1679)
            if on_dev:
                e_ptr, seed, draw = None, noise.seed, noise.next_draws(1)
            else:
                e1 = noise.exponential(V, target_m.probs_dtype if res_mode else torch.float32)
                e_ptr, seed, draw = e1.data_ptr(), 0, 0
            check(lib.sd_multi_resample(p_ptr[choice], q_ptr[choice], ld, V, seq_ptr[choice], gamma, e_ptr, seed, draw,
                                        res_dev.data_ptr(), res_mode, cu), "sd_multi_resample")
            res_host.copy_(res_dev, non_blocking=True)
            torch.cuda.current_stream().synchronize()
            res = SdMultiResult.from_buffer_copy(res_host.numpy().tobytes())
            if res.chosen.flags & 2:
                out = host + [int(res.chosen.drafted[i]) for i in range(l)]    # output_prefix was already cut (:1641)
                raise RuntimeError("prob error")
            t = res.chosen.next_token
            host = host + [int(res.chosen.drafted[i]) for i in range(l)] + [int(t)]
            # ---- everyone continues from the winner: rollback(n+1 | n+2, choice) + the next forward's repeat()
            new_draft = min(L + gamma - 1, n + 1)
            new_target = L + gamma if all_accept else n + 1
            for w in range(W):
                if w != choice:
                    seqs[w][L:n + 2].copy_(seqs[choice][L:n + 2])
                    _copy_positions(d_ses[w], d_ses[choice], d_lo, new_draft)
                    _copy_positions(t_ses[w], t_ses[choice], t_lo, new_target)
            draft_len, target_len = new_draft, new_target
            other_time += process_time_ns() - tt
            out = host
            if sum(1 for x in host if x == eos_token_id) > ori_eos:    # EOS rule (:1688-1695)
                seen, cut = 0, len(host)
                for idx, x in enumerate(host):
                    if x == eos_token_id:
                        seen += 1
                        if seen == ori_eos + 1:
                            cut = idx + 1
                            break
                out = host[:cut]
                break
    except SpecDecError:                                               # an engine failure is never swallowed
        raise
    except (RuntimeError, IndexError) as e:                            # 'norm logits error' / 'prob error' / a bad token id:
        print(e)                                                       # printed and swallowed like the reference (:1689-1690)

    result = torch.tensor([out], dtype=torch.int64, device=prefix.device)
    if verbose:
        print("approx model time", approx_time / 1e9)
        print("target model time", target_time / 1e9)
        print("other time", other_time / 1e9)
        print("acc len", np.mean(acc_len) if acc_len else 0.0, len(acc_len), acc_len)
    if details:
        return result, {"approx_time": approx_time, "target_time": target_time, "other_time": other_time,
                        "acc_len": acc_len, "acc_rate": np.mean(acc_rate) if acc_rate else 0.0,
                        "target_call_times": target_calls, "approx_call_times": approx_calls}
    return result
