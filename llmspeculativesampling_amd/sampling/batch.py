"""Stream-batched speculative sampling ("throughput mode"): B independent prompt streams
decode in lockstep on one GPU and share every pass over the weights.  Each draft step is one forward over the B (or up
to 2B) new rows, each verify is one target forward over B*(gamma+1) rows: the bytes streamed per iteration are those of
ONE stream, the tokens produced are B times as many.

Every stream keeps its own KV arenas, probability arenas, token buffer and Philox stream, and runs exactly the
per-stream algorithm of reference sampling/speculative_sampling.py:1876-2076 (batch size 1 there, :1905): with the
same seeds the outputs equal those of B separate ``speculative_sampling(..., rng=DeviceNoise(seed))`` calls.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence

import numpy as np
import torch

from .._lib import lib, check, SdAcceptItem, SdAcceptResult, SdNormRow
from ..engine import as_specdec_model, batch_forward, _stream, MAX_ROWS_PER_FORWARD, check_token_ids, same_device
from ..noise import DeviceNoise
from .kvcache_model import KVCacheModel


class _Stream:
    __slots__ = ("idx", "draft", "target", "seq32", "host", "noise", "ori_eos", "T", "done", "out", "acc_len",
                 "acc_rate", "calls", "draft_len", "target_len", "err", "prompt_len", "q_ptr", "p_ptr", "seq_ptr",
                 "err_ptr")


@torch.no_grad()
def speculative_sampling_batch(prefixes: Sequence[torch.Tensor], approx_model, target_model, eos_token_id,
                               pad_token_id, max_len: int, gamma: int = 4, temperature: float = 1, top_k: int = 0,
                               top_p: float = 0, random_seed: int = None, details: bool = False,
                               seeds: Optional[Sequence[int]] = None, _timing: Optional[dict] = None):
    """B streams at once; ``prefixes[i]`` is (1, L_i) int64.  Returns a list of (1, len_i) tensors (and a list of
    ``details`` dicts with the reference's keys when ``details``).  Device Philox RNG, stream i seeded ``seeds[i]``."""
    draft_m, target_m = as_specdec_model(approx_model), as_specdec_model(target_model)
    same_device(draft_m, target_m)
    dev = target_m.device
    V = target_m.cfg.vocab_size
    assert draft_m.cfg.vocab_size == V
    for pf in prefixes:
        check_token_ids(pf, V)
    B = len(prefixes)
    assert 1 <= B <= 16 and 1 <= gamma <= 16
    seeds = list(seeds) if seeds is not None else [int(torch.initial_seed()) + i for i in range(B)]
    res_sz = C.sizeof(SdAcceptResult)
    res_dev = torch.zeros((B, res_sz), dtype=torch.uint8, device=dev)
    res_host = torch.zeros((B, res_sz), dtype=torch.uint8).pin_memory()
    n_err = 3 * gamma + 1
    r_const = None
    if random_seed:
        g = torch.Generator().manual_seed(int(random_seed))
        r_const = torch.rand(1, generator=g).repeat(gamma).to(dev)

    streams: List[_Stream] = []
    for i, pf in enumerate(prefixes):
        assert pf.shape[0] == 1, "input batch size must be 1"
        st = _Stream()
        st.idx = i
        L = pf.shape[1]
        st.prompt_len = L
        st.T = L + max_len
        cap = st.T + gamma + 2
        st.draft = KVCacheModel(draft_m, temperature, top_k, top_p, max_seq=cap, full_history=False)
        st.target = KVCacheModel(target_m, temperature, top_k, top_p, max_seq=cap, full_history=False)
        st.draft._ensure(cap)
        st.target._ensure(cap)
        st.seq32 = torch.zeros(cap + 1, dtype=torch.int32, device=dev)
        st.seq32[:L] = pf[0].to(device=dev, dtype=torch.int32)
        st.host = [int(t) for t in pf[0].tolist()]
        st.ori_eos = sum(1 for t in st.host if t == eos_token_id)
        st.noise = DeviceNoise(seeds[i])
        st.done = False
        st.out = st.host
        st.acc_len, st.acc_rate, st.calls = [], [], 0
        st.err = torch.zeros(n_err, dtype=torch.int32, device=dev)
        # prefill everything but the last prompt token stream by stream (rows beyond one pass's budget anyway);
        # the decode loop then starts with 1 new draft row and gamma+1 new target rows like every later iteration
        if L > 1:
            st.draft._session.forward(st.seq32[:L - 1], 0)
            st.target._session.forward(st.seq32[:L - 1], 0)
        st.draft_len = st.target_len = L - 1
        st.q_ptr, st.p_ptr = st.draft._probs.data_ptr(), st.target._probs.data_ptr()
        st.seq_ptr, st.err_ptr = st.seq32.data_ptr(), st.err.data_ptr()
        streams.append(st)

    d_sessions = lambda ss: [s.draft._session for s in ss]
    t_sessions = lambda ss: [s.target._session for s in ss]
    norm_ws = torch.empty(lib.sd_norm_workspace_bytes(MAX_ROWS_PER_FORWARD), dtype=torch.uint8, device=dev)
    cu = _stream()
    res_base = res_dev.data_ptr()
    ld_bytes = streams[0].draft._probs.stride(0) * 4
    max_verify = max(1, MAX_ROWS_PER_FORWARD // (gamma + 1))     # streams per target pass

    while True:
        act = [s for s in streams if not s.done and len(s.host) < s.T]
        for s in streams:
            if not s.done and len(s.host) >= s.T:
                s.done = True
        if not act:
            break
        n = len(act)
        Ls = [len(s.host) for s in act]
        base_draw = [s.noise.next_draws(gamma) for s in act]
        # ---- draft: gamma steps over all active streams
        for i in range(gamma):
            n_new = [L + i - s.draft_len for s, L in zip(act, Ls)]
            for s in act:
                s.draft._session.cache_len = s.draft_len
            logits = batch_forward(d_sessions(act), [s.seq32 for s in act], n_new, [1] * n)
            rows = (SdNormRow * n)()
            for j, (s, L) in enumerate(zip(act, Ls)):
                # raw pointer arithmetic (one tensor view per row costs more host time than the kernels it feeds)
                rows[j].probs_out = s.q_ptr + (L + i - 1) * ld_bytes
                rows[j].err = s.err_ptr + 4 * i
                rows[j].exp_noise = None
                rows[j].philox_seed = s.noise.seed
                rows[j].draw_index = base_draw[j] + i
                rows[j].tok_out = s.seq_ptr + 4 * (L + i)
                rows[j].sample_err = s.err_ptr + 4 * (gamma + i)
                s.draft_len = L + i
            check(lib.sd_norm_batch(logits.data_ptr(), n, V, logits.stride(0), float(temperature), int(top_k or 0),
                                    float(top_p or 0.0), draft_m.norm_mode, rows, 1, norm_ws.data_ptr(), cu), "sd_norm_batch")
        # ---- verify: the uncached rows of every stream, max_verify streams per pass over the target weights
        if _timing is not None:
            ev0 = torch.cuda.Event(enable_timing=True)
            ev0.record()
        for a in range(0, n, max_verify):
            grp = act[a:a + max_verify]
            gL = Ls[a:a + max_verify]
            n_new = [L + gamma - s.target_len for s, L in zip(grp, gL)]
            for s in grp:
                s.target._session.cache_len = s.target_len
            logits = batch_forward(t_sessions(grp), [s.seq32 for s in grp], n_new, n_new)
            rows = (SdNormRow * sum(n_new))()
            k = 0
            for s, L, nn in zip(grp, gL, n_new):
                for r in range(nn):
                    pos = L + gamma - nn + r
                    rows[k].probs_out = s.p_ptr + pos * ld_bytes
                    rows[k].err = s.err_ptr + 4 * (2 * gamma + min(r, gamma))
                    k += 1
            check(lib.sd_norm_batch(logits.data_ptr(), k, V, logits.stride(0), float(temperature), int(top_k or 0),
                                    float(top_p or 0.0), target_m.norm_mode, rows, 0, norm_ws.data_ptr(), cu), "sd_norm_batch")
        if _timing is not None:
            ev1 = torch.cuda.Event(enable_timing=True)
            ev1.record()
            _timing.setdefault("verify", []).append((ev0, ev1, n, sum(Ls) / n + gamma))
        # ---- accept scan + residual / bonus sample, all streams in two launches
        items = (SdAcceptItem * n)()
        for j, (s, L) in enumerate(zip(act, Ls)):
            s.noise.next_draws(1)                                  # the discarded target sample
            seed_before = s.noise.seed
            if random_seed:
                s.noise.reseed(random_seed)
                d_scan = 0
            else:
                d_scan = s.noise.next_draws(gamma)
            it = items[j]
            it.p_hist = s.p_ptr
            it.q_hist = s.q_ptr
            it.seq = s.seq_ptr
            it.L = L
            it.r = r_const.data_ptr() if r_const is not None else None
            it.exp_noise = None
            it.philox_seed = s.noise.seed
            it.draw_scan = d_scan
            it.draw_resample = s.noise.next_draws(1)
            it.res = res_base + s.idx * res_sz
            it.err_flags = s.err_ptr
            it.n_err = n_err
        res_mode = target_m.norm_mode if target_m.norm_mode == draft_m.norm_mode else 0
        check(lib.sd_accept_batch(items, n, act[0].draft._probs.stride(0), V, gamma, res_mode, cu), "sd_accept_batch")
        res_host.copy_(res_dev, non_blocking=True)
        torch.cuda.current_stream().synchronize()
        raw = res_host.numpy()
        for s, L in zip(act, Ls):
            res = SdAcceptResult.from_buffer_copy(raw[s.idx].tobytes())
            if res.flags & 2:
                raise RuntimeError("s")
            if res.flags & 8:
                raise RuntimeError("s")
            l, nn, t = res.n_accepted, res.n, res.next_token
            for i in range(gamma):
                s.acc_rate.append(min(1.0, float(res.p_at[i]) / float(res.q_at[i])))
            s.acc_len.append(l)
            s.calls += 1
            s.host = s.host + [int(res.drafted[i]) for i in range(l)] + [t]
            s.draft_len = min(L + gamma - 1, nn + 1)
            s.target_len = nn + 1
            s.out = s.host
            eos_total = sum(1 for x in s.host if x == eos_token_id)
            if eos_total > s.ori_eos:
                seen, cut = 0, len(s.host)
                for idx, x in enumerate(s.host):
                    if x == eos_token_id:
                        seen += 1
                        if seen == s.ori_eos + 1:
                            cut = idx + 1
                            break
                s.out = s.host[:cut]
                s.done = True

    outs = [torch.tensor([s.out], dtype=torch.int64, device=prefixes[i].device) for i, s in enumerate(streams)]
    if details:
        ds = [{"approx_time": 0, "target_time": 0, "other_time": 0, "acc_len": s.acc_len,
               "acc_rate": float(np.mean(s.acc_rate)) if s.acc_rate else 0.0, "target_call_times": s.calls,
               "approx_call_times": s.calls, "target_model_time": 0, "target_pre_cache_time": 0,
               "target_post_prob_time": 0} for s in streams]
        return outs, ds
    return outs
