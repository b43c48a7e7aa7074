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

from .._lib import lib, check, SdAcceptResult, SdBatchStream
from ..engine import as_specdec_model, _stream, MAX_ROWS_PER_FORWARD, check_token_ids, same_device, batch_prefill
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
        st.draft_len = st.target_len = L - 1
        st.q_ptr, st.p_ptr = st.draft._probs.data_ptr(), st.target._probs.data_ptr()
        st.seq_ptr, st.err_ptr = st.seq32.data_ptr(), st.err.data_ptr()
        streams.append(st)

    # prefill everything but the last prompt token, the B prompts packed into passes of up to 256 rows (engine.batch_prefill:
    # one pass over the weights serves several streams); the decode loop then starts with 1 new draft row and gamma+1 new
    # target rows per stream like every later iteration
    for side in ("draft", "target"):
        batch_prefill([getattr(s, side)._session for s in streams], [s.seq32 for s in streams],
                      [s.prompt_len - 1 for s in streams])

    # the lock-step loop itself runs inside libspecdec (sd_spec_batch_generate): per iteration gamma batched draft steps,
    # the verify passes, the batched accept + residual sample, one copy of the result blocks and one wait - the
    # interpreter sees the finished token buffers and the per-iteration statistics
    norm_ws = torch.empty(lib.sd_norm_workspace_bytes(MAX_ROWS_PER_FORWARD), dtype=torch.uint8, device=dev)
    cu = _stream()
    arr = (SdBatchStream * B)()
    keep = []                                                     # host arrays the native loop writes into
    max_iters = max(1, int(max_len)) + 1
    for s in streams:
        cap = s.T + gamma + 2
        hs = np.zeros(cap, dtype=np.int32)
        hs[:len(s.host)] = s.host
        acc = np.zeros(max_iters, dtype=np.int32)
        pa = np.zeros(max_iters * gamma, dtype=np.float32)
        qa = np.ones(max_iters * gamma, dtype=np.float32)
        keep.append((hs, acc, pa, qa))
        it = arr[s.idx]
        it.draft, it.target = s.draft._session.handle, s.target._session.handle
        it.seq, it.q_hist, it.p_hist, it.err_words = s.seq_ptr, s.q_ptr, s.p_ptr, s.err_ptr
        it.res_dev = res_dev.data_ptr() + s.idx * res_sz
        it.res_host = res_host.data_ptr() + s.idx * res_sz
        it.host_seq = hs.ctypes.data
        it.len, it.T, it.ori_eos_cnt = len(s.host), s.T, s.ori_eos
        it.draft_len, it.target_len = s.draft_len, s.target_len
        it.seed, it.draw = s.noise.seed, s.noise.draw
        it.acc_len_out, it.p_at_out, it.q_at_out = acc.ctypes.data, pa.ctypes.data, qa.ctypes.data
    n_log = max_iters * 2
    v_ms = np.zeros(n_log, dtype=np.float32)
    v_n = np.zeros(n_log, dtype=np.int32)
    v_ctx = np.zeros(n_log, dtype=np.float32)
    c_iters, c_err = C.c_int(0), C.c_int(0)
    d0, t0 = streams[0].draft._session, streams[0].target._session
    check(lib.sd_spec_batch_generate(arr, B, gamma, float(temperature), int(top_k or 0), float(top_p or 0.0), V,
                                     streams[0].draft._probs.stride(0), int(eos_token_id), int(random_seed or 0),
                                     r_const.data_ptr() if r_const is not None else None, draft_m.norm_mode,
                                     target_m.norm_mode, d0.logits.data_ptr(), d0.logits.stride(0), t0.logits.data_ptr(),
                                     t0.logits.stride(0), norm_ws.data_ptr(), MAX_ROWS_PER_FORWARD,
                                     v_ms.ctypes.data, v_n.ctypes.data, v_ctx.ctypes.data, n_log, C.byref(c_iters),
                                     C.byref(c_err), cu), "sd_spec_batch_generate")
    if c_err.value:
        raise RuntimeError("s")
    if _timing is not None:
        class _Ms:                                                # (bench.py reads e0.elapsed_time(e1))
            def __init__(self, ms):
                self.ms = ms

            def elapsed_time(self, _other):
                return self.ms
        for i in range(min(c_iters.value, n_log)):
            _timing.setdefault("verify", []).append((_Ms(float(v_ms[i])), None, int(v_n[i]), float(v_ctx[i])))
    for s, (hs, acc, pa, qa) in zip(streams, keep):
        it = arr[s.idx]
        s.host = hs[:it.len].tolist()
        s.calls = it.calls
        s.acc_len = acc[:it.calls].tolist()
        s.acc_rate = np.minimum(1.0, pa[:it.calls * gamma].astype(np.float64) / qa[:it.calls * gamma].astype(np.float64)).tolist()
        s.draft_len, s.target_len = it.draft_len, it.target_len
        s.noise.seed, s.noise.draw = it.seed, it.draw
        s.draft._session.cache_len, s.target._session.cache_len = s.draft_len, s.target_len
        s.out = s.host
        if sum(1 for x in s.host if x == eos_token_id) > s.ori_eos:
            seen, cut = 0, len(s.host)
            for idx, x in enumerate(s.host):
                if x == eos_token_id:
                    seen += 1
                    if seen == s.ori_eos + 1:
                        cut = idx + 1
                        break
            s.out = s.host[:cut]

    outs = [torch.tensor([s.out], dtype=torch.int64, device=prefixes[i].device) for i, s in enumerate(streams)]
    if details:
        ds = [{"approx_time": 0, "target_time": 0, "other_time": 0, "acc_len": s.acc_len,
               "acc_rate": float(np.mean(s.acc_rate)) if s.acc_rate else 0.0, "target_call_times": s.calls,
               "approx_call_times": s.calls, "target_model_time": 0, "target_pre_cache_time": 0,
               "target_post_prob_time": 0} for s in streams]
        return outs, ds
    return outs
