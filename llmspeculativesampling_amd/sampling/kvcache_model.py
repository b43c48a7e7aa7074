"""Drop-in for the batch-1 decoder-only core of reference sampling/kvcache_model.py.

Same constructor, attributes and methods (kvcache_model.py:24-36, 141-252, 255-310, 359-436), but
the state lives in preallocated device arenas: the KV cache is appended in-kernel and rolled back
by moving a length, and the probability history is an arena indexed by absolute position instead
of a tensor re-concatenated every step.
"""
from __future__ import annotations

import ctypes as C
from time import process_time_ns
from typing import Optional

import torch

from .._lib import lib, check
from ..engine import SpecDecModel, Session, as_specdec_model, _stream, MAX_ROWS_PER_FORWARD
from ..noise import HostTorchNoise


class KVCacheModel:
    def __init__(self, model, temperature: float = 1, top_k: int = 0, top_p: float = 0, *,
                 max_seq: Optional[int] = None, noise=None, full_history: bool = True):
        self._model = as_specdec_model(model)
        self._temperature = temperature
        self._top_k = top_k
        self._top_p = top_p
        self.beam_rollback_flag = False
        self.forward_time_dict = {"_model_time": 0, "norm_prob_time": 0, "prepare_cache_time": 0}
        self._max_seq = max_seq
        self._noise = noise
        self._full_history = full_history
        self._session: Optional[Session] = None
        self._probs: Optional[torch.Tensor] = None      # [max_seq][V] fp32, row = absolute position
        self._err: Optional[torch.Tensor] = None
        self._tok32: Optional[torch.Tensor] = None
        self._hist_len = 0                              # rows of _probs that are in the history
        self._hist_lo = 0                               # first row that was actually normalised
        self.event_log = None                           # bench: list of (start_evt, end_evt, n_new, upto) per forward

    # -- lazily sized arenas ----------------------------------------------------------------
    def _ensure(self, need: int):
        if self._session is not None:
            if need > self._session.max_seq:
                raise RuntimeError(f"sequence length {need} exceeds the KV arena ({self._session.max_seq}); "
                                   "construct KVCacheModel(..., max_seq=) larger")
            return
        m = self._model
        cap = self._max_seq or min(m.max_pos, max(need + 1024, 2048))
        self._session = m.new_session(cap)
        dev = m.device
        self._probs = torch.zeros((self._session.max_seq, m.cfg.vocab_size), dtype=torch.float32, device=dev)
        self._err = torch.zeros(self._session.max_seq, dtype=torch.int32, device=dev)
        self._tok32 = torch.zeros(self._session.max_seq + 1, dtype=torch.int32, device=dev)
        self._norm_ws = torch.empty(lib.sd_norm_workspace_bytes(self._session.max_rows), dtype=torch.uint8, device=dev)

    # -- attributes the reference exposes ---------------------------------------------------
    @property
    def _past_key_values(self):
        if self._session is None or self._session.cache_len == 0:
            return None
        return self._session.past_key_values()

    @property
    def _prob_history(self):
        if self._probs is None:
            return None
        return self._probs[: self._hist_len].unsqueeze(0)

    @property
    def cache_len(self) -> int:
        return 0 if self._session is None else self._session.cache_len

    # -- device-resident core used by the decode loops --------------------------------------
    def forward_rows(self, seq32: torch.Tensor, upto: int, n_rows_out: int) -> None:
        """Feed tokens seq32[cache_len:upto] and write normalised probabilities for the last
        ``n_rows_out`` of them into the arena rows (upto-n_rows_out .. upto-1)."""
        ses = self._session
        t0 = process_time_ns()
        n_new = upto - ses.cache_len
        assert n_new >= 1 and 1 <= n_rows_out <= n_new
        done = 0
        st = _stream()
        V = self._model.cfg.vocab_size
        first = upto - n_rows_out
        ev0 = None
        if self.event_log is not None:
            ev0 = torch.cuda.Event(enable_timing=True)
            ev0.record()
        while done < n_rows_out:                        # logits buffer holds max_rows rows at a time
            # feed everything up to the end of this block of output rows
            blk = min(MAX_ROWS_PER_FORWARD, n_rows_out - done)
            end = first + done + blk
            logits = ses.forward(seq32[ses.cache_len:end], blk)
            t1 = process_time_ns()
            check(lib.sd_norm_probs(logits.data_ptr(), blk, V, logits.stride(0), float(self._temperature),
                                    int(self._top_k or 0), float(self._top_p or 0.0), 0,
                                    self._probs[end - blk].data_ptr(), self._probs.stride(0),
                                    self._err[end - blk].data_ptr(), self._norm_ws.data_ptr(), st), "sd_norm_probs")
            self.forward_time_dict["norm_prob_time"] += process_time_ns() - t1
            self.forward_time_dict["_model_time"] += t1 - t0
            t0 = process_time_ns()
            done += blk
        if ev0 is not None:
            ev1 = torch.cuda.Event(enable_timing=True)
            ev1.record()
            self.event_log.append((ev0, ev1, n_new, upto))
        if self._hist_len == 0:
            self._hist_lo = first
        self._hist_len = upto

    def forward_sample(self, seq32: torch.Tensor, upto: int, noise, samp_err: torch.Tensor) -> None:
        """One draft / autoregressive step: feed seq32[cache_len:upto], normalise the last row into the arena
        and sample the next token straight into seq32[upto] (one fused launch after the forward)."""
        ses = self._session
        t0 = process_time_ns()
        ev0 = None
        n_new = upto - ses.cache_len
        if self.event_log is not None:
            ev0 = torch.cuda.Event(enable_timing=True)
            ev0.record()
        logits = ses.forward(seq32[ses.cache_len:upto], 1)
        t1 = process_time_ns()
        V = self._model.cfg.vocab_size
        row = upto - 1
        if noise.on_device:
            e_ptr, seed, draw = None, noise.seed, noise.next_draws(1)
        else:
            e = noise.exponential(V)
            e_ptr, seed, draw = e.data_ptr(), 0, 0
        check(lib.sd_norm_sample(logits.data_ptr(), V, float(self._temperature), int(self._top_k or 0),
                                 float(self._top_p or 0.0), 0, self._probs[row].data_ptr(), self._err[row].data_ptr(),
                                 e_ptr, seed, draw, seq32[upto].data_ptr(), samp_err.data_ptr(), self._norm_ws.data_ptr(),
                                 _stream()),
              "sd_norm_sample")
        self.forward_time_dict["norm_prob_time"] += process_time_ns() - t1
        self.forward_time_dict["_model_time"] += t1 - t0
        if ev0 is not None:
            ev1 = torch.cuda.Event(enable_timing=True)
            ev1.record()
            self.event_log.append((ev0, ev1, n_new, upto))
        if self._hist_len == 0:
            self._hist_lo = row
        self._hist_len = upto

    def check_errors(self, lo: int, hi: int) -> None:
        if bool(self._err[lo:hi].any()):
            raise RuntimeError("norm logits error")       # reference utils.py:207

    # -- reference API ----------------------------------------------------------------------
    def _forward_with_kvcache(self, input_ids: torch.Tensor, use_debug=False, attention_mask=None,
                              decoder_input_ids=None, copy_cache_index=None) -> torch.Tensor:
        """kvcache_model.py:141-252 for batch 1: returns the last position's distribution (1, V)."""
        assert input_ids.dim() == 2 and input_ids.size(0) == 1, "input batch size must be 1"
        S = input_ids.size(1)
        self._ensure(S + 1)
        ses = self._session
        cached = ses.cache_len
        self._tok32[cached:S] = input_ids[0, cached:S].to(device=self._tok32.device, dtype=torch.int32)
        n_new = S - cached
        rows = n_new if (self._full_history or cached > 0) else 1
        self.forward_rows(self._tok32, S, rows)
        self.check_errors(S - rows, S)
        return self._probs[S - 1].unsqueeze(0)

    @torch.no_grad()
    def generate(self, input: torch.Tensor, gamma: int, decoder_input_ids=None, attention_mask=None,
                 copy_cache_index=None, multi: int = 1, strategy: str = "beam") -> torch.Tensor:
        """kvcache_model.py:300-310 -> :255-298 (multi == 1)."""
        if multi != 1:
            raise NotImplementedError("multi-draft strategies (reference kvcache_model.py:273-290) are out of scope")
        from .utils import sample
        noise = self._noise or HostTorchNoise(self._model.device)
        x = input
        for _ in range(gamma):
            q = self._forward_with_kvcache(x)
            nxt = sample(q, noise=noise)
            x = torch.cat((x, nxt.to(x.device)), dim=1)
        return x

    @torch.no_grad()
    def rollback(self, end_pos: int, choice=None):
        """kvcache_model.py:359-436, choice=None branch: O(1), nothing is copied."""
        if choice is not None:
            raise NotImplementedError("rollback(choice=) serves the multi-draft variants (out of scope)")
        assert self._session is not None and self._session.cache_len > 0
        self._session.rollback(end_pos)
        self._hist_len = min(self._hist_len, int(end_pos))

    # -- out-of-scope beam / tree methods keep their names --------
    def forward_tree_attention(self, *a, **k):
        raise NotImplementedError("tree attention (reference kvcache_model.py:38-136) is out of scope")

    def beam_rollback(self, *a, **k):
        raise NotImplementedError("beam rollback (reference kvcache_model.py:312-324) is out of scope")

    def rollback_tree_attention(self, *a, **k):
        raise NotImplementedError("tree rollback (reference kvcache_model.py:326-353) is out of scope")

    def beam_sample_with_kv_cache(self, *a, **k):
        raise NotImplementedError("beam sampling (reference kvcache_model.py:439-567) is out of scope")

    def beam_sample(self, *a, **k):
        raise NotImplementedError("beam sampling (reference kvcache_model.py:571-1025) is out of scope")
