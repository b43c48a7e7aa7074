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

from .._lib import lib, check, SdNormRow
from ..engine import batch_forward, SpecDecModel, Session, as_specdec_model, _stream, MAX_ROWS_PER_FORWARD, MAX_LOGIT_ROWS, check_token_ids
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
        # width-w replicas of generate(multi=w, strategy="iid") (kvcache_model.py:180-200, 273-276): replica 0 is this
        # object's own arenas; _width is the batch size the reference's cache would have right now
        self._replicas = []                             # [(Session, probs, err, tok32)] for replicas 1..
        self._width = 1

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
        if self._width > 1:                             # (width, H_kv, S, D) like the reference's repeated cache
            S = self._session.cache_len
            kvs = [self._session.kv] + [r[0].kv for r in self._replicas[: self._width - 1]]
            return [(torch.stack([kv[l, 0, :, :S, :] for kv in kvs]), torch.stack([kv[l, 1, :, :S, :] for kv in kvs]))
                    for l in range(self._session.kv.shape[0])]
        return self._session.past_key_values()

    @property
    def _prob_history(self):
        if self._probs is None:
            return None
        if self._width > 1:
            return torch.stack([self._probs[: self._hist_len]] +
                               [r[1][: self._hist_len] for r in self._replicas[: self._width - 1]]).to(self._model.probs_dtype)
        return self._probs[: self._hist_len].unsqueeze(0).to(self._model.probs_dtype)

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
            blk = min(MAX_LOGIT_ROWS, n_rows_out - done)
            end = first + done + blk
            logits = ses.forward(seq32[ses.cache_len:end], blk)
            t1 = process_time_ns()
            check(lib.sd_norm_probs(logits.data_ptr(), blk, V, logits.stride(0), float(self._temperature),
                                    int(self._top_k or 0), float(self._top_p or 0.0), self._model.norm_mode,
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
            e = noise.exponential(V, self._model.probs_dtype)
            e_ptr, seed, draw = e.data_ptr(), 0, 0
        check(lib.sd_norm_sample(logits.data_ptr(), V, float(self._temperature), int(self._top_k or 0),
                                 float(self._top_p or 0.0), self._model.norm_mode, self._probs[row].data_ptr(),
                                 self._err[row].data_ptr(),
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
        check_token_ids(input_ids[0, cached:S], self._model.cfg.vocab_size)
        self._tok32[cached:S] = input_ids[0, cached:S].to(device=self._tok32.device, dtype=torch.int32)
        n_new = S - cached
        rows = n_new if (self._full_history or cached > 0) else 1
        self.forward_rows(self._tok32, S, rows)
        self.check_errors(S - rows, S)
        return self._probs[S - 1].unsqueeze(0).to(self._model.probs_dtype)

    @torch.no_grad()
    def generate(self, input: torch.Tensor, gamma: int, decoder_input_ids=None, attention_mask=None,
                 copy_cache_index=None, multi: int = 1, strategy: str = "beam") -> torch.Tensor:
        """kvcache_model.py:300-310 -> :255-298: multi == 1, or multi > 1 with strategy="iid" (:273-276: the prefix
        is repeated multi times, every replica samples its own continuation; returns (multi, S + gamma))."""
        if multi > 1 or input.size(0) > 1 or self._width > 1:
            if strategy == "beam" and multi > 1:
                raise NotImplementedError                         # reference :286-287
            if multi > 1 and strategy != "iid":
                raise RuntimeError("Strategy Not Implemented " + strategy)      # reference :290
            return self._generate_iid(input, gamma, max(int(multi), 1))
        from .utils import sample
        noise = self._noise or HostTorchNoise(self._model.device)
        x = input
        for _ in range(gamma):
            q = self._forward_with_kvcache(x)
            nxt = sample(q, noise=noise)
            x = torch.cat((x, nxt.to(x.device)), dim=1)
        return x

    def _grow_width(self, W: int) -> None:
        """The reference's ``val.repeat(width, 1, 1, 1)`` / ``_prob_history.repeat(width, 1, 1)`` (kvcache_model.py:
        180-192, 239-242): replicas 1..W-1 get their own arenas holding a copy of replica 0's rows."""
        ses0 = self._session
        m = self._model
        while len(self._replicas) < W - 1:
            ses = m.new_session(ses0.max_seq)
            self._replicas.append((ses, torch.zeros_like(self._probs), torch.zeros_like(self._err),
                                   torch.zeros_like(self._tok32)))
        n, h = ses0.cache_len, self._hist_len
        for ses, probs, err, tok in self._replicas[self._width - 1: W - 1]:
            ses.kv[:, :, :, :n].copy_(ses0.kv[:, :, :, :n])
            ses.cache_len = n
            probs[:h].copy_(self._probs[:h])
            tok[: n + 1].copy_(self._tok32[: n + 1])
        self._width = W

    def _generate_iid(self, input: torch.Tensor, gamma: int, multi: int) -> torch.Tensor:
        assert input.dim() == 2
        x = input.repeat(multi, 1) if multi > 1 else input     # (:273-274)
        W, S0 = x.size(0), x.size(1)
        m = self._model
        V = m.cfg.vocab_size
        dev = m.device
        self._ensure(S0 + gamma + 1)
        ses0 = self._session
        if ses0.cache_len == 0:
            # no cache yet: the reference runs the (W, S0) batch through the model; every row of x is fed on its own
            # replica below, so replica 0 only needs its arenas to exist
            pass
        assert W >= self._width, "the batch cannot shrink without rollback(choice=)"
        if W > self._width:
            self._grow_width(W)
        noise = self._noise or HostTorchNoise(dev)
        on_dev = getattr(noise, "on_device", False)
        sess = [ses0] + [r[0] for r in self._replicas[: W - 1]]
        probs = [self._probs] + [r[1] for r in self._replicas[: W - 1]]
        errs = [self._err] + [r[2] for r in self._replicas[: W - 1]]
        toks = [self._tok32] + [r[3] for r in self._replicas[: W - 1]]
        serr = torch.zeros(W, dtype=torch.int32, device=dev)
        cached = ses0.cache_len
        for w in range(W):
            assert sess[w].cache_len == cached
            check_token_ids(x[w, cached:], V)
            toks[w][cached:S0] = x[w, cached:].to(device=dev, dtype=torch.int32)
        ld_bytes = self._probs.stride(0) * 4
        st = _stream()
        per_pass = MAX_ROWS_PER_FORWARD
        for i in range(gamma):
            upto = S0 + i
            n_new = upto - sess[0].cache_len
            # rows whose probabilities enter the history: all new ones once a cache exists or with full_history
            n_out = n_new if (self._full_history or sess[0].cache_len > 0) else 1
            assert n_new * 1 <= per_pass, "one replica's uncached rows exceed a stream-batched pass; prefill with multi=1 first"
            if on_dev:
                e_base, seed, draw0 = 0, noise.seed, noise.next_draws(W)
            else:
                e = noise.exponential_rows(W, V, m.probs_dtype)   # ONE (W, V) draw, like torch.multinomial on (W, V)
                e_base, seed, draw0 = e.data_ptr(), 0, 0
            grp = max(1, per_pass // n_new)
            for a in range(0, W, grp):
                ws = list(range(a, min(W, a + grp)))
                logits = batch_forward([sess[w] for w in ws], [toks[w] for w in ws], [n_new] * len(ws), [n_out] * len(ws))
                rows = (SdNormRow * (len(ws) * n_out))()
                k = 0
                for w in ws:
                    for r in range(n_out):
                        pos = upto - n_out + r
                        last = r == n_out - 1
                        rows[k].probs_out = probs[w].data_ptr() + pos * ld_bytes
                        rows[k].err = errs[w].data_ptr() + 4 * pos
                        rows[k].exp_noise = (e_base + w * V * 4) if (e_base and last) else None
                        rows[k].philox_seed = seed
                        rows[k].draw_index = draw0 + w
                        rows[k].tok_out = toks[w].data_ptr() + 4 * upto if last else serr.data_ptr()
                        rows[k].sample_err = serr.data_ptr() + 4 * w if last else None
                        k += 1
                if n_out == 1:
                    check(lib.sd_norm_batch(logits.data_ptr(), k, V, logits.stride(0), float(self._temperature),
                                            int(self._top_k or 0), float(self._top_p or 0.0), m.norm_mode, rows, 1,
                                            self._norm_ws.data_ptr(), st), "sd_norm_batch")
                else:
                    # several history rows per replica: normalise all of them, then sample each replica's last row
                    check(lib.sd_norm_batch(logits.data_ptr(), k, V, logits.stride(0), float(self._temperature),
                                            int(self._top_k or 0), float(self._top_p or 0.0), m.norm_mode, rows, 0,
                                            self._norm_ws.data_ptr(), st), "sd_norm_batch")
                    for w in ws:
                        check(lib.sd_sample(probs[w][upto - 1].data_ptr(), V, (e_base + w * V * 4) if e_base else None,
                                            seed, draw0 + w, toks[w][upto:].data_ptr(), serr[w:].data_ptr(), m.norm_mode, st),
                              "sd_sample")
            if self._hist_len == 0:
                self._hist_lo = upto - n_out
            self._hist_len = upto
            if bool(serr.any()):
                raise RuntimeError("prob error")
            for w in range(W):
                if bool(errs[w][upto - n_out:upto].any()):
                    raise RuntimeError("norm logits error")
        out = torch.stack([toks[w][: S0 + gamma] for w in range(W)]).to(torch.int64)
        return out.to(input.device)

    @torch.no_grad()
    def rollback(self, end_pos: int, choice=None):
        """kvcache_model.py:359-436: choice=None trims every replica (O(1), nothing is copied); an integer choice keeps
        only that replica (``k[choice:choice+1, :, :end_pos, :]``, :390-392, 433-434): its arenas become this object's."""
        assert self._session is not None and self._session.cache_len > 0
        end_pos = int(end_pos)
        if choice is not None:
            c = int(choice)
            if not 0 <= c < self._width:
                raise IndexError(f"choice {c} outside the {self._width} cached replicas")
            if c > 0:
                ses, probs, err, tok = self._replicas[c - 1]
                self._replicas[c - 1] = (self._session, self._probs, self._err, self._tok32)
                self._session, self._probs, self._err, self._tok32 = ses, probs, err, tok
            self._width = 1
        self._session.rollback(end_pos)
        for ses, _, _, _ in self._replicas[: self._width - 1]:
            ses.rollback(end_pos)
        self._hist_len = min(self._hist_len, end_pos)

    # -- tree attention ------------------------------------------------
    @torch.no_grad()
    def forward_tree_attention(self, input_ids: torch.Tensor, prefix: torch.Tensor, extra_attention_mask: torch.Tensor,
                               position_ids: torch.Tensor, gather_pos: torch.Tensor) -> torch.Tensor:
        """reference kvcache_model.py:38-136 for one input sequence (extra_sample_cnt == 1): the uncached prefix rows go
        through the ordinary forward, then ONE tree forward (sd_session_forward_tree) scores every node of the draft
        token tree: node i = input_ids[0, i] at position position_ids[0, i], visible keys = the whole prefix + the nodes
        marked in extra_attention_mask[0, i, prefix_len:].  All new rows are normalised into the history; returns the
        rows at gather_pos (slot -1 = the last prefix position; gather_pos[:, 1] is shifted in place, as there)."""
        if input_ids.size(0) != 1 or prefix.size(0) != 1:
            raise NotImplementedError("forward_tree_attention: one input sequence per call (extra_sample_cnt == 1)")
        m = self._model
        V, dev = m.cfg.vocab_size, m.device
        P, N = int(prefix.size(-1)), int(input_ids.size(1))
        if N > MAX_LOGIT_ROWS:
            raise ValueError(f"a tree of {N} nodes exceeds one verify pass ({MAX_LOGIT_ROWS} rows)")
        self._ensure(P + N + 1)
        ses = self._session
        cached = ses.cache_len
        assert cached <= P
        if cached < P:                                            # uncached prefix rows: ordinary causal forward
            check_token_ids(prefix[0, cached:], V)
            self._tok32[cached:P] = prefix[0, cached:].to(device=dev, dtype=torch.int32)
            self.forward_rows(self._tok32, P, P - cached)
            self.check_errors(cached, P)
        check_token_ids(input_ids, V)
        tree_tok = input_ids[0].to(device=dev, dtype=torch.int32).contiguous()
        pos_host = (C.c_int32 * N)(*[int(x) for x in position_ids[0].tolist()])
        em = extra_attention_mask[0, :, P:P + N].to("cpu")
        bits = [int(sum(1 << j for j in range(N) if bool(em[i, j]))) for i in range(N)]
        masks = (C.c_uint64 * N)(*bits)
        logits = torch.empty((N, V), dtype=torch.float32, device=dev)
        t0 = process_time_ns()
        check(lib.sd_session_forward_tree(ses.handle, tree_tok.data_ptr(), pos_host, masks, N, P, logits.data_ptr(),
                                          logits.stride(0), _stream()), "sd_session_forward_tree")
        t1 = process_time_ns()
        check(lib.sd_norm_probs(logits.data_ptr(), N, V, logits.stride(0), float(self._temperature), int(self._top_k or 0),
                                float(self._top_p or 0.0), m.norm_mode, self._probs[P].data_ptr(), self._probs.stride(0),
                                self._err[P].data_ptr(), self._norm_ws.data_ptr() if N <= ses.max_rows else None, _stream()),
              "sd_norm_probs")
        self.forward_time_dict["_model_time"] += t1 - t0
        self.forward_time_dict["norm_prob_time"] += process_time_ns() - t1
        ses.cache_len = P + N                                     # tree rows occupy arena slots P .. P+N-1
        self._hist_len = P + N
        self.check_errors(P, P + N)
        gather_pos[:, 1] += P
        return self._probs[gather_pos[:, 1].to(dev)].to(m.probs_dtype)

    def beam_rollback(self, beam_idx, choice):
        """reference kvcache_model.py:312-324 (parity unpinned, see sampling/beam.py): the draft's cache as it was after
        beam step `beam_idx`, of beam `choice` only."""
        if getattr(self, "_beam", None) is None:
            raise RuntimeError("beam_rollback before beam_sample_with_kv_cache")
        self._beam.beam_rollback(int(beam_idx), int(choice))

    @torch.no_grad()
    def rollback_tree_attention(self, input_idx: torch.Tensor, mask: torch.Tensor):
        """reference kvcache_model.py:326-353 for one kept path: the cache and probability rows at the positions where
        mask[0] is True (the prefix and the accepted nodes) are compacted to the front (sd_session_compact_kv gathers
        the KV rows of every layer / head in one launch)."""
        if input_idx.numel() != 1 or int(input_idx.reshape(-1)[0]) != 0:
            raise NotImplementedError("rollback_tree_attention: one kept path of input 0 (extra_sample_cnt == 1)")
        ses = self._session
        keep = mask.reshape(mask.shape[-2] if mask.dim() > 1 else 1, -1)[0].to("cpu")
        L = int(keep.numel())
        assert L <= ses.cache_len
        P = 0
        while P < L and bool(keep[P]):                            # leading run of kept positions stays where it is
            P += 1
        idx = [i - P for i in range(P, L) if bool(keep[i])]
        if idx:
            idx_dev = torch.tensor(idx, dtype=torch.int32, device=self._model.device)
            check(lib.sd_session_compact_kv(ses.handle, P, idx_dev.data_ptr(), len(idx), _stream()), "sd_session_compact_kv")
            src = self._probs[P:L][idx_dev.long()].clone()
            self._probs[P:P + len(idx)] = src
        ses.cache_len = P + len(idx)
        self._hist_len = P + len(idx)

    def beam_sample_with_kv_cache(self, prefix, gamma, num_beams, top_k=None, top_p=None, **kwargs):
        """reference kvcache_model.py:439-567 -> :571-1025, decoder-only, return_intermediate_results=True,
        optimization=False (the driver's only use, speculative_sampling.py:84-98): `gamma` steps of beam sampling over
        `num_beams` KV arenas of this model.  Parity unpinned (sampling/beam.py)."""
        from .beam import kv_beam_sample_with_kv_cache
        return kv_beam_sample_with_kv_cache(self, prefix, gamma, num_beams, top_k=top_k, top_p=top_p, **kwargs)

    def beam_sample(self, *a, **k):
        raise NotImplementedError("call beam_sample_with_kv_cache (reference kvcache_model.py:439-567): beam_sample's "
                                  "own arguments are transformers 4.35 objects (BeamScorer, LogitsProcessorList)")
