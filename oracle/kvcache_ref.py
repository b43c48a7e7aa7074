"""Oracle KV-cache wrapper (test infrastructure, see oracle/__init__.py).

Restates the decoder-only core of reference sampling/kvcache_model.py:
__init__ :24-36, _forward_with_kvcache :141-252, _generate_with_kvcache :255-298,
generate :300-310, rollback :359-436 (choice=None and integer-choice branches).
Batch 1 on the north-star path; the ``multi`` / ``strategy="iid"`` / ``choice`` parts
(cache replication :180-200, :239-244, repeat :273-276, rollback :390-396, :433-436)
serve multi_speculative_sampling (oracle/multi_ref.py).
"""
from __future__ import annotations

from time import process_time_ns

import torch

from .noise import TorchGlobalNoise
from .sampling_ref import norm_logits, sample


class RefKVCacheModel:
    def __init__(self, model, temperature: float = 1, top_k: int = 0, top_p: float = 0, noise=None):
        self._model = model
        self._past_key_values = None
        self._prob_history = None
        self._temperature = temperature
        self._top_k = top_k
        self._top_p = top_p
        self.beam_rollback_flag = False
        self.forward_time_dict = {"_model_time": 0, "norm_prob_time": 0, "prepare_cache_time": 0}
        self.noise = noise or TorchGlobalNoise()
        self.rows_fed = []           # number of new tokens per forward (1, 2 after an all-accept, gamma+1, ...)

    def _normalise_rows(self, logits):
        # kvcache_model.py:167-168 / :235-236: one norm_logits call per position, written back in place
        out = torch.empty_like(logits)
        for i in range(logits.shape[-2]):
            out[:, i, :] = norm_logits(logits[:, i, :], self._temperature, self._top_k, self._top_p)
        return out

    def _forward_with_kvcache(self, input_ids: torch.Tensor) -> torch.Tensor:
        t0 = process_time_ns()
        if self._past_key_values is None:
            out = self._model(input_ids)                                  # prefill (:156)
            self.rows_fed.append(input_ids.shape[1])
            t1 = process_time_ns()
            self._prob_history = self._normalise_rows(out.logits)
        else:
            cached = self._past_key_values[0][0].shape[2]                 # (:175)
            width = input_ids.size(0)
            if self._past_key_values[0][0].size(0) < width:               # one cached row, width inputs (:180-192)
                self._past_key_values = [tuple(t.repeat(width, 1, 1, 1) for t in kv) for kv in self._past_key_values]
            fresh = input_ids[:, cached:]                                 # (:206)
            out = self._model(fresh, past_key_values=self._past_key_values, use_cache=True)
            self.rows_fed.append(fresh.shape[1])
            t1 = process_time_ns()
            fresh_probs = self._normalise_rows(out.logits)
            if self._prob_history.size(0) < width:                        # (:239-242)
                self._prob_history = self._prob_history.repeat(int(width / self._prob_history.size(0)), 1, 1)
            self._prob_history = torch.cat([self._prob_history, fresh_probs], dim=1)
        self._past_key_values = out.past_key_values
        self.forward_time_dict["_model_time"] += t1 - t0
        self.forward_time_dict["norm_prob_time"] += process_time_ns() - t1
        return self._prob_history[:, -1, :]

    @torch.no_grad()
    def forward_tree_attention(self, input_ids, prefix, extra_attention_mask, position_ids, gather_pos):
        """kvcache_model.py:38-136: one forward over the uncached prefix rows + the nodes of a draft token tree (extra
        attention mask, per-node position ids); every new row is normalised and appended to the history; returns the rows
        at gather_pos (row -1 of the tree = the last prefix position).  gather_pos is modified in place like there."""
        P = prefix.size(-1)
        if self._past_key_values is None:
            if prefix.size(0) < input_ids.size(0):
                prefix = prefix.repeat(input_ids.size(0), 1)
            ids = torch.cat((prefix, input_ids), dim=1)
            pos = torch.cat((torch.arange(P).view(1, -1).repeat(position_ids.size(0), 1), position_ids), dim=1)
            out = self._model(ids, extra_attention_mask=extra_attention_mask, position_ids=pos)
            self._prob_history = self._normalise_rows(out.logits)
        else:
            cached = self._past_key_values[0][0].shape[2]
            ids = torch.cat((prefix, input_ids), dim=1)[:, cached:]
            pos = torch.cat((torch.arange(P).view(1, -1).repeat(position_ids.size(0), 1), position_ids), dim=1)[:, cached:]
            out = self._model(ids, past_key_values=self._past_key_values, use_cache=True,
                              extra_attention_mask=extra_attention_mask, position_ids=pos)
            self._prob_history = torch.cat([self._prob_history, self._normalise_rows(out.logits)], dim=1)
        self._past_key_values = out.past_key_values
        gather_pos[:, 1] += P
        return self._prob_history[gather_pos[:, 0], gather_pos[:, 1]]

    @torch.no_grad()
    def rollback_tree_attention(self, input_idx, mask):
        """kvcache_model.py:326-353: keep, for each of the `width` outputs, the cache rows of batch element input_idx[w]
        at the positions where mask[w] is True (the prefix and the accepted path), compacted."""
        width = input_idx.numel()
        _, nh, _, d = self._past_key_values[0][0].size()

        def pick(t):
            return t[input_idx].transpose(1, 2)[mask].view(width, -1, nh, d).transpose(1, 2)
        self._past_key_values = [(pick(k), pick(v)) for k, v in self._past_key_values]
        V = self._prob_history.size(-1)
        self._prob_history = self._prob_history[input_idx][mask].view(width, -1, V)

    @torch.no_grad()
    def generate(self, input: torch.Tensor, gamma: int, multi: int = 1, strategy: str = "beam") -> torch.Tensor:
        x = input
        if strategy == "iid" and multi > 1:                               # (:273-274)
            x = x.repeat(multi, 1)
        elif multi > 1:
            raise NotImplementedError                                     # (:286-287) beam is served elsewhere
        for _ in range(gamma):                                            # (:279-293)
            q = self._forward_with_kvcache(x)
            x = torch.cat((x, sample(q, self.noise)), dim=1)
        return x

    @torch.no_grad()
    def rollback(self, end_pos: int, choice=None):
        assert self._past_key_values
        if choice is None:
            self._past_key_values = [(k[:, :, :end_pos, :], v[:, :, :end_pos, :]) for k, v in self._past_key_values]
            if self._prob_history is not None:
                self._prob_history = self._prob_history[:, :end_pos, :]
        else:                                                             # integer choice (:390-392, :433-434)
            c = int(choice)
            self._past_key_values = [(k[c:c + 1, :, :end_pos, :], v[c:c + 1, :, :end_pos, :])
                                     for k, v in self._past_key_values]
            if self._prob_history is not None:
                self._prob_history = self._prob_history[c:c + 1, :end_pos, :]
