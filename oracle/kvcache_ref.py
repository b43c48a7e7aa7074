"""Oracle KV-cache wrapper (test infrastructure, see oracle/__init__.py).

Restates the batch-1, decoder-only core of reference sampling/kvcache_model.py:
__init__ :24-36, _forward_with_kvcache :141-252, _generate_with_kvcache :255-298,
generate :300-310, rollback :359-436 (choice=None branch).
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
            fresh = input_ids[:, cached:]                                 # (:206)
            out = self._model(fresh, past_key_values=self._past_key_values, use_cache=True)
            self.rows_fed.append(fresh.shape[1])
            t1 = process_time_ns()
            self._prob_history = torch.cat([self._prob_history, self._normalise_rows(out.logits)], dim=1)
        self._past_key_values = out.past_key_values
        self.forward_time_dict["_model_time"] += t1 - t0
        self.forward_time_dict["norm_prob_time"] += process_time_ns() - t1
        return self._prob_history[:, -1, :]

    @torch.no_grad()
    def generate(self, input: torch.Tensor, gamma: int) -> torch.Tensor:
        x = input
        for _ in range(gamma):                                            # (:279-293)
            q = self._forward_with_kvcache(x)
            x = torch.cat((x, sample(q, self.noise)), dim=1)
        return x

    @torch.no_grad()
    def rollback(self, end_pos: int):
        assert self._past_key_values
        self._past_key_values = [(k[:, :, :end_pos, :], v[:, :, :end_pos, :]) for k, v in self._past_key_values]
        if self._prob_history is not None:
            self._prob_history = self._prob_history[:, :end_pos, :]
