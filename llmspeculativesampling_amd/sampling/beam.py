"""Drop-in for reference sampling/speculative_sampling.py:18-581 (``beam_speculative_sampling_v2``, the tree-attention beam
variant) with ``extra_sample_cnt == 1`` (one input sequence per tree verify), and for the
draft side it rests on: ``KVCacheModel.beam_sample_with_kv_cache`` / ``beam_sample`` / ``beam_rollback`` (reference
sampling/kvcache_model.py:439-567, 571-1025, 312-324).

**Parity unpinned.**  The reference's draft side is HF beam sampling around transformers 4.35.2's ``BeamSearchScorer`` and
``GenerationMixin`` internals, none of which exist in this image (transformers 5.15): the reference's beam path cannot
be run here, so nothing could be recorded from it, and the library is not stood in for.  What the driver consumes of that
code does not involve the scorer at all (``optimization=False``: ``process`` is never called, the ``finalize`` output is
never read), so the per-step intermediate results are restated from the source - oracle/beam_ref.py on the CPU, this
module on the device - and the two are held to each other token for token (tests/).  The TARGET side driven from here
(``forward_tree_attention`` / ``rollback_tree_attention``, ``get_seq_att_mask``, ``get_num_acc_prob``) is pinned by G9.

Mechanism.  Every model forward runs in libspecdec: the draft's beams are ``num_beams`` sessions (KV arenas) of the draft
model that share one pass over its weights per beam step (``sd_batch_forward``); the tree verify is
``sd_session_forward_tree`` and the accepted path is compacted by ``sd_session_compact_kv``.  Where the reference
re-indexes the whole cache of every beam after every step (``_reorder_cache``, kvcache_model.py:901-904) and keeps a
reference to every step's cache for ``beam_rollback`` (:768), only the rows written during the current call differ
between beams, so the reorder and the per-step snapshots move those few rows.  The beam bookkeeping between forwards -
log-softmax + beam score, the two HF warpers, the joint softmax over (beam, token), the top-n draw and the index
arithmetic of the verification loop - is a few elementwise / sort ops on ``num_beams x V`` numbers per step; it runs as
torch device ops, like the reference's, and is not part of the benchmarked path.
"""
from __future__ import annotations

import ctypes as C
import math
from time import process_time_ns
from typing import List, Optional

import numpy as np
import torch

from .._lib import lib, check
from ..engine import as_specdec_model, batch_forward, _stream, check_token_ids, same_device
from ..noise import HostTorchNoise
from .kvcache_model import KVCacheModel
from .speculative_sampling import _make_noise
from .utils import get_expect_cnt_by_thres, get_num_acc_prob, get_seq_att_mask, max_fn, norm_logits


# --------------------------------------------------------------------------- randomness
class _Draws:
    """The two kinds of draw of this variant on top of any noise provider: Exp(1) rows for ``torch.multinomial`` (one draw
    of the distribution's shape per call, utils.py:221) and single uniforms (``torch.rand(1)``, speculative_sampling.py:288)."""

    def __init__(self, noise, device):
        self.noise, self.device = noise, device

    def exponential(self, n: int, dtype=torch.float32) -> torch.Tensor:
        nz = self.noise
        if getattr(nz, "on_device", False):
            out = torch.empty(n, dtype=torch.float32, device=self.device)
            check(lib.sd_philox_exp(nz.seed, nz.next_draws(1), n, out.data_ptr(), _stream()), "sd_philox_exp")
            return out
        return nz.exponential(n, dtype)

    def uniform(self) -> float:
        nz = self.noise
        if getattr(nz, "on_device", False):
            out = torch.empty(1, dtype=torch.float32, device=self.device)
            check(lib.sd_philox_uniform(nz.seed, nz.next_draws(1), 1, out.data_ptr(), _stream()), "sd_philox_uniform")
            return float(out)
        return float(nz.uniform_one())

    def uniform64(self) -> float:
        nz = self.noise
        if getattr(nz, "on_device", False):
            return self.uniform()
        return float(nz.uniform64_one())


def _hf_top_k(scores: torch.Tensor, top_k: int) -> torch.Tensor:
    """transformers TopKLogitsWarper (used at kvcache_model.py:497-498)."""
    k = min(int(top_k), scores.size(-1))
    return scores.masked_fill(scores < torch.topk(scores, k)[0][..., -1, None], -float("inf"))


def _hf_top_p(scores: torch.Tensor, top_p: float) -> torch.Tensor:
    """transformers TopPLogitsWarper, min_tokens_to_keep = 1 (used at kvcache_model.py:499-500)."""
    srt, idx = torch.sort(scores, descending=False)
    cum = srt.softmax(dim=-1).cumsum(dim=-1)
    rm = cum <= (1 - top_p)
    rm[..., -1:] = False
    return scores.masked_fill(rm.scatter(1, idx, rm), -float("inf"))


def _sample_n(probs: torch.Tensor, num_samples: int, draws: _Draws) -> torch.Tensor:
    """reference utils.py:213-233 for num_samples >= 1 on a device tensor: multinomial without replacement = the top-n of
    p / Exp(1) (ATen's exponential trick, one draw over p's shape); with fewer non-zero entries than samples the reference
    draws WITH replacement (:214-215) - ATen's inverse-CDF sampler: sequential fp32 prefix sums of the (few) non-zero
    entries, normalised by their total, one double uniform per sample; a draw on a (near-)zero entry becomes the mode."""
    if not bool((probs.max() < float("inf")) & (probs.min() >= 0)) or bool((probs.sum(-1) == 0).any()):
        raise RuntimeError("prob error")
    # (`numel(probs.nonzero())`, utils.py:214: the count times the tensor's rank - a (1, N) row switches later than a 1-D one)
    if int(torch.count_nonzero(probs)) * probs.dim() < num_samples:
        flat = probs.reshape(-1)
        nz = flat.nonzero().reshape(-1)
        pre = np.cumsum(flat[nz].float().cpu().numpy(), dtype=np.float32)
        cum = (pre / pre[-1]).astype(np.float64)
        nz_host = nz.tolist()
        picks = [nz_host[min(int(np.searchsorted(cum, draws.uniform64(), side="left")), len(nz_host) - 1)] for _ in range(num_samples)]
        idx = torch.tensor(picks, dtype=torch.long, device=probs.device).reshape(probs.shape[:-1] + (num_samples,))
    else:
        lowp = probs.dtype if probs.dtype in (torch.bfloat16, torch.float16) else torch.float32
        q = draws.exponential(probs.numel(), lowp).reshape(probs.shape).to(probs.dtype)
        idx = torch.topk(probs / q, num_samples, dim=-1).indices
    mask = torch.gather(probs, -1, idx) < 1e-9
    if bool(mask.any()):
        idx[mask] = int(torch.argmax(probs))
    return idx


# --------------------------------------------------------------------------- the draft's beams
class BeamDraft:
    """``num_beams`` KV arenas of the draft model + what the reference's ``beam_past_key_values`` list would hold."""

    def __init__(self, model, num_beams: int, cap: int):
        self.model, self.nb, self.cap = model, int(num_beams), int(cap)
        self.sessions = [model.new_session(cap) for _ in range(self.nb)]
        dev = model.device
        self.toks = [torch.zeros(cap + 1, dtype=torch.int32, device=dev) for _ in range(self.nb)]
        self.cache_len = 0          # positions session 0 holds (the reference's single cached row between calls)
        self.sync_len = 0           # positions below which every session's arena equals session 0's
        self.base = 0               # prefix length of the current / last call
        self.snaps: List[Optional[torch.Tensor]] = []      # per beam step: the beams' rows [base, base + step), pre-reorder

    def _rows(self, lo: int, hi: int) -> torch.Tensor:
        return torch.stack([s.kv[:, :, :, lo:hi] for s in self.sessions])

    @torch.no_grad()
    def beam_sample(self, prefix: torch.Tensor, gamma: int, top_k, top_p, padding_input_cnt: int, draws: _Draws):
        """kvcache_model.py:571-1025 (decoder-only, return_intermediate_results=True, optimization=False)."""
        nb, m, dev = self.nb, self.model, self.model.device
        V = m.cfg.vocab_size
        P = int(prefix.size(-1))
        assert prefix.size(0) == 1 and P + gamma + 1 <= self.cap
        c0 = self.cache_len
        assert c0 < P, "the prefix must extend the cached positions"
        check_token_ids(prefix, V)
        ids32 = prefix[0].to(device=dev, dtype=torch.int32)
        ses0 = self.sessions[0]
        ses0.cache_len = c0
        # step 0: all beams are copies of ONE cached row (kvcache_model.py:519-525) and feed the same uncached tokens, so
        # the forward runs once and its K / V rows are copied to the other arenas
        self.toks[0][c0:P] = ids32[c0:P]
        logits0 = ses0.forward(self.toks[0][c0:P], 1).clone()
        lo = min(self.sync_len, c0)
        for b in range(1, nb):
            self.sessions[b].kv[:, :, :, lo:P].copy_(ses0.kv[:, :, :, lo:P])
            self.sessions[b].cache_len = P
            self.toks[b][:P] = ids32
        self.toks[0][:P] = ids32
        self.sync_len = P
        self.base = P
        self.snaps = []
        input_ids = prefix.to(dev).repeat_interleave(nb, dim=0)
        beam_scores = torch.zeros(nb, dtype=torch.float32, device=dev)
        if padding_input_cnt > 0:
            beam_scores[-padding_input_cnt:] = float("-inf")
        input_index = torch.arange(nb, dtype=torch.long, device=dev)
        all_seq, all_beam_idx, all_next_token, all_score, all_prob, all_input_idx = [], [], [], [], [], [input_index]
        for step in range(gamma):
            if step == 0:
                logits = logits0.expand(nb, V)
            else:
                logits = batch_forward(self.sessions, self.toks, [1] * nb, [1] * nb).clone()
            self.snaps.append(self._rows(P, P + step) if step else None)       # the caches BEFORE the reorder (:768)
            lg = logits.to(m.probs_dtype)                                      # OPT keeps its logits in the weight dtype
            scores = torch.nn.functional.log_softmax(lg, dim=-1)
            nts = scores + beam_scores[:, None].expand_as(scores)
            if top_k is not None and top_k > 0:
                nts = _hf_top_k(nts, top_k)
            if top_p is not None and top_p > 0:
                nts = _hf_top_p(nts, top_p)
            nts = nts.reshape(1, nb * V)
            probs = torch.nn.functional.softmax(nts, dim=-1)
            nxt = _sample_n(probs, nb, draws)
            nts_g = torch.clamp(torch.gather(nts, -1, nxt), min=-1e10)
            beam_idx = torch.div(nxt, V, rounding_mode="floor").squeeze(0)
            beam_next = (nxt % V).squeeze(0)
            beam_scores = nts_g.squeeze(0).to(torch.float32)
            all_seq.append(input_ids)
            all_beam_idx.append(beam_idx)
            all_next_token.append(beam_next)
            all_score.append(torch.gather(probs, -1, (beam_idx * V + beam_next).view(1, -1)).view(-1))
            all_prob.append(probs)
            input_index = input_index[beam_idx]
            all_input_idx.append(input_index)
            input_ids = torch.cat([input_ids[beam_idx, :], beam_next.unsqueeze(-1)], dim=-1)
            # _reorder_cache (:901-904): beam b continues from parent beam_idx[b]; only rows [P, P + step) differ between beams
            if step > 0:
                rows = self.snaps[-1][beam_idx]
                for b in range(nb):
                    self.sessions[b].kv[:, :, :, P:P + step].copy_(rows[b])
            tail = input_ids[:, P:].to(torch.int32)
            for b in range(nb):
                self.toks[b][P:P + step + 1] = tail[b]
        all_seq.append(input_ids)
        return all_seq, all_beam_idx, all_next_token, all_score, all_prob, all_input_idx

    @torch.no_grad()
    def beam_rollback(self, beam_idx: int, choice) -> None:
        """kvcache_model.py:312-324 + rollback(None, choice) (:393-395): the cache after the forward of beam step
        `beam_idx` (of the last step when every level was accepted), of beam `choice` only - it becomes session 0's."""
        assert beam_idx >= 0 and self.snaps
        j = beam_idx - 1 if beam_idx == len(self.snaps) else beam_idx
        c = int(choice)
        P = self.base
        if j > 0:
            self.sessions[0].kv[:, :, :, P:P + j].copy_(self.snaps[j][c])
        self.sessions[0].cache_len = P + j
        self.cache_len = P + j
        self.sync_len = min(self.sync_len, P)


def _beam_of(kvm: KVCacheModel, num_beams: int, cap: Optional[int] = None) -> BeamDraft:
    bd = getattr(kvm, "_beam", None)
    if bd is None or bd.nb != int(num_beams):
        m = kvm._model
        cap = int(cap or kvm._max_seq or min(m.max_pos, 2048))
        bd = BeamDraft(m, num_beams, cap)
        kvm._beam = bd
    return bd


def kv_beam_sample_with_kv_cache(kvm: KVCacheModel, prefix, gamma, num_beams, top_k=None, top_p=None, acc_rate_head=None,
                                 acc_rate_thres=0.4, ret_seq_scores=False, return_intermediate_results=False,
                                 padding_input_cnt=0, optimization=False, **kwargs):
    """KVCacheModel.beam_sample_with_kv_cache (reference kvcache_model.py:439-567): returns the tuple the driver unpacks
    (:98) with None in place of the BeamSampleDecoderOnlyOutput nobody reads."""
    if optimization:
        raise NotImplementedError("optimization=True routes the draws through BeamSearchScorer.process (absent here)")
    if not return_intermediate_results:
        raise NotImplementedError("only the intermediate results (return_intermediate_results=True) are produced")
    draws = _Draws(kvm._noise or HostTorchNoise(kvm._model.device), kvm._model.device)
    bd = _beam_of(kvm, num_beams)
    res = bd.beam_sample(prefix, int(gamma), top_k, top_p, int(padding_input_cnt), draws)
    kvm.beam_rollback_flag = False
    return (None,) + tuple(res)


# --------------------------------------------------------------------------- the driver loop
@torch.no_grad()
def beam_speculative_sampling_v2(prefix: torch.Tensor, approx_model, target_model, eos_token_id, pad_token_id, max_len: int,
                                 gamma: int = 4, width: int = 8, num_beams: int = 8, min_num_beams: int = 1,
                                 extra_sample_cnt: int = -1, expect_thres: float = 0.7, temperature: float = 1,
                                 top_k: int = 0, top_p: float = 0, verbose: bool = False, random_seed: int = None,
                                 details: bool = False, debug_dict=None, *, rng=None):
    """reference speculative_sampling.py:18-581 (signature, return value, ``details`` keys, draw order, EOS rule;
    ``random_seed`` is accepted and unused, as there).  ``rng`` as in ``speculative_sampling``."""
    if extra_sample_cnt == -1:
        extra_sample_cnt = num_beams
    if extra_sample_cnt != 1:
        raise NotImplementedError("beam_speculative_sampling_v2: only extra_sample_cnt == 1 (one input sequence per tree "
                                  "verify) is built; the reference harness also sweeps 2 (evaluation.py:862)")
    draft_m, target_m = as_specdec_model(approx_model), as_specdec_model(target_model)
    assert not draft_m.cfg.is_encoder_decoder and not target_m.cfg.is_encoder_decoder
    same_device(draft_m, target_m)
    dev = target_m.device
    nb = int(num_beams)
    padding_input_cnt = nb - extra_sample_cnt
    if pad_token_id is None:
        pad_token_id = eos_token_id
    assert prefix.shape[0] == 1, "input batch size must be 1"
    if gamma * nb > 64:
        raise ValueError(f"a tree of {gamma} x {nb} nodes exceeds one verify pass (64 nodes)")
    prefix = prefix.to(dev)
    seq_len = prefix.shape[1]
    ori_eos_cnt = int((prefix == eos_token_id).int().sum())
    T = seq_len + max_len
    cap = T + gamma * nb + gamma + 4
    noise = _make_noise(rng, dev)
    draws = _Draws(noise, dev)
    approx = KVCacheModel(draft_m, temperature, top_k, top_p, max_seq=cap, noise=noise, full_history=False)
    target = KVCacheModel(target_m, temperature, top_k, top_p, max_seq=cap, noise=noise, full_history=False)
    beams = _beam_of(approx, nb, cap)
    acc_len, acc_rate, num_beams_list, expect_cnt_list = [], [], [], []
    approx_time = target_time = sample_time = compute_expect_time = 0
    target_call_times = approx_call_times = 0
    output_prefix = prefix
    try:
        while output_prefix.shape[1] < T:
            prefix_len = output_prefix.shape[1]
            tt = process_time_ns()
            all_seq, all_beam_idx, all_next_token, all_score, all_prob, all_input_idx = beams.beam_sample(
                output_prefix, gamma, top_k, top_p, padding_input_cnt, draws)
            inc_len = len(all_next_token)
            approx_call_times += 1
            approx_time += process_time_ns() - tt

            tt = process_time_ns()
            out_seq, extra_att_mask, pos, position_ids = get_seq_att_mask(
                extra_sample_cnt, [t.cpu() for t in all_input_idx[1:]], [t.cpu() for t in all_beam_idx],
                [t.cpu() for t in all_next_token], prefix_len, pad_token_id)
            p = target.forward_tree_attention(out_seq, output_prefix[:extra_sample_cnt], extra_att_mask, position_ids, pos)
            target_call_times += 1
            vocab_size = p.size(-1)
            target_time += process_time_ns() - tt

            tt = process_time_ns()
            cur_valid_beam = torch.zeros(nb, dtype=torch.bool, device=dev)       # (:180-186) only the first input is live
            cur_valid_beam[:extra_sample_cnt] = True
            beam_scores = torch.zeros_like(all_score[0])
            n = prefix_len - 1
            max_l = 0
            start = 0
            for i in range(inc_len):
                end = start + (extra_sample_cnt if i == 0 else nb)
                cur_beam_idx = all_beam_idx[i]
                q_scores = all_score[i]
                q_prob = all_prob[i]
                shift = torch.cumsum(cur_valid_beam.long(), dim=0) - 1
                shift_beam_idx = shift[cur_beam_idx]
                cur_p = p[start:start + nb] if i == 0 else p[start:end]
                cur_p = cur_p[cur_valid_beam]
                from_valid_beam = cur_valid_beam[cur_beam_idx]
                p_next_token_scores = beam_scores[cur_valid_beam][:, None].expand_as(cur_p) + cur_p.log()
                p_next_token_scores = norm_logits(p_next_token_scores.reshape(1, -1), temperature, top_k, top_p).view(-1)
                cur_p_prob = p_next_token_scores
                q_prob = q_prob.view(nb, -1)[cur_valid_beam].reshape(-1)
                shift_beam_idx = torch.clamp(shift_beam_idx, min=0)
                cur_sample_idx = shift_beam_idx * vocab_size + all_next_token[i]
                ttt = process_time_ns()
                p_width, e_width = get_num_acc_prob(p_next_token_scores, q_prob, nb)
                compute_expect_time += process_time_ns() - ttt
                if expect_thres < 0:
                    expect_cnt = int(math.floor(float(e_width)))
                else:
                    expect_cnt = get_expect_cnt_by_thres(p_width, expect_thres)
                expect_cnt = max(expect_cnt, min_num_beams)
                expect_cnt_list.append(expect_cnt)
                accept = from_valid_beam.clone()
                acc_cnt = 0
                for j in range(nb):                                               # (:286-311) one beam at a time
                    p_score = cur_p_prob[cur_sample_idx[j]]
                    r = draws.uniform()
                    if acc_cnt >= expect_cnt:
                        accept[j] = False
                        continue
                    if bool(accept[j]):
                        accept[j] = bool(float(p_score / (q_scores[j] + 1e-6)) > r)
                    if not bool(accept[j]):
                        cur_p_prob = max_fn(cur_p_prob - q_prob)
                    else:
                        cur_p_prob = p_next_token_scores
                        acc_cnt += 1
                acc_rate.append(float(accept.float().mean()))
                if acc_cnt >= expect_cnt:
                    assert acc_cnt == expect_cnt
                    num_beams_list.append(acc_cnt)
                    cur_valid_beam = accept
                    p_scores = torch.gather(p_next_token_scores, dim=0, index=cur_sample_idx)
                    p_scores[torch.logical_not(accept)] = 0
                    beam_scores = p_scores.log()
                    n += 1
                    max_l += 1
                    start = end
                else:
                    num_beams_list.append(extra_sample_cnt)
                    break

            end = start + nb
            acc_len.append(max_l)
            all_accepted = max_l == inc_len
            if all_accepted:                                                      # (:343-399)
                cur_p = p[start:end][cur_valid_beam]
                p_next_token_scores = beam_scores[cur_valid_beam][:, None].expand_as(cur_p) + cur_p.log()
                p_next_token_scores = norm_logits(p_next_token_scores.reshape(1, -1), temperature, top_k, top_p).view(-1)
                t = _sample_n(p_next_token_scores, extra_sample_cnt, draws)
            else:                                                                 # (:403-478) the residual of the failed level
                t = _sample_n(cur_p_prob, extra_sample_cnt, draws)
            beam_idx = torch.div(t, vocab_size, rounding_mode="floor").long()
            token = (t % vocab_size)[:, None]
            beam_scores = p_next_token_scores[t].log().view(-1)
            choice = int(cur_valid_beam.nonzero()[beam_idx].reshape(-1)[0])
            src = all_seq[0] if start == 0 else all_seq[(start + padding_input_cnt) // nb]
            output_prefix = src[choice, :n + 1][None, :]
            if int(pos[:, 1].max()) > inc_len:                                    # undo forward_tree_attention's in-place shift
                pos[:, 1] -= prefix_len
            acc_pos = pos[start + choice][None, :]
            accepted_input_idx = acc_pos[:, 0][:extra_sample_cnt]
            accepted_mask = extra_att_mask[acc_pos[:, 0], acc_pos[:, 1]][:extra_sample_cnt].clone()
            if not all_accepted and int(pos[:, 1].min()) == -1:                   # (:463-464) always true there: the tree rows
                accepted_mask[:, prefix_len:] = False                             #  are dropped and re-fed by the next verify
            output_prefix = torch.cat([output_prefix, token.to(output_prefix.dtype)], dim=1)
            target.rollback_tree_attention(accepted_input_idx, accepted_mask)

            if all_accepted:                                                      # (:484-488)
                beams.beam_rollback(max_l, int(all_beam_idx[-1][choice % nb]))
            else:
                beams.beam_rollback(max_l, choice % nb)

            mask = (output_prefix == eos_token_id)                                # (:494-522)
            end_cnt = 0
            for i in range(mask.size(0)):
                if int(mask[i].int().sum()) > ori_eos_cnt:
                    end_cnt = 1000
                    row_mask = torch.cumsum(mask[i].float(), dim=0) < ori_eos_cnt + 1
                    e = int(row_mask.int().sum())
                    if e < mask.size(1):
                        row_mask[e] = True
                    output_prefix = output_prefix[i][row_mask].view(1, -1)
                    break
            if end_cnt >= mask.size(0):
                break
            sample_time += process_time_ns() - tt
    except Exception as e:                                                        # (:528-530)
        print(e)
        raise RuntimeError("")
    output_prefix = output_prefix[0][None, :]
    if debug_dict is not None:                                                    # (the reference's debug_dict is unused; here: the caches)
        debug_dict["approx_cache"], debug_dict["target_cache"] = approx, target
    if verbose:
        print("approx model time", approx_time / 1e9)
        print("target model time", target_time / 1e9)
        print("other time", sample_time / 1e9)
        print("acc len", np.mean(acc_len), len(acc_len), acc_len)
    if details:
        return output_prefix, {
            "approx_time": approx_time, "target_time": target_time, "other_time": sample_time, "acc_len": acc_len,
            "acc_rate": np.mean(acc_rate), "target_call_times": target_call_times, "approx_call_times": approx_call_times,
            "num_beams_list": num_beams_list, "target_model_time": target.forward_time_dict["_model_time"],
            "target_pre_cache_time": target.forward_time_dict["prepare_cache_time"],
            "target_post_prob_time": target.forward_time_dict["norm_prob_time"],
            "compute_expect_time": compute_expect_time, "expect_cnt_list": expect_cnt_list}
    return output_prefix
