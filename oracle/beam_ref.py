"""Oracle for the DRAFT side and the driver loop of the tree-attention beam variant.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  **PARITY UNPINNED**: the reference builds its drafts inside
``KVCacheModel.beam_sample`` around ``transformers.BeamSearchScorer`` / ``GenerationMixin`` internals of transformers 4.35.2
(reference sampling/kvcache_model.py:8, 439-567, 571-1025).  This image ships transformers 5.15, where those names are
gone, so the reference's beam path cannot be run here and nothing could be recorded from it; the library is not stood in
for.  What this file restates is the part of that code whose results the driver actually consumes - with
``optimization=False`` (the only value the driver passes, speculative_sampling.py:96) ``BeamSearchScorer.process`` is never
called, ``is_done`` stays False and the ``finalize`` output (``ret[0]``) is never read (speculative_sampling.py:98) - i.e.
the per-step intermediate results.  The target side it drives (``forward_tree_attention`` / ``rollback_tree_attention``,
``get_seq_att_mask``, ``get_num_acc_prob``) IS pinned (G9, tests/golden/g9_tree.*).

Restated, each with the reference lines it follows:
  hf_top_k / hf_top_p            transformers 4.35.2 TopKLogitsWarper / TopPLogitsWarper.__call__ (used at kvcache_model.py:497-500)
  sample_n                       sampling/utils.py:213-233 with num_samples >= 1 (torch.multinomial without replacement ==
                                 topk(p / Exp(1)); checked against torch in tests/test_oracle_golden.py)
  RefBeamKVCacheModel            kvcache_model.py:439-567 (beam_sample_with_kv_cache), :571-1025 (beam_sample, decoder-only,
                                 return_intermediate_results=True, optimization=False), :312-324 (beam_rollback),
                                 :359-436 (rollback with a tensor choice and end_pos None)
  beam_speculative_sampling_v2   speculative_sampling.py:18-581, extra_sample_cnt == 1 (one input sequence per tree
                                 verify; the harness also sweeps 2, evaluation.py:862 - not restated)
"""
from __future__ import annotations

import math
from time import process_time_ns

import numpy as np
import torch

from .kvcache_ref import RefKVCacheModel
from .noise import TorchGlobalNoise
from .sampling_ref import max_fn, norm_logits
from .tree_ref import get_expect_cnt_by_thres, get_num_acc_prob, get_seq_att_mask


def hf_top_k(scores: torch.Tensor, top_k: int) -> torch.Tensor:
    """TopKLogitsWarper: everything below the k-th largest score of its row becomes -inf."""
    k = min(int(top_k), scores.size(-1))
    return scores.masked_fill(scores < torch.topk(scores, k)[0][..., -1, None], -float("inf"))


def hf_top_p(scores: torch.Tensor, top_p: float) -> torch.Tensor:
    """TopPLogitsWarper (min_tokens_to_keep = 1): ascending sort, tokens whose cumulative probability stays <= 1 - top_p
    are removed, the largest one is always kept."""
    srt, idx = torch.sort(scores, descending=False)
    cum = srt.softmax(dim=-1).cumsum(dim=-1)
    rm = cum <= (1 - top_p)
    rm[..., -1:] = False
    return scores.masked_fill(rm.scatter(1, idx, rm), -float("inf"))


def with_replacement_draws(nz_idx, nz_val, num_samples: int, uniform64) -> list:
    """ATen's CPU multinomial WITH replacement (MultinomialKernel.cpp) for one distribution, given only its non-zero
    entries (index order): the cumulative distribution is accumulated sequentially in fp32 - zeros add nothing, so the
    non-zero entries alone give the same prefixes -, divided by its last value, and every sample is the first slot whose
    prefix is >= one double uniform.  Checked against torch.multinomial(..., replacement=True) in the tests."""
    pre = np.cumsum(np.asarray(nz_val, dtype=np.float32), dtype=np.float32)
    cum = (pre / pre[-1]).astype(np.float64)
    out = []
    for _ in range(num_samples):
        u = float(uniform64())
        out.append(int(nz_idx[min(int(np.searchsorted(cum, u, side="left")), len(nz_idx) - 1)]))
    return out


def sample_n(probs: torch.Tensor, num_samples: int, noise) -> torch.Tensor:
    """utils.py:213-233.  torch.multinomial(p, n, replacement=False) on CPU is topk(p / q, n) with q ~ Exp(1) drawn in ONE
    call over p's shape (ATen multinomial: the same exponential trick as the single draw, argmax replaced by topk); with
    fewer non-zero entries than samples the reference draws WITH replacement (:214-215): ATen's inverse-CDF sampler on one
    double uniform per sample (with_replacement_draws).  A draw that landed on a (near-)zero entry is replaced by the mode
    of the flattened tensor (:228-230)."""
    if not bool((probs.max() < float("inf")) & (probs.min() >= 0)) or bool((probs.sum(-1) == 0).any()):
        raise RuntimeError("prob error")
    if torch.numel(probs.nonzero()) < num_samples:
        flat = probs.reshape(-1)
        nz = flat.nonzero().reshape(-1)
        draws = with_replacement_draws(nz.tolist(), flat[nz].float().tolist(), num_samples, noise.uniform64)
        idx = torch.tensor(draws, dtype=torch.long).reshape(probs.shape[:-1] + (num_samples,))
    else:
        q = noise.exponential(probs)
        idx = torch.topk(probs / q, num_samples, dim=-1).indices
    mask = torch.gather(probs, -1, idx) < 1e-9
    if bool(mask.any()):
        idx[mask] = torch.argmax(probs).item()
    return idx


class RefBeamKVCacheModel(RefKVCacheModel):
    """RefKVCacheModel + the beam-sampling draft of kvcache_model.py:439-1025 (decoder-only)."""

    @torch.no_grad()
    def rollback(self, end_pos, choice=None):
        """kvcache_model.py:359-436 incl. the tensor-choice branch (:393-395, :435-436); end_pos None keeps every position.
        A 0-dim choice (the driver's `.squeeze()` of one index) drops the batch dimension, as tensor indexing does there."""
        if choice is None or isinstance(choice, int):
            return super().rollback(end_pos, choice)
        choice = choice.cpu()
        self._past_key_values = [(k[choice, :, :end_pos, :], v[choice, :, :end_pos, :]) for k, v in self._past_key_values]
        if self._prob_history is not None:
            self._prob_history = self._prob_history[choice, :end_pos, :]

    @torch.no_grad()
    def beam_rollback(self, beam_idx: int, choice):
        """kvcache_model.py:312-324: the cache as it was after the forward of beam step `beam_idx` (the last one when every
        level was accepted), then only the beams in `choice`."""
        assert beam_idx >= 0
        snaps = self.beam_past_key_values
        self._past_key_values = snaps[beam_idx - 1] if beam_idx == len(snaps) else snaps[beam_idx]
        self.rollback(None, choice)

    @torch.no_grad()
    def beam_sample_with_kv_cache(self, prefix, gamma, num_beams, top_k=None, top_p=None, padding_input_cnt=0):
        """kvcache_model.py:439-567 -> :571-1025 with return_intermediate_results=True, ret_seq_scores=True,
        optimization=False: `gamma` steps of beam SAMPLING over `num_beams` beams.  Returns what the driver unpacks from
        ret[1:] (:98): all_seq (the beams' token rows before every step + after the last), all_beam_idx, all_next_token,
        all_score (the joint probability of each drawn (beam, token)), all_prob (the joint distribution of each step) and
        all_input_idx."""
        nb = int(num_beams)
        assert prefix.size(0) == 1, "one input sequence (extra_sample_cnt == 1)"
        input_ids = prefix.repeat_interleave(nb, dim=0)                     # _expand_inputs_for_generation (:512-517)
        if self._past_key_values is not None:                                # (:519-525) one cached row -> num_beams rows
            self._past_key_values = [tuple(val.repeat(nb, 1, 1, 1) for val in kv) for kv in self._past_key_values]
        self.beam_past_key_values = []
        beam_scores = torch.zeros((1, nb), dtype=torch.float)
        if padding_input_cnt > 0:
            beam_scores[:, -padding_input_cnt:] = float("-inf")             # (:649-650) only the first beams are live
        beam_scores = beam_scores.view(nb)
        all_seq, all_beam_idx, all_next_token, all_score, all_prob = [], [], [], [], []
        input_index = torch.arange(nb, dtype=torch.long)
        all_input_idx = [input_index]
        new_len = 0
        while True:
            if self._past_key_values is None:                                # (:676-700) prefill of the num_beams copies
                out = self._model(input_ids)
            else:                                                            # (:704-745) the uncached tail only
                cached = self._past_key_values[0][0].shape[2]
                out = self._model(input_ids[:, cached:], past_key_values=self._past_key_values, use_cache=True)
            self.beam_past_key_values.append(out.past_key_values)            # (:768) BEFORE the reorder below
            logits = out.logits[:, -1, :]
            scores = torch.nn.functional.log_softmax(logits, dim=-1)         # (:778-780)
            nts = scores + beam_scores[:, None].expand_as(scores)            # (:783-786; the processor list is empty)
            if top_k is not None and top_k > 0:                              # (:497-500) the warpers, per beam row
                nts = hf_top_k(nts, top_k)
            if top_p is not None and top_p > 0:
                nts = hf_top_p(nts, top_p)
            V = nts.shape[-1]
            nts = nts.view(1, nb * V)                                        # (:808-811) ONE distribution over (beam, token)
            probs = torch.nn.functional.softmax(nts, dim=-1)
            next_tokens = sample_n(probs, nb, self.noise)                    # (:836)
            nts_g = torch.gather(nts, -1, next_tokens)
            next_indices = torch.div(next_tokens, V, rounding_mode="floor")  # (:872-877)
            next_tokens = next_tokens % V
            nts_g = torch.clamp(nts_g, min=-1e10)
            beam_scores = nts_g.squeeze()
            beam_next_tokens = next_tokens.squeeze()
            beam_idx = next_indices.squeeze()
            all_seq.append(input_ids)                                        # (:884-894)
            all_beam_idx.append(beam_idx)
            all_next_token.append(beam_next_tokens)
            sample_id = (beam_idx * V + beam_next_tokens).view(1, -1)
            all_score.append(torch.gather(probs, -1, sample_id).view(-1))
            all_prob.append(probs)
            input_index = input_index[beam_idx]
            all_input_idx.append(input_index)
            input_ids = torch.cat([input_ids[beam_idx, :], beam_next_tokens.unsqueeze(-1)], dim=-1)        # (:896)
            self._past_key_values = [tuple(t.index_select(0, beam_idx) for t in kv) for kv in out.past_key_values]   # (:901-904)
            new_len += 1
            if new_len == gamma:                                             # (:929; the scorer is never done, no criteria)
                break
        self.beam_rollback_flag = False
        all_seq.append(input_ids)                                            # (:981)
        return all_seq, all_beam_idx, all_next_token, all_score, all_prob, all_input_idx


@torch.no_grad()
def beam_speculative_sampling_v2(prefix, approx_model, target_model, eos_token_id, pad_token_id, max_len: int,
                                 gamma: int = 4, width: int = 8, num_beams: int = 8, min_num_beams: int = 1,
                                 extra_sample_cnt: int = -1, expect_thres: float = 0.7, temperature: float = 1,
                                 top_k: int = 0, top_p: float = 0, verbose: bool = False, random_seed: int = None,
                                 details: bool = False, debug_dict=None, noise=None):
    """speculative_sampling.py:18-581 for extra_sample_cnt == 1 (`random_seed` is accepted and unused, as there)."""
    noise = noise or TorchGlobalNoise()
    if extra_sample_cnt == -1:
        extra_sample_cnt = num_beams
    if extra_sample_cnt != 1:
        raise NotImplementedError("only extra_sample_cnt == 1 (one input sequence per tree verify) is restated")
    padding_input_cnt = num_beams - extra_sample_cnt
    if pad_token_id is None:
        pad_token_id = eos_token_id
    seq_len = prefix.shape[1]
    ori_eos_cnt = int((prefix == eos_token_id).int().sum())
    T = seq_len + max_len
    acc_len, acc_rate, num_beams_list, expect_cnt_list = [], [], [], []
    approx = RefBeamKVCacheModel(approx_model, temperature, top_k, top_p, noise=noise)
    target = RefKVCacheModel(target_model, temperature, top_k, top_p, noise=noise)
    assert prefix.shape[0] == 1, "input batch size must be 1"
    approx_time = target_time = sample_time = compute_expect_time = 0
    target_call_times = approx_call_times = 0
    output_prefix = prefix
    nb = num_beams
    try:
        while output_prefix.shape[1] < T:
            prefix_len = output_prefix.shape[1]
            tt = process_time_ns()
            all_seq, all_beam_idx, all_next_token, all_score, all_prob, all_input_idx = approx.beam_sample_with_kv_cache(
                output_prefix, gamma=gamma, num_beams=nb, top_k=top_k, top_p=top_p, padding_input_cnt=padding_input_cnt)
            inc_len = len(all_next_token)
            approx_call_times += 1
            approx_time += process_time_ns() - tt

            tt = process_time_ns()
            out_seq, extra_att_mask, pos, position_ids = get_seq_att_mask(extra_sample_cnt, all_input_idx[1:], all_beam_idx,
                                                                          all_next_token, prefix_len, pad_token_id)
            p = target.forward_tree_attention(out_seq, output_prefix[:extra_sample_cnt], extra_att_mask, position_ids, pos)
            target_call_times += 1
            vocab_size = p.size(-1)
            target_time += process_time_ns() - tt

            tt = process_time_ns()
            cur_valid_beam = torch.zeros_like(all_beam_idx[0])               # (:180-186) only the first input is live
            cur_valid_beam[:extra_sample_cnt] = 1
            cur_valid_beam = cur_valid_beam.bool()
            beam_scores = torch.zeros_like(all_score[0])
            n = prefix_len - 1
            max_l = 0
            start = 0
            for i in range(inc_len):
                end = start + (extra_sample_cnt if i == 0 else nb)
                cur_beam_idx = all_beam_idx[i]
                q_scores = all_score[i]
                q_prob = all_prob[i]
                shift = torch.cumsum(cur_valid_beam.long(), dim=0) - 1      # (:218-219)
                shift_beam_idx = shift[cur_beam_idx]
                cur_p = p[start:start + nb] if i == 0 else p[start:end]     # (:223-227)
                cur_p = cur_p[cur_valid_beam]
                from_valid_beam = cur_valid_beam[cur_beam_idx]
                p_next_token_scores = beam_scores[cur_valid_beam][:, None].expand_as(cur_p) + cur_p.log()
                p_next_token_scores = norm_logits(p_next_token_scores.view(1, -1), temperature, top_k, top_p).view(-1)
                cur_p_prob = p_next_token_scores
                q_prob = q_prob.view(nb, -1)[cur_valid_beam].view(-1)
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
                for j in range(nb):                                           # (:286-311) one beam at a time
                    p_score = cur_p_prob[cur_sample_idx[j]]
                    r = noise.uniform()
                    if acc_cnt >= expect_cnt:
                        accept[j] = False
                        continue
                    if bool(accept[j]):
                        accept[j] = bool(((p_score / (q_scores[j] + 1e-6)) > r).item())
                    if not bool(accept[j]):
                        cur_p_prob = max_fn(cur_p_prob - q_prob)
                    else:
                        cur_p_prob = p_next_token_scores
                        acc_cnt += 1
                acc_rate.append(accept.float().mean().item())
                if acc_cnt >= expect_cnt:                                     # (:319-334)
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
            if max_l == inc_len:                                              # (:343-399) every level accepted
                cur_p = p[start:end][cur_valid_beam]
                p_next_token_scores = beam_scores[cur_valid_beam][:, None].expand_as(cur_p) + cur_p.log()
                p_next_token_scores = norm_logits(p_next_token_scores.view(1, -1), temperature, top_k, top_p).squeeze()
                t = sample_n(p_next_token_scores, extra_sample_cnt, noise)
                beam_idx = torch.div(t, vocab_size, rounding_mode="floor").long()
                token = (t % vocab_size)[:, None]
                beam_scores = p_next_token_scores[t].log().view(-1)
                choice = cur_valid_beam.nonzero()[beam_idx].squeeze()
                src = all_seq[0] if start == 0 else all_seq[(start + padding_input_cnt) // nb]
                output_prefix = src[choice, :n + 1]
                if pos[:, 1].max() > inc_len:                                 # undo forward_tree_attention's in-place shift
                    pos[:, 1] -= prefix_len
                acc_pos = pos[start + choice]
                output_prefix = output_prefix[None, :]
                acc_pos = acc_pos[None, :]
                output_prefix = torch.cat([output_prefix, token], dim=1)
                accepted_input_idx = acc_pos[:, 0][:extra_sample_cnt]
                accepted_mask = extra_att_mask[acc_pos[:, 0], acc_pos[:, 1]][:extra_sample_cnt]
                target.rollback_tree_attention(accepted_input_idx, accepted_mask)
            else:                                                             # (:403-478) a level was rejected
                t = sample_n(cur_p_prob, extra_sample_cnt, noise)
                beam_idx = torch.div(t, vocab_size, rounding_mode="floor").long()
                token = (t % vocab_size)[:, None]
                choice = cur_valid_beam.nonzero()[beam_idx].squeeze()
                src = all_seq[0] if start == 0 else all_seq[(start + padding_input_cnt) // nb]
                output_prefix = src[choice, :n + 1]
                if pos[:, 1].max() > inc_len:
                    pos[:, 1] -= prefix_len
                acc_pos = pos[start + choice]
                output_prefix = output_prefix[None, :]
                acc_pos = acc_pos[None, :]
                accepted_input_idx = acc_pos[:, 0]
                accepted_mask = extra_att_mask[acc_pos[:, 0], acc_pos[:, 1]]
                if pos[:, 1].min() == -1:                                     # (:463-464) always true: the tree rows are dropped,
                    accepted_mask[:, prefix_len:] = False                     #  the accepted tokens are re-fed next time
                beam_scores = p_next_token_scores[t].log().view(-1)
                accepted_input_idx = accepted_input_idx[:extra_sample_cnt]
                accepted_mask = accepted_mask[:extra_sample_cnt]
                output_prefix = torch.cat([output_prefix, token], dim=1)
                target.rollback_tree_attention(accepted_input_idx, accepted_mask)

            if max_l == inc_len:                                              # (:484-488)
                last_beam_idx = all_beam_idx[-1]
                approx.beam_rollback(max_l, last_beam_idx[choice % nb])
            else:
                approx.beam_rollback(max_l, choice % nb)

            mask = (output_prefix == eos_token_id)                            # (:494-522) cut after the first new EOS
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
    except Exception as e:                                                    # (:528-530)
        print(e)
        raise RuntimeError("")
    output_prefix = output_prefix[0][None, :]
    if debug_dict is not None:                                                    # (the reference's debug_dict is unused; here: the caches)
        debug_dict["approx_cache"], debug_dict["target_cache"] = approx, target
    if details:
        return output_prefix, {
            "approx_time": approx_time, "target_time": target_time, "other_time": sample_time, "acc_len": acc_len,
            "acc_rate": np.mean(acc_rate), "target_call_times": target_call_times, "approx_call_times": approx_call_times,
            "num_beams_list": num_beams_list, "target_model_time": target.forward_time_dict["_model_time"],
            "target_pre_cache_time": target.forward_time_dict["prepare_cache_time"],
            "target_post_prob_time": target.forward_time_dict["norm_prob_time"],
            "compute_expect_time": compute_expect_time, "expect_cnt_list": expect_cnt_list}
    return output_prefix
