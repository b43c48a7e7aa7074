"""Oracle for the host-side helpers of the tree-attention variant (test infrastructure, see oracle/__init__.py):
reference sampling/utils.py:95-148 (get_seq_att_mask) and :247-350 (the acceptance-count recursion)."""
from __future__ import annotations

import torch


def get_seq_att_mask(input_cnt, all_input_idx, all_beam_idx, all_next_token, input_len, pad_token_id):
    """utils.py:95-148.  Level by level, beam j of a level appends token all_next_token[l][j] to input sequence
    all_input_idx[l][j]; its tree mask is its parent beam's mask (all_beam_idx[l][j], from the previous level) padded
    with False up to its own slot, plus True for itself.  Returns (tokens, full mask incl. the all-True prefix part,
    pos = [[input, slot]] with `input_cnt` leading [i, -1] entries, position ids)."""
    seqs = [[] for _ in range(input_cnt)]
    masks = [[] for _ in range(input_cnt)]
    pids = [[] for _ in range(input_cnt)]
    prev = [[] for _ in range(all_input_idx[0].numel())]
    pos = [[i, -1] for i in range(input_cnt)]
    depth = input_len
    for inp_l, beam_l, tok_l in zip(all_input_idx, all_beam_idx, all_next_token):
        cur = []
        for j in range(inp_l.numel()):
            i, tok, b = int(inp_l[j]), int(tok_l[j]), int(beam_l[j])
            slot = len(seqs[i])
            pos.append([i, slot])
            seqs[i].append(tok)
            pids[i].append(depth)
            m = prev[b] + [False] * (slot - len(prev[b])) + [True]
            masks[i].append(m)
            cur.append(m)
        prev = cur
        depth += 1
    n = max(len(s) for s in seqs)
    for i in range(input_cnt):
        pids[i] += [0] * (n - len(seqs[i]))
        seqs[i] += [pad_token_id] * (n - len(seqs[i]))
        masks[i] = [r + [False] * (n - len(r)) for r in masks[i]] + [[False] * n for _ in range(n - len(masks[i]))]
    full = torch.ones(input_cnt, n, n + input_len, dtype=torch.bool)
    full[:, :, input_len:] = torch.tensor(masks, dtype=torch.float32).bool()
    return torch.tensor(seqs, dtype=torch.long), full, torch.tensor(pos, dtype=torch.long), torch.tensor(pids, dtype=torch.long)


def get_num_acc_prob(p, q, m):
    """utils.py:247-337: distribution of the number of accepted drafts among m i.i.d. draws from q verified against p
    with residual updates.  alpha_i = sum(min(1, p_i / (q + 1e-6)) * q) for p_0 = p, p_{i+1} = norm(max(p_i - q, 0));
    P(m, k) = sum_i alpha_i * prod_{j<i}(1 - alpha_j) * P(m - i, k - 1) (the recursion restarts at p_0, as there).
    Returns (prob, expect) with the reference's index quirk: prob[k - 1] = P(m, k), so prob[m] holds P(m, 0)."""
    alphas = []
    cur = p.clone()
    for _ in range(m):
        r = cur / (q + 1e-6)
        alphas.append(torch.sum(torch.where(r > 1, torch.ones_like(r), r) * q))
        d = cur - q
        d = torch.where(d < 0, torch.zeros_like(d), d)
        cur = d / (d.sum() + 1e-6)
    memo = {}

    def first(i):                                   # first accepted draft is number i (1-based)
        pr = 1.0
        for j in range(i - 1):
            pr = pr * (1 - alphas[j])
        return pr * alphas[i - 1]

    def P(mm, k):
        if mm < k:
            return 0
        if mm == 0 and k == 0:
            return 1
        if (mm, k) in memo:
            return memo[(mm, k)]
        if k == 0:
            pr = 1.0
            for j in range(mm):
                pr = pr * (1 - alphas[j])
            return pr
        tot = 0
        for i in range(1, mm + 1):
            tot = tot + first(i) * P(mm - i, k - 1)
        memo[(mm, k)] = tot
        return tot
    prob = torch.zeros(m + 1)
    expect = 0.0
    for k in range(m + 1):
        v = P(m, k)
        prob[k - 1] = v
        expect = expect + v * k
    return prob, expect


def get_expect_cnt_by_thres(p_width, expect_thres):
    """utils.py:339-350: largest n whose tail mass sum_{j >= n} p_width[j] reaches the threshold."""
    n, cum = p_width.numel(), 0
    while cum < expect_thres and n > 0:
        n -= 1
        cum += p_width[n]
    return int(n)
