#!/usr/bin/env python3
"""Offline counterpart of the reference's evaluation.py driver: the same three timed
loops over a prompt set - target-only autoregressive sampling, speculative sampling, and the width-w i.i.d. variant -
with the reference's accounting (process_time_ns around each call, tokens = len(output) - len(prompt), get_score per
output, power integrated from a poller process) and its log lines, around this package's ``sampling`` functions.

What differs, because nothing can be fetched here: models come from a local HF checkpoint directory when the name is
a path that exists, else random-init weights of the named local config (``opt-125m``, ``llama-68m`` ...); the tokenizer
is a local directory (``--tokenizer``) or a byte-level stand-in; ``--dataset chatalpaca`` reads ``--data-path``
(chatalpaca-10k.json) and falls back to synthetic prompts with lengths ~ U{32..512}.  ROUGE / exact-match scoring
(reference utils.py:8-93, hf ``evaluate``) is dataset scoring, not decode, and is not reproduced.

    python tools/evaluate_offline.py --approx_model_name llama-68m --target_model_name llama-2-13b --max_tokens 128 \
        --dataset chatalpaca [--data-path chatalpaca-10k.json --tokenizer /path/to/tokenizer] [--n-prompts 100]
"""
import argparse
import os
import sys
import time
from time import process_time_ns

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from llmspeculativesampling_amd import harness  # noqa: E402
from llmspeculativesampling_amd.config import load_config  # noqa: E402
from llmspeculativesampling_amd.engine import SpecDecModel  # noqa: E402
from llmspeculativesampling_amd.quality import get_score  # noqa: E402
from llmspeculativesampling_amd.sampling import (autoregressive_sampling, multi_speculative_sampling,  # noqa: E402
                                                 speculative_sampling)


def parse_arguments():
    p = argparse.ArgumentParser(description="offline evaluation driver")
    p.add_argument("--approx_model_name", type=str, default="facebook/opt-125m")
    p.add_argument("--target_model_name", type=str, default="facebook/opt-350m")
    p.add_argument("--seed", "-s", type=int, default=None)
    p.add_argument("--max_tokens", "-M", type=int, default=20)
    p.add_argument("--gamma", "-g", type=int, default=4)
    p.add_argument("--width", "-w", type=int, default=2)
    p.add_argument("--log_file", type=str, default="logs/log.txt")
    p.add_argument("--dataset", type=str, default="chatalpaca")
    p.add_argument("--max_seconds", type=int, default=7200)
    p.add_argument("--data-path", type=str, default="chatalpaca-10k.json")
    p.add_argument("--tokenizer", type=str, default=None, help="local HF tokenizer directory")
    p.add_argument("--n-prompts", type=int, default=100)
    p.add_argument("--repeats", type=int, default=2)
    p.add_argument("--dtype", default="bfloat16", choices=["bfloat16", "float32"])
    p.add_argument("--rng", default="host", choices=["host", "device"])
    p.add_argument("--skip", default="", help="comma list of loops to skip: ar,ss,iid")
    return p.parse_args()


def build_model(name: str, dtype, seed: int) -> SpecDecModel:
    if os.path.isdir(name):                                          # local checkpoint, never the hub
        from transformers import AutoModelForCausalLM
        hf = AutoModelForCausalLM.from_pretrained(name, local_files_only=True, torch_dtype=dtype)
        return SpecDecModel.from_hf(hf, dtype=dtype)
    return SpecDecModel.synthetic(load_config(os.path.basename(name)), seed=seed, dtype=dtype)


def main():
    args = parse_arguments()
    dtype = getattr(torch, args.dtype)
    small = build_model(args.approx_model_name, dtype, 0)
    large = build_model(args.target_model_name, dtype, 1)
    V = large.cfg.vocab_size
    tok = harness.load_tokenizer(args.tokenizer, V)
    top_k, top_p = 20, 0.9                                           # evaluation.py:254-255
    if args.dataset == "chatalpaca" and os.path.exists(args.data_path):
        prompts, _answers = harness.read_chatalpaca(args.data_path)
        ds = [tok.encode(s, return_tensors="pt") for s in prompts]
        source = args.data_path
    else:
        ds = harness.synthetic_prompts(args.n_prompts, V)
        source = "synthetic prompts, lengths ~ U{32..512}, seed 5"
    room = large.max_pos - args.max_tokens - args.gamma - 2
    ds = [d for d in ds if d.size(-1) <= room][:args.n_prompts]
    os.makedirs(os.path.dirname(args.log_file) or ".", exist_ok=True)
    log_f = open(args.log_file, "a")

    def emit(lines):
        for ln in lines:
            print(ln)
            print(ln, file=log_f)
        log_f.flush()

    emit([f"{args.approx_model_name} -> {args.target_model_name}, dataset {args.dataset} ({source}), "
          f"max_tokens {args.max_tokens}, gamma {args.gamma}, width {args.width}, dtype {args.dtype}, rng {args.rng}"])
    skip = set(args.skip.split(","))
    for rep in range(args.repeats):
        print(f"input length 0-100000, {len(ds)} data in total")
        print("total_input_tokens", sum(d.size(1) for d in ds))
        if args.seed is not None:
            torch.manual_seed(args.seed)

        # ---- target-only baseline (evaluation.py:421-480)
        if "ar" not in skip:
            total_ns = tokens = 0
            scores = []
            with harness.PowerMonitor() as pm:
                for n_done, ids in enumerate(ds, 1):
                    ids = ids.cuda()
                    t = process_time_ns()
                    out = autoregressive_sampling(ids, large, args.max_tokens, eos_token_id=tok.eos_token_id,
                                                  top_k=top_k, top_p=top_p, pad_token_id=tok.pad_token_id,
                                                  rng=args.rng)
                    total_ns += process_time_ns() - t
                    tokens += len(out[0]) - ids.size(1)
                    scores.append(get_score(out, large, ids.size(1)).item())
                    if total_ns / 1e9 > args.max_seconds:
                        emit([f"terminated at {n_done}"])
                        break
            emit(harness.large_model_log_lines(total_ns, tokens, scores, pm.total()))

        # ---- speculative sampling (evaluation.py:515-583) and the iid variant
        loops = []
        if "ss" not in skip:
            loops.append(("google speculative decoding (with KVCache)",
                          lambda ids: speculative_sampling(ids, small, large, eos_token_id=tok.eos_token_id,
                                                           pad_token_id=tok.pad_token_id, max_len=args.max_tokens,
                                                           gamma=args.gamma, top_k=top_k, top_p=top_p,
                                                           random_seed=args.seed, details=True, rng=args.rng)))
        if "iid" not in skip:
            loops.append((f"iid multi-draft speculative decoding (gamma {args.gamma}, width {args.width})",
                          lambda ids: multi_speculative_sampling(ids, small, large, eos_token_id=tok.eos_token_id,
                                                                 pad_token_id=tok.pad_token_id, max_len=args.max_tokens,
                                                                 gamma=args.gamma, width=args.width, strategy="iid",
                                                                 top_k=top_k, top_p=top_p, random_seed=args.seed,
                                                                 details=True, rng=args.rng)))
        for title, run in loops:
            total_ns = tokens = 0
            agg = dict(approx_time=0, target_time=0, other_time=0, acc_len_sum=0, acc_rate=[], target_call_times=0,
                       approx_call_times=0, target_model_time=0, target_pre_cache_time=0, target_post_prob_time=0)
            scores = []
            wall0 = time.time()
            with harness.PowerMonitor() as pm:
                for n_done, ids in enumerate(ds, 1):
                    ids = ids.cuda()
                    t = process_time_ns()
                    out, d = run(ids)
                    total_ns += process_time_ns() - t
                    tokens += len(out[0]) - ids.size(1)
                    for k in ("approx_time", "target_time", "other_time", "target_call_times", "approx_call_times"):
                        agg[k] += d[k]
                    for k in ("target_model_time", "target_pre_cache_time", "target_post_prob_time"):   # evaluation.py:540-542
                        agg[k] += d.get(k, 0)
                    agg["acc_len_sum"] += float(np.sum(d["acc_len"]))
                    agg["acc_rate"].append(float(d["acc_rate"]))
                    scores.append(get_score(out, large, ids.size(1)).item())
                    if total_ns / 1e9 > args.max_seconds:
                        emit([f"terminated at {n_done}"])
                        break
            emit(harness.speculative_log_lines(title, total_ns, tokens, agg, scores, pm.total()))
            emit([f"wall time {time.time() - wall0} s (get_score and power polling included)"])
    log_f.close()


if __name__ == "__main__":
    main()
