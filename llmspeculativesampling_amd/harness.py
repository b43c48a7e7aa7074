"""Host-side pieces of the reference's evaluation driver that sit either side of the decode path
(reference evaluation.py).  No model math here: prompt sources, the power-log
integration and the log lines; ``tools/evaluate_offline.py`` strings them around ``sampling.*``.

* ``read_chatalpaca``   chatalpaca-10k JSONL -> one prompt per assistant turn with the cumulative history
                        (evaluation.py:347-364)
* ``synthetic_prompts`` the offline stand-in: lengths ~ U{32..512}, seed 5
* ``ByteTokenizer``     stand-in tokenizer when no local tokenizer directory is given (ids 3.. = bytes, eos 2)
* ``PowerMonitor`` / ``total_power``   the poller process and the sum over [t1, t2] (evaluation.py:135-152, 418,
                        471-475), fed by ``tools/rocm_power_monitor.py`` (rocm-smi instead of nvidia-smi)
* ``*_log_lines``       the result lines in the reference's wording (evaluation.py:479-480, 567-583)
"""
from __future__ import annotations

import json
import os
import subprocess
import sys
import time
from typing import Iterable, List, Optional, Sequence, Tuple

import numpy as np
import torch


def read_chatalpaca(path: str) -> Tuple[List[str], List[str]]:
    """(prompts, answers): every assistant turn yields one prompt made of all earlier turns, each followed by a
    newline, and its own text as the reference answer."""
    prompts, answers = [], []
    with open(path, "r") as f:
        for line in f:
            line = line.strip()
            if not line:
                continue
            history = ""
            for turn in json.loads(line)["conversations"]:
                if turn["from"] != "human":
                    prompts.append(history)
                    answers.append(turn["value"])
                history += turn["value"] + "\n"
    return prompts, answers


def synthetic_prompts(n: int, vocab: int, seed: int = 5, lo: int = 32, hi: int = 512) -> List[torch.Tensor]:
    rng = np.random.default_rng(seed)
    lens = rng.integers(lo, hi + 1, size=n)
    return [torch.from_numpy(rng.integers(3, vocab, size=(1, int(L)))) for L in lens]


class ByteTokenizer:
    """UTF-8 bytes shifted by 3 (0 pad-like, 1 bos, 2 eos): enough to push text through the path offline."""
    eos_token_id = 2
    pad_token_id = None

    def __init__(self, vocab: int):
        assert vocab >= 259
        self.vocab = vocab

    def encode(self, text: str, return_tensors: Optional[str] = None):
        ids = [1] + [3 + b for b in text.encode("utf-8")]
        return torch.tensor([ids], dtype=torch.int64) if return_tensors == "pt" else ids

    def decode(self, ids: Iterable[int], skip_special_tokens: bool = True) -> str:
        bs = bytes(int(i) - 3 for i in ids if 3 <= int(i) < 259)
        return bs.decode("utf-8", errors="replace")


def load_tokenizer(path: Optional[str], vocab: int):
    """A local HF tokenizer directory (never the hub), or the byte stand-in."""
    if path:
        from transformers import AutoTokenizer
        return AutoTokenizer.from_pretrained(path, local_files_only=True)
    return ByteTokenizer(vocab)


class PowerMonitor:
    """Runs the poller as a child process for the duration of one loop and integrates what it printed."""

    def __init__(self, script: Optional[str] = None):
        here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        self.script = script or os.path.join(here, "tools", "rocm_power_monitor.py")
        self.proc = None
        self.t1 = self.t2 = 0.0

    def __enter__(self):
        self.proc = subprocess.Popen([sys.executable, "-u", self.script], text=True, stdout=subprocess.PIPE,
                                     stderr=subprocess.DEVNULL)
        self.t1 = time.time()
        return self

    def __exit__(self, *exc):
        self.t2 = time.time()
        self.proc.terminate()
        try:
            self.lines = self.proc.communicate(timeout=10)[0].splitlines()
        except subprocess.TimeoutExpired:
            self.proc.kill()
            self.lines = self.proc.communicate()[0].splitlines()
        return False

    def total(self) -> float:
        return total_power(self.lines, self.t1, self.t2)


def total_power(lines: Sequence[str], t1: float, t2: float) -> float:
    """Sum of the samples strictly inside (t1, t2), the first of them left out (evaluation.py:135-152); malformed
    (cut-off) lines are skipped."""
    total, first = 0.0, True
    for ln in lines:
        parts = ln.strip().split()
        if len(parts) < 2:
            continue
        try:
            ts, watts = float(parts[0]), float(parts[1])
        except ValueError:
            continue
        if t1 < ts < t2:
            if first:
                first = False
            else:
                total += watts
    return total


def large_model_log_lines(total_ns: int, tokens: int, scores: Sequence[float], power: float) -> List[str]:
    s = total_ns / 1e9
    return [f"\nlarge model total time {s} s, total tokens {tokens}, average time {s / max(tokens, 1)} s/token, "
            f"prob_score = {np.mean(scores)}, prob score cut = {np.mean(scores)}",
            f"total power consumption: {power}", f"power/token: {power / max(tokens, 1)}"]


def speculative_log_lines(title: str, total_ns: int, tokens: int, agg: dict, scores: Sequence[float], power: float
                          ) -> List[str]:
    """agg: sums of the ``details`` fields over the prompts (evaluation.py:533-542)."""
    s = total_ns / 1e9
    calls = max(agg["target_call_times"], 1)
    return [f"\n {title} total time {s} s, total tokens {tokens}, average time {s / max(tokens, 1)} s/token",
            f"approx time {agg['approx_time'] / 1e9}, target time {agg['target_time'] / 1e9}, "
            f"other time {agg['other_time'] / 1e9}",
            f"average accepted len {agg['acc_len_sum'] / calls}, target call times {agg['target_call_times']}, "
            f"acc rate {np.mean(agg['acc_rate']) if agg['acc_rate'] else 0.0}, "
            f"approx call times {agg['approx_call_times']}",
            f"prob score = {np.mean(scores)}, prob score cut = {np.mean(scores)}",
            f"total power consumption: {power}", f"power/token: {power / max(tokens, 1)}",
            # evaluation.py:581-582 (the wrapper's forward_time_dict summed over the prompts, kvcache_model.py:163-250)
            f"target_model_time: {agg.get('target_model_time', 0) / 1e9}, "
            f"pre cache time: {agg.get('target_pre_cache_time', 0) / 1e9}, "
            f"post prob time: {agg.get('target_post_prob_time', 0) / 1e9}"]
