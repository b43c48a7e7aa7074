#!/usr/bin/env python3
"""Times multi_speculative_sampling(strategy="iid") on the headline pair (random-init weights, device RNG).

    python tools/multi_bench.py --width 4 [--draft llama-68m --target llama-2-13b --gamma 4 --prompt-len 128 --max-len 64]

Prints one JSON line: tokens/s of the whole call (prefills included), iterations, mean accepted length.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llmspeculativesampling_amd.config import load_config  # noqa: E402
from llmspeculativesampling_amd.engine import SpecDecModel  # noqa: E402
from llmspeculativesampling_amd.noise import DeviceNoise  # noqa: E402
from llmspeculativesampling_amd.sampling import multi_speculative_sampling, speculative_sampling  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--draft", default="llama-68m")
    ap.add_argument("--target", default="llama-2-13b")
    ap.add_argument("--width", type=int, default=4)
    ap.add_argument("--gamma", type=int, default=4)
    ap.add_argument("--prompt-len", type=int, default=128)
    ap.add_argument("--max-len", type=int, default=64)
    ap.add_argument("--reps", type=int, default=2)
    a = ap.parse_args()
    dcfg, tcfg = load_config(a.draft), load_config(a.target)
    dm = SpecDecModel.synthetic(dcfg, seed=0, dtype=torch.bfloat16)
    tm = SpecDecModel.synthetic(tcfg, seed=1, dtype=torch.bfloat16)
    prompt = torch.from_numpy(np.random.default_rng(3).integers(3, dcfg.vocab_size, size=(1, a.prompt_len))).cuda()
    out = {}
    for name, fn in (("single", lambda s: speculative_sampling(prompt, dm, tm, 2, None, a.max_len, gamma=a.gamma, top_k=20,
                                                               top_p=0.9, details=True, rng=DeviceNoise(s))),
                     ("multi", lambda s: multi_speculative_sampling(prompt, dm, tm, 2, None, a.max_len, gamma=a.gamma,
                                                                    width=a.width, strategy="iid", top_k=20, top_p=0.9,
                                                                    details=True, rng=DeviceNoise(s)))):
        fn(0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        toks = iters = 0
        accs = []
        for r in range(a.reps):
            o, d = fn(10 + r)
            toks += o.shape[1] - a.prompt_len
            iters += d["target_call_times"]
            accs += d["acc_len"]
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        out[name] = {"tokens_per_s": toks / dt, "ms_per_iteration": dt / iters * 1e3, "iterations": iters,
                     "mean_accept_len": float(np.mean(accs))}
    out["config"] = vars(a)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
