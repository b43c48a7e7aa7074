#!/usr/bin/env python3
"""Draft-step micro-benchmark: llama-68m (or --draft) drafting for a same-sized target, native device-RNG loop, so an
iteration is gamma draft steps + one small verify; prints the HIP-event time of the draft phase per step (the quantity
bench.py reports as roofline.draft_step_avg_ms) for each value of the environment knobs given on the command line.

    python tools/draft_step_bench.py [--gamma 8] [--max-len 256] KEY=VAL[,KEY=VAL...] ...
Each positional argument is one configuration (comma-separated environment settings, '-' = defaults)."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from llmspeculativesampling_amd.config import load_config
from llmspeculativesampling_amd.engine import SpecDecModel
from llmspeculativesampling_amd.noise import DeviceNoise
from llmspeculativesampling_amd.sampling import speculative_sampling

ap = argparse.ArgumentParser()
ap.add_argument("--draft", default="llama-68m")
ap.add_argument("--gamma", type=int, default=8)
ap.add_argument("--max-len", type=int, default=256)
ap.add_argument("--prompt-len", type=int, default=128)
ap.add_argument("configs", nargs="*", default=["-"])
a = ap.parse_args()
cfg = load_config(a.draft)
mp = a.prompt_len + a.max_len + a.gamma + 8
dm = SpecDecModel.synthetic(cfg, seed=1, dtype=torch.bfloat16, max_pos=mp)
tm = SpecDecModel.synthetic(cfg, seed=2, dtype=torch.bfloat16, max_pos=mp)
prompt = torch.randint(3, cfg.vocab_size, (1, a.prompt_len), generator=torch.Generator().manual_seed(7)).cuda()
for conf in a.configs:
    sets = {} if conf == "-" else dict(kv.split("=") for kv in conf.split(","))
    for k, v in sets.items():
        os.environ[k] = v
    res = []
    for rep in range(3):
        logs = {"draft_ms": [], "target": []}
        torch.cuda.synchronize()
        t0 = time.time()
        out, d = speculative_sampling(prompt, dm, tm, -1, None, a.max_len, gamma=a.gamma, top_k=20, top_p=0.9, details=True,
                                      rng=DeviceNoise(seed=5 + rep), _event_logs=logs)
        torch.cuda.synchronize()
        wall = time.time() - t0
        res.append((float(np.median(logs["draft_ms"][1:])) / a.gamma * 1e3, wall / d["target_call_times"] * 1e3, d["target_call_times"]))
    for k in sets:
        os.environ.pop(k)
    print(f"{conf:60s} draft step {min(r[0] for r in res):7.1f} us (median of iterations, best of 3); "
          f"iteration wall {min(r[1] for r in res):6.3f} ms; tokens {out[0, -4:].tolist()}", flush=True)
