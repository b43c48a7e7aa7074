#!/usr/bin/env python3
"""Time of one target forward (llama-2-13b, random-init bf16) over M new rows at a ~200-token context, the shape of
a stream-batched verify (M = streams * (gamma+1)) or of one prefill chunk (M = 64):

    python tools/forward_rows_bench.py [M ...]          # env SD_GEMM_NTW / SD_GEMM_UNITS select GEMM variants
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llmspeculativesampling_amd.config import load_config  # noqa: E402
from llmspeculativesampling_amd.engine import SpecDecModel  # noqa: E402


def main():
    Ms = [int(a) for a in sys.argv[1:]] or [5, 16, 40, 64]
    cfg = load_config(os.environ.get("TARGET", "llama-2-13b"))
    m = SpecDecModel.synthetic(cfg, seed=1, dtype=torch.bfloat16)
    ctx = 192
    toks = torch.from_numpy(np.random.default_rng(0).integers(3, cfg.vocab_size, size=ctx + 256)).to(torch.int32).cuda()
    ses = m.new_session(ctx + 260)
    for lo in range(0, ctx, 64):
        ses.forward(toks[lo:lo + 64], 0)
    variants = [v for v in os.environ.get("VARIANTS", "auto").split(",")]     # tiles per wave; auto = the engine's policy
    for M in Ms:
        for v in variants:
            if v == "auto":
                os.environ.pop("SD_GEMM_NTW", None)
            else:
                os.environ["SD_GEMM_NTW"] = v
            for _ in range(2):
                ses.rollback(ctx)
                ses.forward(toks[ctx:ctx + M], min(M, 5))
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 5
            e0.record()
            for _ in range(reps):
                ses.rollback(ctx)
                ses.forward(toks[ctx:ctx + M], min(M, 5))
            e1.record()
            torch.cuda.synchronize()
            ses.profile(True)
            ses.rollback(ctx)
            ses.forward(toks[ctx:ctx + M], min(M, 5))
            prof = ses.profile_read()
            ses.profile(False)
            print(f"M={M:3d} ntw={v} forward {e0.elapsed_time(e1) / reps:7.3f} ms   per class (event-timed): "
                  + ", ".join(f"{k} {ms:.2f}" for k, (ms, n) in prof.items()), flush=True)


if __name__ == "__main__":
    main()
