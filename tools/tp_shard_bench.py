#!/usr/bin/env python3
"""What ONE rank of BASELINE config 5 does per verify step, on the one GPU this pool offers: rank 0's Megatron shard of
Llama-2-70b under TP = 8 (8 of 64 query heads, 1 of 8 KV heads, 3584 of 28672 MLP columns, all 80 layers: 8.82 G streamed
parameters = 17.6 GB of bf16), fp8 KV arena, a 128-token prompt and gamma + 1 = 5 verify rows.

Two timings of the same shard (HIP events around the verify forwards):
  * `alone`     - no group: the O / down projections' partial sums feed the residual directly (a wrong model, the right
                  bytes): the rank's own kernels.
  * `rccl-1`    - a ONE-rank RCCL communicator kept on the session (SD_TP_FORCE=1): every O / down projection goes through the
                  slab fold + ncclAllReduce + residual of the real tensor-parallel path, with a collective that has nobody
                  to talk to - the launch overhead of 160 all-reduces per verify without their xGMI latency.
Neither is the TP = 8 number (that needs eight GPUs; SCALE is the driver's to run); together they bound what the shard's
kernels cost and what the collective's launches add."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from llmspeculativesampling_amd import tp
from llmspeculativesampling_amd.config import load_config

ap = argparse.ArgumentParser()
ap.add_argument("--world", type=int, default=8)
ap.add_argument("--layers", type=int, default=0, help="0 = all of the model's layers")
ap.add_argument("--prompt-len", type=int, default=128)
ap.add_argument("--rows", type=int, default=5)
ap.add_argument("--reps", type=int, default=10)
a = ap.parse_args()

cfg = load_config("llama-2-70b")
if a.layers:
    from dataclasses import replace
    cfg = replace(cfg, num_hidden_layers=a.layers)
mp = a.prompt_len + a.rows + 16
ids = torch.from_numpy(np.random.default_rng(2).integers(3, cfg.vocab_size, size=(mp,))).to(torch.int32).cuda()


def time_verify(ses):
    ses.forward(ids[:a.prompt_len], 0)
    for _ in range(2):
        ses.rollback(a.prompt_len)
        ses.forward(ids[a.prompt_len:a.prompt_len + a.rows], a.rows)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.reps):
        ses.rollback(a.prompt_len)
        ses.forward(ids[a.prompt_len:a.prompt_len + a.rows], a.rows)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / a.reps


m = tp.synthetic_shard(cfg, 0, a.world, seed=2, dtype=torch.bfloat16, max_pos=mp)
wbytes = m.cfg.n_params(streamed_only=True) * 2          # bf16 bytes the shard's GEMMs stream per forward
t_alone = time_verify(m.new_session(mp, kv_dtype="fp8"))
print(f"shard 0 of {a.world}: {m.cfg.num_hidden_layers} layers, {wbytes / 1e9:.2f} GB streamed per verify")
print(f"alone : verify over {a.rows} rows {t_alone:.3f} ms = {wbytes / t_alone / 1e9:.2f} TB/s = {wbytes / t_alone / 1e9 / 8:.3f} of 8 TB/s")
grp = tp.TPGroup.rccl(0, 1, lambda b: b)
os.environ["SD_TP_FORCE"] = "1"
try:
    ses = m.new_session(mp, kv_dtype="fp8")
    grp.bind(ses)
    t_rccl = time_verify(ses)
finally:
    os.environ.pop("SD_TP_FORCE", None)
n_ar = 2 * m.cfg.num_hidden_layers
print(f"rccl-1: verify over {a.rows} rows {t_rccl:.3f} ms: {n_ar} fold + one-rank ncclAllReduce pairs add {(t_rccl - t_alone) * 1e3 / n_ar:.1f} us each "
      f"({wbytes / t_rccl / 1e9 / 8:.3f} of 8 TB/s)")
