#!/usr/bin/env python3
"""Time sd_norm_sample / sd_norm_probs in isolation for a few input distributions."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from llmspeculativesampling_amd._lib import lib, check

def t(fn, iters=200):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

st = torch.cuda.current_stream().cuda_stream
for V in (32000, 50272):
    for name, x in (("gauss4", torch.randn(V, device="cuda") * 4), ("gauss4_bf16grid", (torch.randn(V, device="cuda") * 4).bfloat16().float()),
                    ("gauss0.3", torch.randn(V, device="cuda") * 0.3), ("const", torch.zeros(V, device="cuda"))):
        out = torch.empty(V, device="cuda"); tok = torch.zeros(1, dtype=torch.int32, device="cuda"); err = torch.zeros(2, dtype=torch.int32, device="cuda")
        wst = torch.empty(lib.sd_norm_workspace_bytes(1), dtype=torch.uint8, device="cuda")
        for (k, p, ws) in ((20, 0.9, None), (20, 0.9, wst.data_ptr()), (20, 0.0, wst.data_ptr()), (0, 0.0, None), (0, 0.9, None), (200, 0.9, None)):
            us = t(lambda: check(lib.sd_norm_sample(x.data_ptr(), V, 1.0, k, p, 0, out.data_ptr(), err.data_ptr(), None, 1, 2, tok.data_ptr(), err[1:].data_ptr(), ws, st)))
            us2 = t(lambda: check(lib.sd_norm_probs(x.data_ptr(), 1, V, V, 1.0, k, p, 0, out.data_ptr(), V, err.data_ptr(), ws, st)))
            print(f"V={V} {name:16s} k={k:3d} p={p} ws={ws is not None}: norm_sample {us:7.1f} us   norm_probs {us2:7.1f} us  nnz={int((out>0).sum())}", flush=True)
