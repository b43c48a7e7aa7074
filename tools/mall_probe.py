#!/usr/bin/env python3
"""How fast does gemm_bf16_stream run when its weights sit in the 256 MiB Infinity Cache (same matrix
re-read back to back) vs streamed from HBM (cycling through > 600 MB of copies)?"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from llmspeculativesampling_amd._lib import lib, check  # noqa: E402


def run(N, K, copies, iters=40, M=5):
    W = [torch.randn(N, K, device="cuda", dtype=torch.bfloat16) * 0.02 for _ in range(copies)]
    Wp = []
    for w in W:
        o = torch.empty_like(w)
        check(lib.sd_pack_weight_bf16(w.data_ptr(), o.data_ptr(), N, K, None))
        Wp.append(o)
    del W
    x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
    part = torch.empty(16 * 16 * N, dtype=torch.float32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    for i in range(5):
        check(lib.sd_gemm_bf16(Wp[i % copies].data_ptr(), x.data_ptr(), 0, M, N, K, part.data_ptr(), part.numel(), None, None, st))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        check(lib.sd_gemm_bf16(Wp[i % copies].data_ptr(), x.data_ptr(), 0, M, N, K, part.data_ptr(), part.numel(), None, None, st))
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    return ms * 1e3, N * K * 2 / ms / 1e6


for name, (N, K) in {"o": (5120, 5120), "down": (5120, 13824), "qkv": (15360, 5120), "gate_up": (27648, 5120)}.items():
    nb = N * K * 2
    for copies in (1, 2, max(2, int(700e6 // nb) + 1)):
        us, gbs = run(N, K, copies)
        print(f"{name:8s} {nb/1e6:7.1f} MB x {copies:2d} copies ({copies*nb/1e6:7.1f} MB working set): {us:7.1f} us  {gbs:8.1f} GB/s", flush=True)
