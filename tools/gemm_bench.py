#!/usr/bin/env python3
"""Kernel-level roofline run of the weight-streaming GEMM (sd_gemm_bf16) at the Llama-2-13b verify shapes.
Cycles through enough distinct weight copies that nothing is served from the 256 MiB Infinity Cache."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from llmspeculativesampling_amd._lib import lib, check  # noqa: E402

SHAPES = {"qkv": (15360, 5120), "o": (5120, 5120), "gate_up": (27648, 5120), "down": (5120, 13824),
          "lm_head": (32000, 5120), "draft_lm_head": (32000, 768), "draft_qkv": (2304, 768)}


def bench(name, N, K, M=5, units=None, iters=30):
    if units:
        os.environ["SD_GEMM_UNITS"] = str(units)
    elif "SD_GEMM_UNITS" in os.environ:
        del os.environ["SD_GEMM_UNITS"]
    nbytes = N * K * 2
    copies = max(2, int(600e6 // nbytes) + 1)
    W = [torch.randn(N, K, device="cuda", dtype=torch.bfloat16) * 0.02 for _ in range(copies)]
    Wp = []
    for w in W:
        o = torch.empty_like(w)
        check(lib.sd_pack_weight_bf16(w.data_ptr(), o.data_ptr(), N, K, None))
        Wp.append(o)
    torch.cuda.synchronize()
    x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
    xt = torch.zeros((M + 15) // 16 * 16 * K, device="cuda", dtype=torch.bfloat16)      # operand layout of the forward
    check(lib.sd_pack_activation_bf16(x.data_ptr(), xt.data_ptr(), M, K, None))
    part = torch.empty(max(64 * 64, 16 * max(M, 16)) * N, dtype=torch.float32, device="cuda")
    out = torch.empty(M, N, dtype=torch.float32, device="cuda")
    S = C.c_int(0)
    st = torch.cuda.current_stream().cuda_stream
    for i in range(3):
        check(lib.sd_gemm_bf16(Wp[i % copies].data_ptr(), xt.data_ptr(), 1, M, N, K, part.data_ptr(), part.numel(),
                               out.data_ptr(), C.byref(S), st))
    ref = x.float() @ W[2 % copies].float().t()
    err = float((out - ref).abs().max() / ref.abs().max())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        check(lib.sd_gemm_bf16(Wp[i % copies].data_ptr(), xt.data_ptr(), 1, M, N, K, part.data_ptr(), part.numel(),
                               None, None, st))
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print(f"{name:14s} N={N:6d} K={K:6d} M={M:2d} units={units or 'auto':>6} S={S.value:3d} {ms*1e3:8.1f} us "
          f"{nbytes/ms/1e6:8.1f} GB/s relerr={err:.2e}", flush=True)
    return nbytes / ms / 1e6


if __name__ == "__main__":
    which = [a for a in sys.argv[1:] if a != "rows"] or ([] if sys.argv[1:] == ["rows"] else ["qkv", "o", "gate_up", "down", "lm_head"])
    for n in which:
        N, K = SHAPES[n]
        for u in (None, 512, 768, 1024, 1280, 2048, 2560, 4096):
            bench(n, N, K, units=u)
    if not sys.argv[1:]:
        for M in (1, 5, 16, 17, 32, 48, 64, 128, 256):
            for n in ("qkv", "o", "gate_up", "down"):
                bench(n, *SHAPES[n], M=M)
    if sys.argv[1:] == ["rows"]:
        # 17..64 rows (stream-batched verify): the balanced one-workgroup-per-CU kernel against the streaming kernel
        for rows_kernel in ("1", "0"):
            os.environ["SD_GEMM_ROWS"] = rows_kernel
            for n in ("qkv", "o", "gate_up", "down", "lm_head"):
                for M in (20, 40, 60, 64):
                    print("balanced" if rows_kernel == "1" else "stream  ", end=" ")
                    bench(n, *SHAPES[n], M=M)
