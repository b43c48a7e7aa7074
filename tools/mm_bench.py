#!/usr/bin/env python3
"""Prefill GEMMs (145..256 rows) at the Llama-2-13b layer shapes: gemm_bf16_tiled against gemm_bf16_mm (mm_kernels.h), through the
public one-off entry sd_gemm_bf16.  Integer-valued operands, so the result must equal an integer matmul bit for bit whatever
the summation order; timings cycle through enough weight copies that nothing is served from the 256 MiB Infinity Cache.

    python tools/mm_bench.py [rows ...]            (default rows: 256 192 160)
"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from llmspeculativesampling_amd._lib import lib, check  # noqa: E402

SHAPES = {"qkv": (15360, 5120), "o": (5120, 5120), "gate_up": (27648, 5120), "down": (5120, 13824)}


def setenv(**kw):
    for k, v in kw.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = str(v)


def make(N, K, M):
    g = torch.Generator(device="cuda").manual_seed(N + K + M)
    copies = max(2, int(600e6 // (N * K * 2)) + 1)
    W = [torch.randint(-2, 3, (N, K), device="cuda", generator=g).to(torch.bfloat16) for _ in range(copies)]
    Wp = []
    for w in W:
        o = torch.empty_like(w)
        check(lib.sd_pack_weight_bf16(w.data_ptr(), o.data_ptr(), N, K, None))
        Wp.append(o)
    x = torch.randint(-4, 5, (M, K), device="cuda", generator=g).to(torch.bfloat16)
    xt = torch.zeros((M + 15) // 16 * 16 * K, device="cuda", dtype=torch.bfloat16)
    check(lib.sd_pack_activation_bf16(x.data_ptr(), xt.data_ptr(), M, K, None))
    ref = x.float() @ W[1].float().t()
    return W, Wp, x, xt, ref


def run(tag, N, K, M, Wp, xt, ref, iters=20):
    part = torch.empty(16 * 256 * N, dtype=torch.float32, device="cuda")
    out = torch.empty(M, N, dtype=torch.float32, device="cuda")
    S = C.c_int(0)
    st = torch.cuda.current_stream().cuda_stream
    rc = lib.sd_gemm_bf16(Wp[1].data_ptr(), xt.data_ptr(), 1, M, N, K, part.data_ptr(), part.numel(), out.data_ptr(), C.byref(S), st)
    if rc != 0:
        print(f"{tag}: rc={rc}", flush=True)
        return None
    torch.cuda.synchronize()
    exact = bool(torch.equal(out, ref))
    for i in range(3):
        check(lib.sd_gemm_bf16(Wp[i % len(Wp)].data_ptr(), xt.data_ptr(), 1, M, N, K, part.data_ptr(), part.numel(), None, None, st))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        check(lib.sd_gemm_bf16(Wp[i % len(Wp)].data_ptr(), xt.data_ptr(), 1, M, N, K, part.data_ptr(), part.numel(), None, None, st))
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / iters * 1e3
    tf = 2.0 * M * N * K / us / 1e6
    print(f"{tag:34s} S={S.value:2d} {us:8.1f} us {tf:7.1f} TFLOP/s {N*K*2/us/1e6:6.2f} TB/s(w) exact={exact}", flush=True)
    return us


if __name__ == "__main__":
    rows = [int(a) for a in sys.argv[1:]] or [256, 192, 160]
    for M in rows:
        tot = {}
        for name, (N, K) in SHAPES.items():
            W, Wp, x, xt, ref = make(N, K, M)
            setenv(SD_GEMM_MM=0, SD_MM_MTW=None, SD_MM_S=None, SD_MM_NT=None, SD_GEMM_ROWS_MAX=64, SD_MM_SLABS_MIN=1000)
            us = run(f"M={M} {name:8s} tiled", N, K, M, Wp, xt, ref)
            tot.setdefault("tiled", 0.0)
            tot["tiled"] += us or 0.0
            if M <= 144:
                setenv(SD_GEMM_ROWS_MAX=None, SD_GEMM_MM=1)      # (the engine's default planner without the k-slab GEMMs on gemm_bf16_mm)
                us = run(f"M={M} {name:8s} rows (balanced kernel)", N, K, M, Wp, xt, ref)
                tot.setdefault("rows", 0.0)
                tot["rows"] += us or 0.0
                setenv(SD_GEMM_ROWS_MAX=64)
            best = None
            for mtw in (4, 2):
                if mtw == 2 and False:
                    continue
                for S in ((0, 1, 2, 3, 4, 6) if name != "gate_up" else (0, 1, 2)):
                    for nt in (1,):
                        setenv(SD_GEMM_MM=1, SD_MM_MTW=mtw, SD_MM_S=S or None, SD_MM_SLABS_MIN=1)
                        us = run(f"M={M} {name:8s} mm mtw={mtw} S={S}", N, K, M, Wp, xt, ref)
                        if us and (best is None or us < best):
                            best = us
            tot.setdefault("mm_best", 0.0)
            tot["mm_best"] += best or 0.0
            del W, Wp, x, xt, ref
            torch.cuda.empty_cache()
        print(f"== M={M}: per layer tiled {tot['tiled']:.1f} us, mm (best per shape) {tot['mm_best']:.1f} us" +
              (f", balanced rows kernel {tot['rows']:.1f} us" if "rows" in tot else ""), flush=True)
