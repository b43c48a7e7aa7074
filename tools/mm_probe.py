#!/usr/bin/env python3
"""gemm_bf16_mm variants (SD_MM_VAR: stage depth, stagger, k-rotation), 13b shapes."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from mm_bench import SHAPES, make, run, setenv  # noqa: E402

VARS4 = {0: "kt2 nbuf3", 1: "kt2 nbuf3 stag", 2: "kt2 nbuf3 rot", 3: "kt2 nbuf3 stag rot", 4: "kt1 nbuf6", 5: "kt1 nbuf6 stag rot", 6: "kt1 nbuf6 rot"}
VARS2 = {0: "kt2 nbuf3", 1: "kt2 nbuf4 stag rot", 2: "kt2 nbuf3 rot", 3: "kt2 nbuf3 stag rot", 4: "kt1 nbuf6", 5: "kt1 nbuf8 stag rot"}

if __name__ == "__main__":
    rows = [int(a) for a in sys.argv[1:]] or [256]
    for M in rows:
        for name, (N, K) in SHAPES.items():
            W, Wp, x, xt, ref = make(N, K, M)
            for mtw, vs in ((4, VARS4), (2, VARS2)):
                for v, what in vs.items():
                    setenv(SD_GEMM_MM=1, SD_MM_MTW=mtw, SD_MM_S=None, SD_MM_NT=1, SD_MM_VAR=v, SD_GEMM_ROWS_MAX=64)
                    run(f"M={M} {name:8s} mtw{mtw} {what}", N, K, M, Wp, xt, ref)
            del W, Wp, x, xt, ref
            torch.cuda.empty_cache()
