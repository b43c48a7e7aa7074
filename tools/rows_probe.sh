#!/bin/bash
# timing probes of the balanced many-row GEMM (results are wrong for probe != 0): which part of an iteration costs what
for p in 0 1 2 3 4 7; do
  echo "== SD_ROWS_PROBE=$p"
  SD_ROWS_PROBE=$p SD_GEMM_ROWS=1 python - <<'PY'
import os, sys
sys.path.insert(0, ".")
sys.argv = ["x"]
import importlib.util
spec = importlib.util.spec_from_file_location("gb", "tools/gemm_bench.py")
gb = importlib.util.module_from_spec(spec); spec.loader.exec_module(gb)
for n in ("qkv", "o", "gate_up", "down"):
    for M in (40, 64):
        gb.bench(n, *gb.SHAPES[n], M=M)
PY
done
