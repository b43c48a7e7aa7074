#!/bin/bash
# Timing of the balanced many-row GEMM at the 13b shapes.  GR_PROBE variants (results wrong, timing only) need a rebuild:
#   SD_EXTRA_HIPCC_FLAGS=-DGR_PROBE=7 python -m llmspeculativesampling_amd._build --force
python - <<'PY'
import os, sys
sys.path.insert(0, ".")
sys.argv = ["x"]
import importlib.util
spec = importlib.util.spec_from_file_location("gb", "tools/gemm_bench.py")
gb = importlib.util.module_from_spec(spec); spec.loader.exec_module(gb)
for n in ("qkv", "o", "gate_up", "down"):
    for M in (20, 40, 60, 64):
        gb.bench(n, *gb.SHAPES[n], M=M)
PY
