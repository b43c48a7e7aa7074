import os, sys
sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import gemm_bench as g
CASES = (("base", {}), ("W16", {"SD_PROBE_W16": "1"}))
for rep in range(3):
    for tag, env in CASES:
        for k in ("SD_PROBE_W16",):
            os.environ.pop(k, None)
        os.environ.update(env)
        for n, units in (("down", None), ("o", None)):
            print(f"{tag:7s}", end=" ")
            g.bench(n, *g.SHAPES[n], M=5, units=units, iters=60)
