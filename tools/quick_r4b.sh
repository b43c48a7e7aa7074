#!/bin/bash
# round-4 second session: targeted tests + A/B of the lock-step loop's sampling tail + draft step against a baseline library
set -o pipefail
O=gpurun_out/$1; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -4 $O/pytest.log
grep -q "pytest rc 0" $O/pytest.log || exit 1
BASE=$PWD/llmspeculativesampling_amd/libspecdec_base.so
for i in 1 2; do
  [ -f $BASE ] && SD_LIBSPECDEC=$BASE timeout -k 10 200 python tools/draft_step_bench.py 2>&1 | tail -1 | sed "s/^/base /" | tee -a $O/draft_ab.txt
  timeout -k 10 200 python tools/draft_step_bench.py 2>&1 | tail -1 | sed "s/^/new  /" | tee -a $O/draft_ab.txt
done
for mode in 0 1; do
SD_BATCH_FUSED_TAIL=$mode timeout -k 10 300 python bench.py --steps 1 --warmup 1 --cpu-baseline 0 --accept-sweep 0 --batch-streams 8 > $O/bench_b8_tail$mode.json 2>$O/bench_b8_tail$mode.err || exit 1
python - <<PY
import json
d=json.loads([l for l in open("$O/bench_b8_tail$mode.json") if l.startswith("{")][0])
print("b8 tail$mode: value", d["value"], "ms/step", d["ms_per_step"], "verify", d["roofline"]["avg_launch_ms"], d["roofline"]["frac"])
PY
done
