#!/bin/bash
# quick A/B evidence for one code state: GPU tests, stamps (cold / warm), headline bench, throughput bench, draft step
set -o pipefail
O=gpurun_out/$1; mkdir -p $O
shift
for what in "$@"; do
case $what in
tests) timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -3 $O/pytest.log;;
stamps) timeout -k 10 120 python tools/ao_stamps.py --layers 1 > $O/stamps_cold.txt 2>&1; timeout -k 10 200 python tools/ao_stamps.py --layers 8 --flush 0 > $O/stamps_warm.txt 2>&1; grep "== rep\|O workgroups" $O/stamps_cold.txt $O/stamps_warm.txt;;
bench) timeout -k 10 400 python bench.py --steps 4 --cpu-baseline 0 --accept-sweep 0 > $O/bench.json 2> $O/bench.err; python - <<PY
import json
d=json.loads([l for l in open("$O/bench.json") if l.startswith("{")][0])
print("bench: value", d["value"], "ms/step", d["ms_per_step"], "roofline", d["roofline"]["achieved"], d["roofline"]["frac"], {k: d.get(k) for k in ("verify_ms","draft_step_avg_ms","target_prefill_avg_ms") if k in d})
print({k: v for k, v in d.items() if "ms" in k and not isinstance(v, (dict, list))})
PY
;;
b8) timeout -k 10 300 python bench.py --steps 1 --warmup 1 --cpu-baseline 0 --accept-sweep 0 --batch-streams 8 $B8_EXTRA > $O/bench_b8.json 2>$O/bench_b8.err; python - <<PY
import json
d=json.loads([l for l in open("$O/bench_b8.json") if l.startswith("{")][0])
print("b8: value", d["value"], "roofline", d["roofline"])
PY
;;
g8) timeout -k 10 300 python bench.py --gamma 8 --steps 3 --cpu-baseline 0 --accept-sweep 0 > $O/bench_g8.json 2>$O/bench_g8.err; python - <<PY
import json
d=json.loads([l for l in open("$O/bench_g8.json") if l.startswith("{")][0])
print("g8: value", d["value"], "roofline", d["roofline"]["achieved"], d["roofline"]["frac"])
PY
;;
opt) timeout -k 10 400 python bench.py --draft opt-125m --target opt-13b --steps 3 --cpu-baseline 0 --accept-sweep 0 > $O/bench_opt.json 2>$O/bench_opt.err; python - <<PY
import json
d=json.loads([l for l in open("$O/bench_opt.json") if l.startswith("{")][0])
print("opt: value", d["value"], "roofline", d["roofline"]["achieved"], d["roofline"]["frac"], {k: v for k, v in d.items() if "ms" in k and not isinstance(v, (dict, list))})
PY
;;
draft) timeout -k 10 200 python tools/draft_step_bench.py > $O/draft_step.txt 2>&1; tail -5 $O/draft_step.txt;;
esac
done
