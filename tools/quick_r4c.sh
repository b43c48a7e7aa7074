#!/bin/bash
# A/B of the whole library against llmspeculativesampling_amd/libspecdec_base.so on ONE box: full GPU tests, then headline /
# 8-stream / opt / gamma-8 bench lines and the draft step for both libraries
set -o pipefail
O=gpurun_out/$1; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -4 $O/pytest.log
grep -q "pytest rc 0" $O/pytest.log || exit 1
BASE=$PWD/llmspeculativesampling_amd/libspecdec_base.so
B="--cpu-baseline 0 --accept-sweep 0 --profile-classes 0"
show() { python - "$1" "$2" <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][0]); r=d["roofline"]
print(sys.argv[2], "value %.1f" % d["value"], "verify %.3f ms frac %.3f" % (r["avg_launch_ms"], r["frac"]), "draft", r.get("draft_step_avg_ms"), "prefill", r.get("target_prefill_avg_ms"))
PY
}
for lib in base new; do
  if [ $lib = base ]; then export SD_LIBSPECDEC=$BASE; else unset SD_LIBSPECDEC; fi
  timeout -k 10 300 python bench.py --steps 3 $B > $O/bench_$lib.json 2>$O/bench_$lib.err && show $O/bench_$lib.json "headline $lib"
  timeout -k 10 300 python bench.py --steps 1 --warmup 1 $B --batch-streams 8 > $O/b8_$lib.json 2>$O/b8_$lib.err && show $O/b8_$lib.json "b8 $lib"
  timeout -k 10 300 python bench.py --draft opt-125m --target opt-13b --steps 2 $B > $O/opt_$lib.json 2>$O/opt_$lib.err && show $O/opt_$lib.json "opt $lib"
  timeout -k 10 300 python bench.py --gamma 8 --steps 2 $B > $O/g8_$lib.json 2>$O/g8_$lib.err && show $O/g8_$lib.json "g8 $lib"
  timeout -k 10 200 python tools/draft_step_bench.py 2>&1 | tail -1 | sed "s/^/$lib /" | tee -a $O/draft_ab.txt
done
