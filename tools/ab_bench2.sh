#!/bin/bash
# A/B of extra bench configurations: $1 = env assignment for A; B = defaults.  Remaining args: bench.py flags.
a=$1; shift
for tag in A B; do
  if [ $tag = A ]; then export $a; else unset ${a%%=*}; fi
  timeout -k 10 500 python bench.py "$@" --cpu-baseline 0 --accept-sweep 0 > gpurun_out/ab_bench2.json 2>/dev/null || exit 1
  python -c "
import json; d=json.loads(open('gpurun_out/ab_bench2.json').read().strip().splitlines()[-1]); print('$tag', '$*', round(d['value'],1), round(d['roofline']['avg_launch_ms'],4), round(d['roofline']['frac'],4))"
done
