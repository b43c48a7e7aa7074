#!/bin/bash
# A/B/C bench of the verify step under env settings given as arguments ("-" = defaults); prints tokens/s, verify ms, fraction
for a in "$@"; do
  if [ "$a" != "-" ]; then export $a; fi
  timeout -k 10 200 python bench.py --steps 4 --warmup 1 --sweep-steps 0 --cpu-baseline 0 > gpurun_out/ab_bench.json 2>/dev/null || exit 1
  python -c "
import json; d=json.loads(open('gpurun_out/ab_bench.json').read().strip().splitlines()[-1]); print('$a', round(d['value'],1), round(d['roofline']['avg_launch_ms'],4), round(d['roofline']['frac'],4))"
  if [ "$a" != "-" ]; then unset ${a%%=*}; fi
done
