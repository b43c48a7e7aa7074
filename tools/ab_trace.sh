#!/bin/bash
# A/B kernel-trace of the verify step: $1 = env assignment for A (e.g. SD_NORM_ON_LOAD=0), B = defaults; further arguments
# go to bench.py.  Output gpurun_out/ab/.
set -o pipefail
O=gpurun_out/ab; mkdir -p $O
A=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for tag in A B; do
  if [ $tag = A ]; then export $A; else unset ${A%%=*}; fi
  timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $O/prof$tag -o t -- python3 bench.py --steps 1 --warmup 1 --cpu-baseline 0 --profile-classes 0 --accept-sweep 0 "$@" > $O/bench_$tag.log 2>&1 || exit 1
  t=$(find $O/prof$tag -name "*kernel_trace.csv" | head -1)
  python tools/trace_by_grid.py $t > $O/by_grid_$tag.txt 2>&1
  python tools/trace_gaps.py $t > $O/gaps_$tag.txt 2>&1
  rm -rf $O/prof$tag
  tail -1 $O/bench_$tag.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', d['value'], d['roofline']['avg_launch_ms'])"
done
