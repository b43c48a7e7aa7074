#!/bin/bash
# kernel trace of any python command: tools/trace_cmd.sh <out-tag> <ENV=VAL[,ENV=VAL]|-> <script.py> [args...]
# -> gpurun_out/<out-tag>_by_grid.txt, _gaps.txt (rocprofv3 --kernel-trace; the program itself follows `--`)
set -o pipefail
O=gpurun_out/$1; ENVS=$2; shift 2
mkdir -p $(dirname $O)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
if [ "$ENVS" != "-" ]; then IFS=',' read -ra KV <<< "$ENVS"; for kv in "${KV[@]}"; do export "$kv"; done; fi
rm -rf ${O}_prof
timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv -d ${O}_prof -o t -- python3 "$@" > ${O}_run.log 2>&1 || { tail -5 ${O}_run.log; exit 1; }
t=$(find ${O}_prof -name "*kernel_trace.csv" | head -1)
python tools/trace_by_grid.py $t 60 > ${O}_by_grid.txt 2>&1
python tools/trace_gaps.py $t > ${O}_gaps.txt 2>&1
rm -rf ${O}_prof
tail -2 ${O}_run.log
