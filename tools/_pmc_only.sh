#!/bin/bash
set -o pipefail
O=gpurun_out/r03; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do timeout -k 10 400 rocprofv3 --pmc $c --output-format csv -d $O/pmc_$c -o p -- python3 bench.py --steps 1 --warmup 0 --cpu-baseline 0 --profile-classes 0 --accept-sweep 0 --max-len 32 > $O/pmc_$c.log 2>&1; done
ff=$(find $O/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1); fw=$(find $O/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1)
python tools/pmc_traffic.py $ff $fw > $O/pmc_traffic.json 2> $O/pmc_traffic.err; rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
head -14 $O/pmc_traffic.json; tail -3 $O/pmc_traffic.json
