set -o pipefail
O=gpurun_out/r04b; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="--cpu-baseline 0 --accept-sweep 0"
for bs in 8 12 16; do timeout -k 10 300 python bench.py --steps 1 --warmup 1 $B --batch-streams $bs > $O/bench_throughput_b$bs.json 2>/dev/null; done
for g in 2 8; do timeout -k 10 300 python bench.py --steps 1 --warmup 1 $B --batch-streams 8 --gamma $g > $O/bench_throughput_b8_gamma$g.json 2>/dev/null; done
SD_BATCH_FUSED_TAIL=0 timeout -k 10 300 python bench.py --steps 1 --warmup 1 $B --batch-streams 8 > $O/bench_throughput_b8_tail0.json 2>/dev/null
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $O/profb -o b8 -- python3 bench.py --steps 1 --warmup 1 --cpu-baseline 0 --accept-sweep 0 --batch-streams 8 --profile-classes 0 > /dev/null 2>&1
t=$(find $O/profb -name "*kernel_trace.csv" | head -1); python tools/trace_by_grid.py $t > $O/kernel_by_grid_throughput_b8.txt 2>&1; rm -rf $O/profb
timeout -k 10 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_mfma8 -o m -- python3 bench.py --steps 1 --warmup 0 --cpu-baseline 0 --profile-classes 0 --accept-sweep 0 --max-len 16 --batch-streams 8 > $O/pmc_mfma_b8.log 2>&1
fm=$(find $O/pmc_mfma8 -name "*counter_collection.csv" | head -1); python tools/pmc_mfma.py $fm > $O/pmc_mfma_throughput_b8.json 2> $O/pmc_mfma_b8.err; rm -rf $O/pmc_mfma8
timeout -k 10 300 python tools/forward_rows_bench.py 5 9 16 24 40 60 64 72 80 127 132 150 200 256 > $O/forward_rows.txt 2>&1
timeout -k 10 400 python bench.py --draft opt-125m --target opt-13b --prompt-lens synthetic-c3 --steps 8 $B > $O/bench_opt13b_config3_c3prompts.json 2>/dev/null
timeout -k 10 500 python bench.py > $O/bench.json 2> $O/bench.err
ls $O
