#!/bin/bash
# Round-4 evidence run (one gpurun call): bench lines, rocprofv3 kernel stats, PMC traffic + MFMA passes.  Outputs under gpurun_out/r04/.
set -o pipefail
O=gpurun_out/r04; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="--cpu-baseline 0 --accept-sweep 0"
echo "[1] default bench"; timeout -k 10 500 python bench.py > $O/bench.json 2> $O/bench.err
echo "[2] throughput"; for bs in 8 12 16; do timeout -k 10 300 python bench.py --steps 1 --warmup 1 $B --batch-streams $bs > $O/bench_throughput_b$bs.json 2>/dev/null; done
for g in 2 8; do timeout -k 10 300 python bench.py --steps 1 --warmup 1 $B --batch-streams 8 --gamma $g > $O/bench_throughput_b8_gamma$g.json 2>/dev/null; done
echo "[3] gamma"; for g in 2 8; do timeout -k 10 300 python bench.py --gamma $g --steps 3 $B > $O/bench_gamma$g.json 2>/dev/null; done
echo "[4] config 3"; timeout -k 10 400 python bench.py --draft opt-125m --target opt-13b --steps 3 $B > $O/bench_opt13b_config3.json 2>/dev/null
timeout -k 10 400 python bench.py --draft opt-125m --target opt-13b --prompt-lens synthetic-c3 --steps 8 $B > $O/bench_opt13b_config3_c3prompts.json 2>/dev/null
echo "[5] kernel stats"; timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o ks -- python3 bench.py --steps 1 --warmup 1 --cpu-baseline 0 --profile-classes 0 --accept-sweep 0 > $O/prof_bench.log 2>&1
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); cp $f $O/rocprofv3_kernel_stats.csv; python tools/rocprof_summary.py $f > $O/rocprofv3_kernel_stats_summary.txt 2>&1
t=$(find $O/prof -name "*kernel_trace.csv" | head -1); python tools/trace_by_grid.py $t > $O/kernel_by_grid.txt 2>&1; rm -f $t
echo "[6] pmc traffic"; for c in FETCH_SIZE WRITE_SIZE; do timeout -k 10 400 rocprofv3 --pmc $c --output-format csv -d $O/pmc_$c -o p -- python3 bench.py --steps 1 --warmup 0 --cpu-baseline 0 --profile-classes 0 --accept-sweep 0 --max-len 32 > $O/pmc_$c.log 2>&1; done
ff=$(find $O/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1); fw=$(find $O/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1)
python tools/pmc_traffic.py $ff $fw > $O/pmc_traffic.json 2> $O/pmc_traffic.err; rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/prof
echo "[7] pmc mfma"; timeout -k 10 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_mfma -o m -- python3 bench.py --steps 1 --warmup 0 --cpu-baseline 0 --profile-classes 0 --accept-sweep 0 --max-len 32 > $O/pmc_mfma.log 2>&1
fm=$(find $O/pmc_mfma -name "*counter_collection.csv" | head -1); python tools/pmc_mfma.py $fm > $O/pmc_mfma.json 2> $O/pmc_mfma.err; rm -rf $O/pmc_mfma
timeout -k 10 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_mfma8 -o m -- python3 bench.py --steps 1 --warmup 0 --cpu-baseline 0 --profile-classes 0 --accept-sweep 0 --max-len 16 --batch-streams 8 > $O/pmc_mfma_b8.log 2>&1
fm=$(find $O/pmc_mfma8 -name "*counter_collection.csv" | head -1); python tools/pmc_mfma.py $fm > $O/pmc_mfma_throughput_b8.json 2> $O/pmc_mfma_b8.err; rm -rf $O/pmc_mfma8
echo "[8] throughput by grid"; timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $O/profb -o b8 -- python3 bench.py --steps 1 --warmup 1 --cpu-baseline 0 --accept-sweep 0 --batch-streams 8 --profile-classes 0 > /dev/null 2>&1
t=$(find $O/profb -name "*kernel_trace.csv" | head -1); python tools/trace_by_grid.py $t > $O/kernel_by_grid_throughput_b8.txt 2>&1; rm -rf $O/profb
echo "[9] draft step"; timeout -k 10 200 python tools/draft_step_bench.py - SD_SMALL_PATH=0 SD_SMALL_PATH=2 > $O/draft_step_bench.txt 2>&1
timeout -k 10 200 python tools/draft_step_bench.py --draft opt-125m - SD_SMALL_PATH=0 SD_SMALL_PATH=2 >> $O/draft_step_bench.txt 2>&1
echo "[9b] lock-step tail A/B"; for mode in 0 1; do SD_BATCH_FUSED_TAIL=$mode timeout -k 10 300 python bench.py --steps 1 --warmup 1 $B --batch-streams 8 > $O/bench_throughput_b8_tail$mode.json 2>/dev/null; done
echo "[9c] tp shard"; timeout -k 10 300 python tools/tp_shard_bench.py > $O/tp8_shard_one_gpu.txt 2>&1
echo "[9d] small-path by grid"; bash tools/trace_cmd.sh r04/small_path_llama68m - tools/draft_step_bench.py --max-len 64 > /dev/null 2>&1; bash tools/trace_cmd.sh r04/small_path_opt125m - tools/draft_step_bench.py --draft opt-125m --max-len 64 > /dev/null 2>&1
echo "[10] 70b"; timeout -k 10 500 python bench.py --target llama-2-70b --kv-dtype fp8 --steps 2 $B > $O/bench_llama70b_fp8kv_1gpu.json 2>/dev/null
echo "[11] rows"; timeout -k 10 300 python tools/forward_rows_bench.py 5 9 16 40 64 72 127 132 150 200 256 > $O/forward_rows.txt 2>&1
SD_GEMM_MM=0 SD_PREFILL_ATTN=0 timeout -k 10 300 python tools/forward_rows_bench.py 127 132 150 200 256 > $O/forward_rows_round3_prefill_path.txt 2>&1
echo "[12] prefill GEMM sweep"; timeout -k 10 500 python tools/mm_bench.py 256 200 132 > $O/mm_gemm_sweep.txt 2>&1
ls -la $O
