#!/bin/bash
# draft-step A/B on one box: llmspeculativesampling_amd/libspecdec_base.so against the current library, both draft models
O=gpurun_out/$1; mkdir -p $O
BASE=$PWD/llmspeculativesampling_amd/libspecdec_base.so
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "small_model_path or native_bf16 or draft or opt_bf16" > $O/pytest.log 2>&1; tail -2 $O/pytest.log
for i in 1 2; do for d in llama-68m opt-125m; do
  SD_LIBSPECDEC=$BASE timeout -k 10 200 python tools/draft_step_bench.py --draft $d 2>&1 | tail -1 | sed "s/^/base $d /" | tee -a $O/draft_ab.txt
  timeout -k 10 200 python tools/draft_step_bench.py --draft $d 2>&1 | tail -1 | sed "s/^/new  $d /" | tee -a $O/draft_ab.txt
done; done
