#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
for e in 0 1; do
  echo "OMK_LORA_EXT=$e"
  OMK_LORA_EXT=$e timeout 900 python tools/bench_model.py train --stage align --tasks mmu --batch 8 --seqlen 2048 --steps 4 --warmup 2 2>&1 | tail -1 | cut -c1-200
  OMK_LORA_EXT=$e timeout 900 python tools/bench_model.py train --stage finetune --tasks t2i,mmu --batch 2 --seqlen 8192 --steps 3 --warmup 2 2>&1 | tail -1 | cut -c1-200
done | tee gpurun_out/r02_lora_ext.txt
timeout 600 python -m pytest tests/test_lora_ext.py tests/test_stack_decode_train.py tests/test_configs_gpu.py -m gpu -q 2>&1 | tail -3 | tee -a gpurun_out/r02_lora_ext.txt
