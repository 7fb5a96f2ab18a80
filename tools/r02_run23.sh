#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
timeout 600 python -m pytest tests/test_lora_ext.py tests/test_ops_lora_add.py -m gpu -q 2>&1 | tail -2 | tee gpurun_out/r02_lora_up.txt
for f in 0 1; do echo "OMK_LORA_UP_FUSED=$f"; OMK_LORA_UP_FUSED=$f timeout 900 python tools/bench_model.py train --stage align --tasks mmu --batch 8 --seqlen 2048 --steps 4 --warmup 2 2>&1 | tail -1 | cut -c1-200; done | tee -a gpurun_out/r02_lora_up.txt
bash tools/prof_train_cfg.sh align mmu 8 2048
grep "lora_up_bwd\|lora_add" gpurun_out/train_kernel_stats_align_8x2048.csv | cut -c1-160 | tee -a gpurun_out/r02_lora_up.txt
