#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
timeout 600 python -m pytest tests/test_mamba2_module.py tests/test_stack_decode_train.py -m gpu -q 2>&1 | tail -2
timeout 900 python tools/bench_model.py train --stage align --tasks mmu --batch 8 --seqlen 2048 --steps 4 --warmup 2 2>&1 | tail -1 | cut -c1-260 | tee gpurun_out/r02_train_wo.txt
timeout 900 python tools/bench_model.py train --stage align --tasks mmu --batch 16 --seqlen 2048 --steps 3 --warmup 2 2>&1 | tail -1 | cut -c1-260 | tee -a gpurun_out/r02_train_wo.txt
timeout 900 python tools/bench_model.py train --stage finetune --tasks t2i,mmu --batch 2 --seqlen 8192 --steps 3 --warmup 2 2>&1 | tail -1 | cut -c1-260 | tee -a gpurun_out/r02_train_wo.txt
