#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
run() { echo "== $*"; env "$@" timeout 900 python tools/bench_model.py train --stage finetune --tasks t2i,mmu --batch 2 --seqlen 8192 --steps 3 --warmup 2 2>&1 | tail -1 | cut -c1-330; }
run A=1 | tee gpurun_out/r02_stage2_diag.txt
run OMK_LORA_UP_FUSED=0 | tee -a gpurun_out/r02_stage2_diag.txt
run OMK_LORA_EXT=0 | tee -a gpurun_out/r02_stage2_diag.txt
run OMK_CONV_FWD_TL=32 OMK_CONV_BWD_VAR=2324 | tee -a gpurun_out/r02_stage2_diag.txt
