#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
bash tools/prof_train_cfg.sh align mmu 8 2048
bash tools/prof_train_cfg.sh finetune t2i,mmu 2 8192
