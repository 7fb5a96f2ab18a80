#!/bin/bash
# rocprofv3 kernel statistics of the 1.3B training step: $1 = stage (align | finetune), $2 = tasks (mmu | t2i,mmu), $3 = batch, $4 = seqlen
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
tag=${1:-align}_${3:-8}x${4:-2048}
timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_train_$tag -o train -- python $R/tools/bench_model.py train --stage ${1:-align} --tasks ${2:-mmu} --batch ${3:-8} --seqlen ${4:-2048} --steps 2 --warmup 1 > /tmp/train_prof_$tag.log 2>&1
tail -1 /tmp/train_prof_$tag.log | cut -c1-300
f=$(find /tmp/prof_train_$tag -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" $R/gpurun_out/train_kernel_stats_$tag.csv
