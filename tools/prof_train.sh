#!/bin/bash
# rocprofv3 kernel statistics of the 1.3B stage-2 step (tools/bench_model.py train); summary -> gpurun_out/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_train -o train -- python $R/tools/bench_model.py train --batch ${1:-8} --steps 2 --warmup 1 > /tmp/train_prof.log 2>&1
tail -1 /tmp/train_prof.log | cut -c1-400
f=$(find /tmp/prof_train -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" $R/gpurun_out/train_kernel_stats.csv
