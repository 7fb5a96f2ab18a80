#!/bin/bash
# kernel-trace statistics of the 1.3B decode loop at batch $1 (fp32 weights)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b$1 -o dec -- python $R/tools/bench_model.py decode --batch $1 > /tmp/dec_b$1.log 2>&1
grep ms_per_token /tmp/dec_b$1.log
f=$(find /tmp/prof_b$1 -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" $R/gpurun_out/decode_kernel_stats_b$1.csv
