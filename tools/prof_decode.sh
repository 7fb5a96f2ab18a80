#!/bin/bash
# kernel-trace statistics of the 1.3B decode loop (fp32 and bf16 weights); summaries -> gpurun_out/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
for w in f32 bf16; do
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$w -o dec -- python $R/tools/bench_model.py decode --weights $w > /tmp/dec_$w.log 2>&1
  grep ms_per_token /tmp/dec_$w.log
  tail -3 /tmp/dec_$w.log | cut -c1-300
  f=$(find /tmp/prof_$w -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" $R/gpurun_out/decode_kernel_stats_$w.csv
done
ls -la $R/gpurun_out/ /tmp/prof_f32 2>&1 | head -20
