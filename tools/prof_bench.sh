#!/bin/bash
# rocprofv3 kernel statistics of the default bench.py run + a plain bench.py line; summaries -> gpurun_out/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -o bench -- python $R/bench.py > /tmp/bench_prof.log 2>&1
tail -1 /tmp/bench_prof.log | cut -c1-400
f=$(find /tmp/prof_bench -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" $R/gpurun_out/bench_kernel_stats.csv
cd $R && python bench.py 2>/dev/null | tail -1 > $R/gpurun_out/bench.json
cat $R/gpurun_out/bench.json
