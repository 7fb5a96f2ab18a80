#!/bin/bash
# round 2: full GPU suite in one process, default bench line, kernel statistics of the bench, backward traffic counters
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r02_gputests_full.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02_gputests_full.log
tail -4 gpurun_out/r02_gputests_full.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 900 python bench.py > gpurun_out/r02_bench_final.json 2> gpurun_out/r02_bench_final.err; tail -c 2500 gpurun_out/r02_bench_final.json
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -o bench -- python $R/bench.py --steps 5 --warmup 2 --min-seconds 0 --no-cpu-baseline --no-train-1p3b --no-selscan-cfg1 > /tmp/bench_prof.log 2>&1
f=$(find /tmp/prof_bench -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $R/gpurun_out/r02_bench_kernel_stats.csv && head -30 $R/gpurun_out/r02_bench_kernel_stats.csv | cut -c1-160
cd $R && bash tools/pmc_bwd.sh > gpurun_out/r02_pmc_bwd_final.txt 2>&1; grep -E "^FETCH|^WRITE" gpurun_out/r02_pmc_bwd_final.txt | grep omk
