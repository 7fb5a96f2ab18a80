#!/bin/bash
# chunked Mamba-1 backward: where does the time go (developer ablation bits) + kernel trace
R=$GRAFT_REPO_ROOT
cd $R
for d in 0 1 2 3 4 7; do echo "DBG=$d"; OMK_SELSCAN_BWD_DBG=$d timeout 300 python tools/bench_selscan.py --bwd 2>&1 | grep "bwd" | head -2; done | tee gpurun_out/r02_selscan_bwd_ablate.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ss -o ss -- python $R/tools/bench_selscan.py --bwd > /tmp/ss.log 2>&1
f=$(find /tmp/prof_ss -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -8 "$f" | cut -c1-200 | tee -a $R/gpurun_out/r02_selscan_bwd_ablate.txt
