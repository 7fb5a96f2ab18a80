#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
timeout 900 python -m pytest tests/test_ops_selscan.py tests/test_mamba1_module.py -m gpu -q 2>&1 | tail -2
for cfg in "16 16" "8 8" "8 16" "16 8"; do set -- $cfg; echo "NW=$1 NB=$2"; OMK_SELSCAN_BWD_NW=$1 OMK_SELSCAN_BWD_NB=$2 timeout 300 python tools/bench_selscan.py --bwd 2>&1 | grep "bwd"; done | tee gpurun_out/r02_selscan_bwd3.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ss -o ss -- python $R/tools/bench_selscan.py --bwd > /tmp/ss.log 2>&1
f=$(find /tmp/prof_ss -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -8 "$f" | cut -c1-220 | tee -a $R/gpurun_out/r02_selscan_bwd3.txt
