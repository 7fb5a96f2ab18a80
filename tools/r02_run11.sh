#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
timeout 900 python -m pytest tests/test_ops_selscan.py -m gpu -q 2>&1 | tail -2
for o in auto 1 2 4; do echo "OCT=$o"; if [ $o = auto ]; then unset OMK_SELSCAN_BWD_OCT; else export OMK_SELSCAN_BWD_OCT=$o; fi; timeout 300 python tools/bench_selscan.py --bwd 2>&1 | grep "bwd"; done | tee gpurun_out/r02_selscan_bwd2.txt
unset OMK_SELSCAN_BWD_OCT
for d in 2 4 6; do echo "DBG=$d"; OMK_SELSCAN_BWD_DBG=$d timeout 300 python tools/bench_selscan.py --bwd 2>&1 | grep "bwd" | head -3; done | tee -a gpurun_out/r02_selscan_bwd2.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ss -o ss -- python $R/tools/bench_selscan.py --bwd > /tmp/ss.log 2>&1
f=$(find /tmp/prof_ss -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -12 "$f" | cut -c1-220 | tee -a $R/gpurun_out/r02_selscan_bwd2.txt
