#!/bin/bash
# round 2, GPU call 3: GPU suite one file per process (an abort in one file must not hide the others) + v5 vs v3 scan timing
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/t
cd $R
for f in tests/test_*.py; do
  n=$(basename $f .py)
  timeout 900 python -m pytest $f -m gpu -q -x > gpurun_out/t/$n.log 2>&1
  echo "$n rc=$? $(tail -1 gpurun_out/t/$n.log | cut -c1-150)"
done | tee gpurun_out/r02_gputests_c.txt
echo "--- v5 (two waves per head)"; timeout 300 python tools/bench_scan.py --bwd 2>&1 | tail -8 | tee gpurun_out/r02_scan_v5.txt
echo "--- v3 (round 1 kernels)";   OMK_SSD_V5=0 timeout 300 python tools/bench_scan.py --bwd 2>&1 | tail -8 | tee gpurun_out/r02_scan_v3.txt
