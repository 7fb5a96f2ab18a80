#!/bin/bash
# round 2, call 9: chunked Mamba-1 backward -- GPU tests, forward / backward timings, NU=2 variant of the forward
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/t
cd $R
for n in test_ops_selscan test_mamba1_module; do
  timeout 900 python -m pytest tests/$n.py -m gpu -q > gpurun_out/t/$n.log 2>&1
  echo "$n rc=$? $(tail -1 gpurun_out/t/$n.log | cut -c1-150)"
done | tee gpurun_out/r02_gputests_h.txt
timeout 300 python tools/bench_selscan.py --bwd 2>&1 | grep "B=\|bwd" | tee gpurun_out/r02_selscan_bwd.txt
for o in 1 2 4; do echo "OCT=$o"; OMK_SELSCAN_BWD_OCT=$o timeout 300 python tools/bench_selscan.py --bwd 2>&1 | grep "bwd"; done | tee -a gpurun_out/r02_selscan_bwd.txt
echo "SEQ (round-1 backward)"; OMK_SELSCAN_SEQ=1 timeout 600 python tools/bench_selscan.py --bwd 2>&1 | grep "B=\|bwd" | head -4 | tee -a gpurun_out/r02_selscan_bwd.txt
echo "NU=2"; OMK_SELSCAN_NU=2 timeout 200 python tools/bench_selscan.py 2>&1 | grep "B=" | tee -a gpurun_out/r02_selscan_bwd.txt
