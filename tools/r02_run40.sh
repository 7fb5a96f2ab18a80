#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
timeout 900 python -m pytest tests/test_ops_ssd.py -m gpu -q -k "precise or v6" 2>&1 | tail -2 | tee gpurun_out/r02_precise.txt
for v in 0 1; do echo "PRECISE=$v"; OMK_SSD_PRECISE=$v python tools/bench_scan.py 2>&1 | grep "fwd  B=8"; done | tee -a gpurun_out/r02_precise.txt
OMK_SSD_PRECISE=1 timeout 900 python -m pytest tests/test_configs_gpu.py -m gpu -q -k "cfg2" -s 2>&1 | grep -i "arith\|passed\|failed" | head -5 | tee -a gpurun_out/r02_precise.txt
