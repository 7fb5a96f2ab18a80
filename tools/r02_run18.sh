#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
for v in "" 2324 2328 2644 2648 21288 4324 4644 1648; do echo "VAR=$v"; OMK_CONV_BWD_VAR=$v timeout 120 python tools/bench_conv.py 2>&1 | tail -1; done | tee gpurun_out/r02_conv_bwd_var.txt
