#!/bin/bash
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/t
cd $R
for n in test_ops_selscan test_ops_ssd test_mamba2_module test_golden test_reference_fixtures; do
  timeout 900 python -m pytest tests/$n.py -m gpu -q > gpurun_out/t/$n.log 2>&1
  echo "$n rc=$? $(tail -1 gpurun_out/t/$n.log | cut -c1-150)"
done | tee gpurun_out/r02_gputests_f.txt
for lc in 8 16; do OMK_SELSCAN_LC=$lc timeout 200 python tools/bench_selscan.py 2>&1 | grep "B="; done | tee gpurun_out/r02_selscan_share.txt
timeout 300 python tools/bench_scan.py --bwd 2>&1 | tail -6 | tee gpurun_out/r02_scan_g.txt
timeout 300 python bench.py --no-train-1p3b --no-cpu-baseline --no-selscan-cfg1 --min-seconds 2 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms/step', j['ms_per_step'], 'fwd', j['roofline']['launch_ms'], j['roofline']['frac'], 'bwd', j['roofline_bwd']['launch_ms'], j['roofline_bwd']['frac'])"
