#!/bin/bash
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/t
cd $R
for n in test_configs_gpu test_fused_ce test_ops_selscan test_ops_lora_add test_stack_decode_train; do
  timeout 900 python -m pytest tests/$n.py -m gpu -q -s > gpurun_out/t/$n.log 2>&1
  echo "$n rc=$? $(tail -1 gpurun_out/t/$n.log | cut -c1-150)"
done | tee gpurun_out/r02_gputests_e.txt
grep -E "^E |FAILED|arithmetic part" gpurun_out/t/test_configs_gpu.log | cut -c1-900 | head -20
for lc in 8 16; do OMK_SELSCAN_LC=$lc timeout 200 python tools/bench_selscan.py 2>&1 | grep "B="; done | tee gpurun_out/r02_selscan.txt
timeout 400 python bench.py --no-cpu-baseline --no-selscan-cfg1 --min-seconds 1 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('train_1p3b', json.dumps(j['train_1p3b']))" | tee gpurun_out/r02_train_1p3b_fusedce.txt
