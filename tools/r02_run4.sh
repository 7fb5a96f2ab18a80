#!/bin/bash
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/t
cd $R
for n in test_configs_gpu test_ops_selscan test_ops_ssd test_mamba2_module test_golden; do
  timeout 900 python -m pytest tests/$n.py -m gpu -q > gpurun_out/t/$n.log 2>&1
  echo "$n rc=$? $(tail -1 gpurun_out/t/$n.log | cut -c1-150)"
done | tee gpurun_out/r02_gputests_d.txt
grep -E "^E |FAILED" gpurun_out/t/test_configs_gpu.log | head -20
timeout 300 python bench.py --no-train-1p3b --no-cpu-baseline --min-seconds 1 > gpurun_out/r02_bench_b.json 2>/dev/null; python - <<PY
import json
j=json.loads(open("gpurun_out/r02_bench_b.json").read().strip().splitlines()[-1])
print("ms/step", j["ms_per_step"], "fwd", j["roofline"]["launch_ms"], j["roofline"]["frac"], "bwd", j["roofline_bwd"]["launch_ms"], j["roofline_bwd"]["frac"])
print(json.dumps(j["selscan_cfg1"]))
PY
timeout 300 python tools/bench_scan.py --bwd 2>&1 | tail -6
