#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
timeout 900 python -m pytest tests/test_ops_ssd.py tests/test_mamba2_module.py tests/test_configs_gpu.py -m gpu -q 2>&1 | tail -2 | tee gpurun_out/r02_bufops.txt
for i in 1 2; do python tools/bench_scan.py --bwd 2>&1 | grep "B=8 L=4096" | tee -a gpurun_out/r02_bufops.txt; done
timeout 600 python bench.py --steps 20 --warmup 5 --min-seconds 2 --no-cpu-baseline --no-train-1p3b --no-selscan-cfg1 > /tmp/b.json 2>/dev/null
python - <<PY | tee -a gpurun_out/r02_bufops.txt
import json
j=json.loads(open("/tmp/b.json").read().strip().splitlines()[-1])
print("ms/step", j["ms_per_step"], "fwd", j["roofline"]["launch_ms"], j["roofline"]["frac"], "bwd", j["roofline_bwd"]["launch_ms"], j["roofline_bwd"]["frac"])
PY
