#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
timeout 900 python -m pytest tests/test_ops_ssd.py tests/test_configs_gpu.py -m gpu -q 2>&1 | tail -2
timeout 600 python bench.py --steps 20 --warmup 5 --min-seconds 2 --no-cpu-baseline --no-train-1p3b --no-selscan-cfg1 > gpurun_out/r02_bench_dxnolo.json 2>/dev/null
python - <<PY
import json
j=json.loads(open("gpurun_out/r02_bench_dxnolo.json").read().strip().splitlines()[-1])
print("ms/step", j["ms_per_step"], "fwd", j["roofline"]["launch_ms"], "bwd", j["roofline_bwd"]["launch_ms"], j["roofline_bwd"]["frac"])
PY
