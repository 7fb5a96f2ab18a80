#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
OMK_SSD_V6=1 timeout 900 python -m pytest tests/test_ops_ssd.py tests/test_mamba2_module.py tests/test_configs_gpu.py -m gpu -q 2>&1 | tail -2 | tee gpurun_out/r02_v6.txt
for v in 0 1 0 1; do
  OMK_SSD_V6=$v timeout 600 python bench.py --steps 20 --warmup 5 --min-seconds 2 --no-cpu-baseline --no-train-1p3b --no-selscan-cfg1 > /tmp/b.json 2>/dev/null
  python - <<PY | tee -a gpurun_out/r02_v6.txt
import json
j=json.loads(open("/tmp/b.json").read().strip().splitlines()[-1])
print("V6=$v ms/step", j["ms_per_step"], "fwd", j["roofline"]["launch_ms"], j["roofline"]["frac"], "bwd", j["roofline_bwd"]["launch_ms"], j["roofline_bwd"]["frac"])
PY
done
cd /tmp && export TMPDIR=/tmp
OMK_SSD_V6=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_v6 -o v6 -- python $R/bench.py --steps 5 --warmup 2 --min-seconds 0 --no-cpu-baseline --no-train-1p3b --no-selscan-cfg1 > /tmp/v6.log 2>&1
f=$(find /tmp/prof_v6 -name "*kernel_stats.csv" | head -1); grep "ssd_mfma" "$f" | cut -c1-150 | tee -a $R/gpurun_out/r02_v6.txt
