#!/bin/bash
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/t
cd $R
for n in test_ops_selscan test_mamba1_module test_bench_contract; do
  timeout 900 python -m pytest tests/$n.py -m gpu -q > gpurun_out/t/$n.log 2>&1
  echo "$n rc=$? $(tail -1 gpurun_out/t/$n.log | cut -c1-150)"
done | tee gpurun_out/r02_gputests_g.txt
for lc in 8 16; do OMK_SELSCAN_LC=$lc timeout 200 python tools/bench_selscan.py 2>&1 | grep "B="; done | tee gpurun_out/r02_selscan_shared2.txt
timeout 900 python bench.py > gpurun_out/r02_bench_c.json 2> gpurun_out/r02_bench_c.err; python - <<PY
import json
j=json.loads(open("gpurun_out/r02_bench_c.json").read().strip().splitlines()[-1])
print("ms/step", j["ms_per_step"], "fwd", j["roofline"]["launch_ms"], j["roofline"]["frac"], "bwd", j["roofline_bwd"]["launch_ms"], j["roofline_bwd"]["frac"])
for k in ("train_1p3b", "train_1p3b_stage2", "selscan_cfg1"): print(k, json.dumps(j[k]))
PY
tail -3 gpurun_out/r02_bench_c.err
