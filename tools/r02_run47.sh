#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
timeout 900 python -m pytest tests/test_ops_norms.py tests/test_mamba2_module.py tests/test_stack_decode_train.py -m gpu -q 2>&1 | tail -1 | tee gpurun_out/r02_norms.txt
timeout 600 python bench.py --steps 20 --warmup 5 --min-seconds 2 --no-cpu-baseline --no-train-1p3b --no-selscan-cfg1 > /tmp/b.json 2>/dev/null
python - <<PY | tee -a gpurun_out/r02_norms.txt
import json
j=json.loads(open("/tmp/b.json").read().strip().splitlines()[-1])
print("ms/step", j["ms_per_step"], "fwd", j["roofline"]["launch_ms"], j["roofline"]["frac"], "bwd", j["roofline_bwd"]["launch_ms"], j["roofline_bwd"]["frac"])
PY
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b -o b -- python $R/bench.py --steps 5 --warmup 2 --min-seconds 0 --no-cpu-baseline --no-train-1p3b --no-selscan-cfg1 > /tmp/pb.log 2>&1
f=$(find /tmp/prof_b -name "*kernel_stats.csv" | head -1); grep "norm_\|conv1d" "$f" | cut -c1-150 | tee -a $R/gpurun_out/r02_norms.txt
timeout 900 python tools/bench_model.py train --stage align --tasks mmu --batch 8 --seqlen 2048 --steps 4 --warmup 2 2>&1 | tail -1 | cut -c1-200 | tee -a $R/gpurun_out/r02_norms.txt
