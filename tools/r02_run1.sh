#!/bin/bash
# round 2, GPU call 1: full -m gpu suite, default bench line, backward-scan kernel stats + PMC traffic, b3 ablations
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r02_gputests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02_gputests.log
tail -5 gpurun_out/r02_gputests.log
timeout 600 python bench.py > gpurun_out/r02_bench_a.json 2> gpurun_out/r02_bench_a.err; tail -c 3000 gpurun_out/r02_bench_a.json
bash tools/pmc_bwd.sh > gpurun_out/r02_pmc_bwd.txt 2>&1; tail -40 gpurun_out/r02_pmc_bwd.txt
cd $R
for m in 0 1 2 4 8 16 32 40 63; do
  echo "ablate_b=$m: $(OMK_ABLATE_B=$m timeout 120 python tools/with_lib.py omnimamba_amd/lib/libomnimamba_hip_prof.so tools/ab_bwd.py 8 4096 2>&1 | tail -1)"
done | tee gpurun_out/r02_ablate_b.txt
