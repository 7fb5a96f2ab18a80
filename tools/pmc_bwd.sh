#!/bin/bash
# kernel statistics + HBM-traffic PMC passes (separate passes, --kernel-trace only) of the backward scan launch
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_bwd
mkdir -p $OUT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o stats -- python $R/tools/prof_bwd.py > $OUT/run_stats.log 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT -o $C -- python $R/tools/prof_bwd.py > $OUT/run_$C.log 2>&1
done
python - <<PY
import csv, glob, collections
out = "$OUT"
for f in glob.glob(out + "/**/*stats_kernel_stats.csv", recursive=True):
    print(open(f).read()[:3000])
for C in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(out + f"/**/{C}_counter_collection.csv", recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == C:
                acc[r["Kernel_Name"][:70]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            print(C, k, "n=%d mean=%.1f KB" % (len(v), sum(v) / len(v)))
PY
