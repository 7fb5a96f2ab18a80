#!/bin/bash
# HBM-traffic PMC passes (FETCH_SIZE, WRITE_SIZE: one counter per pass, --kernel-trace only) + kernel statistics of ANY command:
#   tools/pmc_any.sh <tag> python tools/bench_selscan.py --bwd
# prints per kernel: average duration, FETCH_SIZE x 2 (gfx950: 128-byte requests are tallied at 64 bytes) and WRITE_SIZE in MB
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1; shift
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
(cd $R && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o stats -- "$@" > $OUT/run_stats.log 2>&1)
for C in FETCH_SIZE WRITE_SIZE; do
  (cd $R && timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT -o $C -- "$@" > $OUT/run_$C.log 2>&1)
done
python - <<PY
import csv, glob, collections
out = "$OUT"
dur = {}
for f in glob.glob(out + "/**/*stats_kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Name"][:90]] = (float(r["AverageNs"]) / 1e3, int(r["Calls"]))
tr = collections.defaultdict(dict)
for C in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(out + f"/**/{C}_counter_collection.csv", recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == C:
                acc[r["Kernel_Name"][:90]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            tr[k][C] = sum(v) / len(v)
print("%-92s %9s %6s %10s %10s" % ("kernel", "avg us", "calls", "read MB", "write MB"))
for k, (us, n) in sorted(dur.items(), key=lambda kv: -kv[1][0] * kv[1][1])[:14]:
    t = tr.get(k, {})
    print("%-92s %9.1f %6d %10.1f %10.1f" % (k, us, n, 2 * t.get("FETCH_SIZE", 0) / 1024, t.get("WRITE_SIZE", 0) / 1024))
PY
