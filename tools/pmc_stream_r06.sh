#!/bin/bash
# Round 6: HBM traffic and SQ counters of the streaming helpers (tools/bench_stream.py: conv1d fwd / bwd, gated RMSNorm fwd / bwd at the cfg 2 shape),
# separate --pmc passes (--kernel-trace only).  -> gpurun_out/r06/pmc_stream/counters.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r06/pmc_stream
mkdir -p $OUT
CMD="python $R/tools/bench_stream.py"
(cd $R && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o stats -- $CMD > $OUT/run_stats.log 2>&1)
i=0
for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU"; do
  i=$((i+1))
  (cd $R && timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT -o p$i -- $CMD > $OUT/run$i.log 2>&1)
done
python - <<PY > $OUT/counters.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/**/p*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "omk::" in r["Kernel_Name"]:
            acc[r["Kernel_Name"][:100]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob("$OUT/**/stats_kernel_stats.csv", recursive=True):
    print("kernel statistics (rocprofv3 --kernel-trace --stats):")
    for r in sorted(csv.DictReader(open(f)), key=lambda r: -float(r["TotalDurationNs"])):
        if "omk::" in r["Name"]:
            print(f"  {int(r['Calls']):4d} calls  avg {float(r['AverageNs'])/1e3:9.1f} us  {r['Name'][:110]}")
print()
print("per-dispatch counter averages (FETCH_SIZE / WRITE_SIZE in KB as reported; FETCH_SIZE counts 64 B per 128-B request on gfx950: x2).")
print("algorithmic bytes: conv fwd 285 MB in + 285 out; conv bwd 570 in + 285 out; norm fwd 537 in + 268 out; norm bwd 805 in + 537 out (+ partial rows)")
for k in sorted(acc):
    print(k)
    for c, v in sorted(acc[k].items()):
        print(f"    {c:28s} {sum(v)/len(v):18.1f}   (n={len(v)})")
PY
cat $OUT/counters.txt
