#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for nb in 1 8; do for v in in_plain in_res in_lora out_plain out_gate; do
  NB=$nb VARIANT=$v timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pnl_${nb}_$v -o p -- python $R/tools/probe_nl_batched.py > /dev/null 2>&1
  f=$(find /tmp/pnl_${nb}_$v -name "*kernel_stats.csv" | head -1)
  python3 - "$f" "$nb" "$v" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "norm_linear" in r["Name"]:
        print(f"NB={sys.argv[2]} {sys.argv[3]:10s} {float(r['AverageNs'])/1e3:7.2f} us  ({r['Calls']} calls)  {r['Name'][10:70]}")
PY
done; done
