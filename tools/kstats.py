"""Print the omk:: rows of a rocprofv3 *_kernel_stats.csv: calls, average us, name (template arguments kept)."""
import csv
import glob
import sys

for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if "omk::" in r["Name"]]
    for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"])):
        print(f"{int(r['Calls']):5d} {float(r['AverageNs']) / 1e3:10.1f} us  {r['Name'][:110]}")
