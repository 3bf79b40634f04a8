"""Average of one PMC counter per kernel from a rocprofv3 --pmc run (counter_collection.csv).
usage: pmc_summary.py <dir> [kernel-name substring ...]"""
import csv, glob, os, sys, collections
d = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        d[row["Kernel_Name"].split("(")[0][:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, cs in sorted(d.items()):
    if len(sys.argv) > 2 and not any(s in k for s in sys.argv[2:]):
        continue
    for c, v in cs.items():
        print(f"{k:62s} {c:28s} n={len(v):4d} avg={sum(v) / len(v):.4g}")
