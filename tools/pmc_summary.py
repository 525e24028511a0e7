"""Summarise a rocprofv3 --pmc run: per kernel, dispatches and the per-dispatch mean of every counter.
usage: python tools/pmc_summary.py <dir with *_counter_collection.csv> [out.csv]"""
import csv, glob, os, sys
from collections import defaultdict
d = sys.argv[1]
files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
acc = defaultdict(lambda: defaultdict(float)); nd = defaultdict(set)
for f in files:
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); nd[k].add(r["Dispatch_Id"])
rows = []
for k in sorted(acc):
    n = len(nd[k])
    for c in sorted(acc[k]):
        rows.append((k, n, c, acc[k][c] / n))
out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
out.write("kernel,dispatches,counter,mean_per_dispatch\n")
for r in rows:
    out.write("%s,%d,%s,%.6g\n" % r)
