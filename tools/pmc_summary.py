#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output (one directory per pass) per kernel: mean counter value per dispatch."""
import csv, glob, os, sys, collections
root = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(root, "*", "*", "*counter_collection.csv")):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row["Kernel_Name"].split("(")[0][:60]
            agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in agg.items():
    n = max(len(v) for v in d.values())
    if n < 3: continue
    print(f"== {k}  ({n} dispatches)")
    for c, v in sorted(d.items()):
        v = sorted(v)
        print(f"   {c:28s} mean {sum(v)/len(v):16.1f}  median {v[len(v)//2]:16.1f}")
