#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc csv output (one directory per pass) per kernel: mean counter value per dispatch."""
import csv, glob, sys, collections
root = sys.argv[1]
want = sys.argv[2] if len(sys.argv) > 2 else "conv"
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{root}/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if want in k:
            acc[k[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c in sorted(d):
        v = d[c]
        print(f"   {c:32s} n={len(v):3d} mean={sum(v)/len(v):16.1f}")
