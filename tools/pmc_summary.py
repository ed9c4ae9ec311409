#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs: per kernel name, mean counter value per dispatch."""
import csv, glob, sys, collections
root = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    if not any(s in k for s in sys.argv[2:] or [""]):
        continue
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:28s} n={len(v):4d} mean={sum(v)/len(v):16.1f}")
