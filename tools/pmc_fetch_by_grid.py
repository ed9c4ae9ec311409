#!/usr/bin/env python3
"""FETCH_SIZE per (kernel, grid size) of one rocprofv3 --pmc FETCH_SIZE pass over bench.py: which launches over-fetch.
usage: pmc_fetch_by_grid.py <dir with *counter_collection.csv> <forwards>"""
import csv, glob, sys, collections, re
root, nfwd = sys.argv[1], float(sys.argv[2])
agg = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(f"{root}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if not ("gemm" in n or "attention" in n or "gn_" in n or "splitk" in n or "concat" in n):
            continue
        m = re.search(r"(gemm_\w+?_kernel(?:<[^>]*>|I\w+?EE)?|attention_\w+|gn_\w+_kernel|splitk_reduce|concat_gstat)", n)
        key = ((m.group(1) if m else n[:40]), r.get("Grid_Size", "?"))
        a = agg[key]
        a[0] += float(r["Counter_Value"]); a[1] += 1
rows = sorted(agg.items(), key=lambda kv: -kv[1][0])
tot = sum(v[0] for v in agg.values())
print(f"total FETCH_SIZE x2 = {2*tot*1024/nfwd/1e9:.2f} GB per forward")
print(f"{'kernel':44s} {'grid':>9s} {'launches/fwd':>12s} {'GB/fwd (x2)':>12s} {'MB each':>9s}")
for (k, g), (v, c) in rows[:45]:
    print(f"{k:44s} {g:>9s} {c/nfwd:12.1f} {2*v*1024/nfwd/1e9:12.3f} {2*v*1024/c/1e6:9.1f}")
