#!/usr/bin/env python3
"""Where the replayed UNet graph loses time against the same launches replayed hot in isolation.
Inputs: two rocprofv3 --kernel-trace CSVs -- (1) `bench.py --steps 3 ...` (the graph), (2) `tools/plan_profile.py 2` (every
recorded launch 1 + 5 times back to back FROM ITS TRUE OPERANDS: snapshot / restore, see plan_profile.py).  Kernels are keyed by (name, grid, workgroup); for the
graph the LAST complete forward is taken, for the isolated run the mean over the replays of a key.
    python tools/ingraph_vs_hot.py graph_kernel_trace.csv hot_kernel_trace.csv"""
import collections
import csv
import re
import sys


def load(path):
    rows = list(csv.DictReader(open(path)))
    out = []
    for r in rows:
        grid = tuple(int(r.get(k, 1) or 1) for k in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z")) if "Grid_Size_X" in r else (int(r.get("Grid_Size", 0)),)
        wg = int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 0)) or 0)
        out.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], grid, wg))
    out.sort(key=lambda t: t[0])
    return out


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    return n[:70]


g, h = load(sys.argv[1]), load(sys.argv[2])
starts = [i for i, k in enumerate(g) if "ncthw_to_nhwc" in k[2]]
fw = g[starts[-2]:starts[-1]]
span = fw[-1][1] - fw[0][0]
busy = sum(e - s for s, e, *_ in fw)
print(f"graph: {len(fw)} kernels in the last forward, span {span / 1e6:.3f} ms, kernel time {busy / 1e6:.3f} ms, idle {(span - busy) / 1e6:.3f} ms")
gk = collections.defaultdict(lambda: [0, 0])
for s, e, n, grid, wg in fw:
    k = (short(n), grid, wg)
    gk[k][0] += e - s
    gk[k][1] += 1
# isolated run (tools/plan_profile.py, true-operand mode): per key, every step instance contributes 1 in-sequence launch (the eager
# pass that produces the operand snapshots) + 1 warm-up + REP timed launches from restored operands -> the MEDIAN duration of a key is
# an isolated, warm, true-operand launch
hd = collections.defaultdict(list)
for s, e, n, grid, wg in h:
    hd[(short(n), grid, wg)].append(e - s)
hk = {k: [sorted(v)[len(v) // 2], 1] for k, v in hd.items()}
rows = []
tot_g = tot_h = 0.0
missing = 0
for k, (t, c) in gk.items():
    if k not in hk:
        missing += t
        continue
    hot = hk[k][0] / hk[k][1] * c
    rows.append((t - hot, t, hot, c, k))
    tot_g += t
    tot_h += hot
print(f"matched kernel time: graph {tot_g / 1e6:.3f} ms vs hot-isolated {tot_h / 1e6:.3f} ms (+{(tot_g - tot_h) / 1e6:.3f} ms); unmatched in graph {missing / 1e6:.3f} ms")
print(f"{'delta_us':>9s} {'graph_us':>9s} {'hot_us':>9s} {'n':>4s} {'ratio':>6s}  kernel / grid / wg")
for d, t, hot, c, k in sorted(rows, key=lambda r: -r[1]):
    print(f"{d / 1e3:9.1f} {t / 1e3:9.1f} {hot / 1e3:9.1f} {c:4d} {t / max(hot, 1):6.2f}  {k[0]} {k[1]} {k[2]}")
