#!/usr/bin/env python3
"""Idle time between consecutive kernels of the replayed UNet graph, from a rocprofv3 --kernel-trace CSV: the last complete
forward (from one ncthw_to_nhwc launch to the next), gaps grouped by the kernel that FOLLOWS the gap."""
import csv, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda t: t[0])
starts = [i for i, k in enumerate(ks) if "ncthw_to_nhwc" in k[2]]
a, b = starts[-2], starts[-1]
fw = ks[a:b]


def short(n):
    m = re.search(r"(gemm_\w+?_kernel|attention_v4|attention_kernel|temporal_attention|gn_\w+?_kernel|layernorm|splitk_reduce|concat|silu|timestep|nhwc|ncthw|cfg|ddim)", n)
    return m.group(1) if m else n[:30]


busy = sum(e - s for s, e, _ in fw)
span = fw[-1][1] - fw[0][0]
print(f"kernels in one forward: {len(fw)}   span {span/1e6:.3f} ms   kernel time {busy/1e6:.3f} ms   idle {(span-busy)/1e6:.3f} ms")
gaps = collections.defaultdict(lambda: [0, 0])
dur = collections.defaultdict(lambda: [0, 0])
for (s0, e0, n0), (s1, e1, n1) in zip(fw, fw[1:]):
    g = gaps[short(n1)]
    g[0] += max(0, s1 - e0); g[1] += 1
for s, e, n in fw:
    d = dur[short(n)]
    d[0] += e - s; d[1] += 1
print(f"{'kernel (after the gap)':34s} {'launches':>8s} {'gap total us':>13s} {'gap each us':>12s} {'kernel total us':>16s} {'each us':>9s}")
for k, (t, c) in sorted(gaps.items(), key=lambda kv: -kv[1][0]):
    print(f"{k:34s} {c:8d} {t/1e3:13.1f} {t/1e3/c:12.2f} {dur[k][0]/1e3:16.1f} {dur[k][0]/1e3/max(1,dur[k][1]):9.1f}")
