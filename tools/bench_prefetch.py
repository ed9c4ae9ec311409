#!/usr/bin/env python3
"""What would prefetching a layer's WEIGHTS into the Infinity Cache buy?  For weight-heavy launches of the forward: time the launch
(a) hot (replayed back to back), (b) after a 1 GiB streaming write (everything cold), (c) after the same flush followed by a read
pass over W and over A only (what a prefetcher running one layer ahead would have done; A is warm inside the graph anyway).
    python tools/bench_prefetch.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from moca_video_amd import lib as L, ops
DEV = "cuda"
torch.manual_seed(0)
flush = torch.empty(1 << 28, dtype=torch.float32, device=DEV)


def timed(fn, pre):
    ts = []
    for _ in range(5):
        pre()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return sorted(ts)[len(ts) // 2]


def case(tag, M, N, K, mode, conv=None, tconv=None, splits=1):
    a = (torch.randn(M, conv[0] if conv else (tconv[0] if tconv else K), device=DEV)).half()
    w = torch.randn(N, K, device=DEV) * K ** -0.5
    pw = ops._finish(w, torch.randn(N, device=DEV), DEV)
    out = torch.empty(M, N, dtype=torch.float16, device=DEV)
    ws = torch.empty(splits * M, N, dtype=torch.float32, device=DEV) if splits > 1 else None
    kw = dict(M=M, mode=mode, splits=splits, splitk_ws=ws)
    if conv: kw["conv"] = conv
    if tconv: kw["tconv"] = tconv
    fn = lambda: ops.gemm(a, pw, out, **kw)
    fn(); fn()
    hot = timed(fn, lambda: None)
    cold = timed(fn, lambda: flush.fill_(1.0))
    pre_w = timed(fn, lambda: (flush.fill_(1.0), pw.w.float().sum(), a.float().sum()))
    pre_a = timed(fn, lambda: (flush.fill_(1.0), a.float().sum()))
    print(f"{tag:44s} hot {hot:7.1f} us   cold {cold:7.1f}   flush + touch A {pre_a:7.1f}   flush + touch W and A {pre_w:7.1f}   (W {pw.w.numel() * 2 / 1e6:.1f} MB)")


case("conv 3x3 M=5120 N=1280 K=11520", 5120, 1280, 11520, L.MOCA_A_CONV3X3, conv=(1280, 10, 16, 10, 16, 1, 0))
case("conv 3x3 M=5120 N=1280 K=23040", 5120, 1280, 23040, L.MOCA_A_CONV3X3, conv=(2560, 10, 16, 10, 16, 1, 0))
case("tconv M=5120 N=1280 K=3840", 5120, 1280, 3840, L.MOCA_A_TCONV3, tconv=(1280, 16, 160))
case("lin M=5120 N=1280 K=5120", 5120, 1280, 5120, L.MOCA_A_LINEAR)
case("conv 3x3 M=1280 N=1280 K=11520 splits=5", 1280, 1280, 11520, L.MOCA_A_CONV3X3, conv=(1280, 5, 8, 5, 8, 1, 0), splits=5)
case("tconv M=1280 N=1280 K=3840 splits=4", 1280, 1280, 3840, L.MOCA_A_TCONV3, tconv=(1280, 16, 40), splits=4)
case("conv 3x3 M=20480 N=640 K=5760", 20480, 640, 5760, L.MOCA_A_CONV3X3, conv=(640, 20, 32, 20, 32, 1, 0))
case("lin M=20480 N=640 K=2560", 20480, 640, 2560, L.MOCA_A_LINEAR)
case("conv 3x3 M=81920 N=320 K=2880", 81920, 320, 2880, L.MOCA_A_CONV3X3, conv=(320, 40, 64, 40, 64, 1, 0))
