#!/usr/bin/env python3
"""The 1280-channel level's GEMMs (M = 1280 rows at B=2: too few 256-row tiles to fill 256 CUs) under the split-k variants:
partials + splitk_reduce at the plan's split factor and a smaller one, the 256-row and 128-row kernels without split
(the one-launch form with the last-arriving block summing the partials: profiles/r03_ab_splitk_in_kernel.txt).
    python tools/bench_splitk.py > gpurun_out/bench_splitk.txt"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from moca_video_amd import ops, lib as L

DEV = "cuda"
F, H, W = 32, 5, 8
M = F * H * W


def timeit(fn, iters=20):
    """GPU time per call: 20 calls captured in one graph (no host launch cost between them), 10 replays"""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (10 * iters)


def case(name, a, pw, kw, flops, variants):
    res = torch.randn(M, pw.N, device=DEV).half()
    out = torch.empty(M, pw.N, device=DEV, dtype=torch.float16)
    line = f"{name:34s}"
    for label, splits, small in variants:
        ws = torch.empty(max(splits, 1) * M * pw.N, device=DEV, dtype=torch.float32)
        fn = lambda: ops.gemm(a, pw, out, M=M, residual=res, splits=splits, splitk_ws=ws if splits > 1 else None,
                              force_small=small, **kw)
        us = timeit(fn)
        line += f"  {label}: {us:6.1f} us ({flops / us / 1e6:4.0f} TF/s)"
    print(line, flush=True)


def variants(s):
    return [(f"s{s}", s, False), ("128-row s1", 1, True), ("128-row s2", 2, True), (f"s{max(s - 2, 2)}", max(s - 2, 2), False), ("256-row s1", 1, False)]


c = 1280
x = torch.randn(M, c, device=DEV).half()
w = torch.randn(c, c, 3, 1, 1, device=DEV) * (3 * c) ** -0.5
case("tconv 1280->1280 K=3840", x, ops.pack_tconv3(w, torch.zeros(c, device=DEV)), dict(mode=L.MOCA_A_TCONV3, tconv=(c, 16, H * W)), 2.0 * M * c * 3 * c, variants(4))
w = torch.randn(c, c, 3, 3, device=DEV) * (9 * c) ** -0.5
case("conv3x3 1280->1280 K=11520", x, ops.pack_conv3x3(w, torch.zeros(c, device=DEV)), dict(mode=L.MOCA_A_CONV3X3, conv=(c, H, W, H, W, 1, 0)), 2.0 * M * c * 9 * c, variants(5))
x2 = torch.randn(M, 2 * c, device=DEV).half()
w = torch.randn(c, 2 * c, 3, 3, device=DEV) * (18 * c) ** -0.5
case("conv3x3 2560->1280 K=23040", x2, ops.pack_conv3x3(w, torch.zeros(c, device=DEV)), dict(mode=L.MOCA_A_CONV3X3, conv=(2 * c, H, W, H, W, 1, 0)), 2.0 * M * c * 18 * c, variants(5))
w = torch.randn(c, 2 * c, device=DEV) * (2 * c) ** -0.5
case("linear 2560->1280", x2, ops.pack_linear(w, torch.zeros(c, device=DEV)), {}, 2.0 * M * c * 2 * c, variants(4))
w = torch.randn(c, 4 * c, device=DEV) * (4 * c) ** -0.5
x4 = torch.randn(M, 4 * c, device=DEV).half()
case("linear 5120->1280 (ff out)", x4, ops.pack_linear(w, torch.zeros(c, device=DEV)), {}, 2.0 * M * c * 4 * c, variants(4))
w = torch.randn(c, c, device=DEV) * c ** -0.5
case("linear 1280->1280", x, ops.pack_linear(w, torch.zeros(c, device=DEV)), {}, 2.0 * M * c * c, variants(2))
