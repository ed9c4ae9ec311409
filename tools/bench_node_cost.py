#!/usr/bin/env python3
"""Fixed cost of one kernel node in a replayed graph: chains of N dependent launches of (a) a 1-block no-op-sized kernel, (b) a
256-block kernel that touches 1 KB per block, (c) the smallest real GEMM, captured in one graph, timed per node with HIP events."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from moca_video_amd import ops, lib as L

DEV = "cuda"
N = 200


def per_node(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(N):
            fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (10 * N)


a1 = torch.zeros(64, device=DEV)
print(f"torch add_ on 64 floats (1 block)           {per_node(lambda: a1.add_(1.0)):6.2f} us per node")
a2 = torch.zeros(256 * 256, device=DEV)
print(f"torch add_ on 64 Ki floats (256 blocks)     {per_node(lambda: a2.add_(1.0)):6.2f} us per node")
a3 = torch.zeros(16 << 20, device=DEV)
print(f"torch add_ on 16 Mi floats (128 MB traffic) {per_node(lambda: a3.add_(1.0)):6.2f} us per node")
for M, N_, K in ((256, 128, 64), (5120, 1280, 64), (5120, 1280, 320), (5120, 1280, 1280), (5120, 1280, 2560), (20480, 640, 640), (81920, 320, 320), (81920, 320, 1280)):
    x = torch.randn(M, K, device=DEV).half()
    pw = ops.pack_linear(torch.randn(N_, K, device=DEV) * K ** -0.5, torch.zeros(N_, device=DEV))
    out = torch.empty(M, pw.N, device=DEV, dtype=torch.float16)
    us = per_node(lambda: ops.gemm(x, pw, out, M=M))
    res = torch.randn(M, pw.N, device=DEV).half()
    us_r = per_node(lambda: ops.gemm(x, pw, out, M=M, residual=res))
    print(f"gemm lin M={M} N={N_} K={K}: {us:6.2f} us per node ({2.0 * M * N_ * K / us / 1e6:5.0f} TF/s); + residual {us_r:6.2f} us")
