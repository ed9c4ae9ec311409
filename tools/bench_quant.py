#!/usr/bin/env python3
"""Does a 200-tile launch lose 22 % of the chip, or does the power budget of the 56 idle CUs speed the others up?  The 1280-channel
conv / temporal conv / linear shapes at M = 5120 (200 tiles of 256 x 128) against M = 6400 (250 tiles) and 6656 (260: two rounds)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from moca_video_amd import ops, lib as L
DEV = "cuda"


def run(name, fn, flops, iters=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    print(f"{name:48s} {dt*1e6:8.1f} us {flops/dt/1e12:7.1f} TF/s", flush=True)


ops.set_stream(None)
C = 1280
for frames in (32, 40, 42, 48, 51, 52):
    H, W = 10, 16
    M = frames * H * W
    x = torch.randn(M, C, device=DEV).half()
    pw = ops.pack_conv3x3(torch.randn(C, C, 3, 3, device=DEV) * (9 * C) ** -0.5, torch.zeros(C, device=DEV))
    out = torch.empty(M, C, device=DEV, dtype=torch.float16)
    run(f"conv 1280->1280 M={M} tiles={-(-M//256)*10}", lambda: ops.gemm(x, pw, out, M=M, mode=L.MOCA_A_CONV3X3, conv=(C, H, W, H, W, 1, 0)), 2.0 * M * C * 9 * C)
    pl = ops.pack_linear(torch.randn(C, C, device=DEV) * C ** -0.5, torch.zeros(C, device=DEV))
    run(f"lin  1280->1280 M={M} tiles={-(-M//256)*10}", lambda: ops.gemm(x, pl, out, M=M), 2.0 * M * C * C)
