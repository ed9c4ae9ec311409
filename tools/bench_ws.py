#!/usr/bin/env python3
"""Same-box alternating A/B of the weight-stationary 320 -> 320 kernel (gemm_ws.hip, MOCA_TUNE_GEMM_WS = 2: wherever it applies; env MOCA_WS_4WAVE=1: its first, 4-wave form) against the staggered 160 x 320
tiling it replaces, on the UNet's launches of that shape: plain / +rowsum / +res / +res +rowsum at B = 2 (M = 81920) and B = 16
(M = 655360).  HOT = the same operands every launch (what an isolated replay sees: at B = 2 they sit in the 256 MB Infinity Cache);
COLD = the launches rotate through enough operand sets to exceed it (what a launch inside the graph sees: A and the residual were
written > 256 MB of traffic ago).  HIP events around `iters` launches; GB/s = algorithmic bytes (A + residual + out) / time."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from moca_video_amd import lib as L
from moca_video_amd import ops

DEV = "cuda"


def bench(M, res, rows, cold, knob, iters=30, gstat=False):
    old = L.set_tuning(L.MOCA_TUNE_GEMM_WS, knob)
    try:
        nset = max(1, int(1.5 * (256 << 20) / (M * 640 * (3 if res else 2)))) + 1 if cold else 1
        nset = min(nset, 12)
        g = torch.Generator(device=DEV).manual_seed(1)
        pw = ops.pack_linear(torch.randn(320, 320, device=DEV, generator=g) * 320 ** -0.5, torch.randn(320, device=DEV, generator=g))
        xs = [torch.randn(M, 320, device=DEV, generator=g).half() for _ in range(nset)]
        rs = [torch.randn(M, 320, device=DEV, generator=g).half() for _ in range(nset)] if res else [None] * nset
        outs = [torch.empty(M, 320, device=DEV, dtype=torch.float16) for _ in range(nset)]
        cols = ops.gemm_rowsum_cols(xs[0], pw, M=M, residual=rs[0], rowsum=True) if rows else 0
        part = torch.empty(320 // cols * M, 2, device=DEV, dtype=torch.float32) if rows else None
        gst = torch.zeros(M // 2560 * 64, dtype=torch.int64, device=DEV) if gstat else None          # per-frame statistics, 2560 pixels per frame
        fn = lambda i: ops.gemm(xs[i % nset], pw, outs[i % nset], M=M, residual=rs[i % nset], rowsum=part, gstat=(gst, 2560) if gstat else None)
        for i in range(2 * nset + 2):
            fn(i)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(iters):
            fn(i)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / iters
        return us, M * 640 * (3 if res else 2) / us * 1e-3
    finally:
        L.set_tuning(L.MOCA_TUNE_GEMM_WS, old)


if __name__ == "__main__":
    ops.set_stream(None)
    rounds = int(os.environ.get("ROUNDS", "2"))
    print(f"{'launch':44s} {'tiled us':>9s} {'GB/s':>6s} {'ws us':>9s} {'GB/s':>6s}  ws/tiled")
    for M in (81920, 655360):
        for res, rows in ((False, False), (False, True), (True, False), (True, True), (True, "gstat")):
            for cold in (False, True):
                t = {0: [], 2: []}
                for _ in range(rounds):
                    for knob in (0, 2):
                        t[knob].append(bench(M, res, rows is True, cold, knob, gstat=rows == "gstat"))
                a, b = ((min(x[0] for x in tt), max(x[1] for x in tt)) for tt in (t[0], t[2]))
                name = f"lin M={M} 320x320{' +res' if res else ''}{' +gstat' if rows == 'gstat' else (' +rowsum' if rows else '')} {'COLD' if cold else 'hot'}"
                print(f"{name:44s} {a[0]:9.1f} {a[1]:6.0f} {b[0]:9.1f} {b[1]:6.0f}  {b[0] / a[0]:.3f}", flush=True)
