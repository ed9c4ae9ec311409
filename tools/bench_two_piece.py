#!/usr/bin/env python3
"""Two blocks per output tile (sk_big) against the plain launch on the 200-tile shapes of the 1280-channel level.  GPU only."""
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
    print(f"{name:56s} {dt*1e6:8.1f} us {flops/dt/1e12:7.1f} TF/s", flush=True)


ops.set_stream(None)
M, C, H, W, T = 5120, 1280, 10, 16, 16
x = torch.randn(M, C, device=DEV).half()
out = torch.empty(M, C, device=DEV, dtype=torch.float16)
os.environ["MOCA_GEMM_TWO_PIECE"] = "1"
cases = [("conv 1280", ops.pack_conv3x3(torch.randn(C, C, 3, 3, device=DEV) * (9 * C) ** -0.5, torch.zeros(C, device=DEV)), dict(mode=L.MOCA_A_CONV3X3, conv=(C, H, W, H, W, 1, 0)), x),
         ("tconv 1280", ops.pack_tconv3(torch.randn(C, C, 3, 1, 1, device=DEV) * (3 * C) ** -0.5, torch.zeros(C, device=DEV)), dict(mode=L.MOCA_A_TCONV3, tconv=(C, T, H * W)), x),
         ("lin 1280", ops.pack_linear(torch.randn(C, C, device=DEV) * C ** -0.5, torch.zeros(C, device=DEV)), {}, x),
         ("lin 2560", ops.pack_linear(torch.randn(C, 2 * C, device=DEV) * C ** -0.5, torch.zeros(C, device=DEV)), {}, torch.randn(M, 2 * C, device=DEV).half()),
         ("lin 5120", ops.pack_linear(torch.randn(C, 4 * C, device=DEV) * C ** -0.5, torch.zeros(C, device=DEV)), {}, torch.randn(M, 4 * C, device=DEV).half())]
for name, pw, kw, a in cases:
    fl = 2.0 * M * pw.N * pw.w.shape[1]
    run(name + " plain", lambda: ops.gemm(a, pw, out, M=M, **kw), fl)
    big, wsb, sw = ops.gemm_two_piece(a, pw, M=M, **kw)
    if big == 0:
        print(name, "does not qualify"); continue
    ws = torch.empty(wsb // 4, dtype=torch.float32, device=DEV)
    sync = torch.zeros(sw, dtype=torch.int32, device=DEV)
    nk = pw.w.shape[1] // 64
    for b in sorted({big, max(1, big - nk // 16), min(nk - 1, big + nk // 16)}):
        run(name + f" two-piece big={b}/{nk}", lambda: ops.gemm(a, pw, out, M=M, two_piece=(b, ws, sync), **kw), fl)
