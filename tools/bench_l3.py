#!/usr/bin/env python3
"""Split-k / tile-size landscape of the M = 1280 (5 x 8 latent) level: every GEMM shape of that level with splits 1..8 on the
256-row kernel and on the 128-row kernel (MOCA_FORCE_SMALL_TILE).  GPU only."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from moca_video_amd import ops, lib as L

DEV = "cuda"
M, T, H, W = 1280, 16, 5, 8
MS = [int(a) for a in sys.argv[1:]] or [1280]


def run(name, fn, flops, iters=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    print(f"{name:56s} {dt*1e6:8.1f} us {flops/dt/1e12:7.1f} TF/s", flush=True)


def case(kind, c_in, n, M):
    F = M // (H * W)
    x = torch.randn(M, c_in, device=DEV).half()
    if kind == "lin":
        pw = ops.pack_linear(torch.randn(n, c_in, device=DEV) * c_in ** -0.5, torch.zeros(n, device=DEV)); kw = {}; K = c_in
    elif kind == "tconv":
        pw = ops.pack_tconv3(torch.randn(n, c_in, 3, 1, 1, device=DEV) * (3 * c_in) ** -0.5, torch.zeros(n, device=DEV))
        kw = dict(mode=L.MOCA_A_TCONV3, tconv=(c_in, T, H * W)); K = 3 * c_in
    else:
        pw = ops.pack_conv3x3(torch.randn(n, c_in, 3, 3, device=DEV) * (9 * c_in) ** -0.5, torch.zeros(n, device=DEV))
        kw = dict(mode=L.MOCA_A_CONV3X3, conv=(c_in, H, W, H, W, 1, 0)); K = 9 * c_in
    out = torch.empty(M, n, device=DEV, dtype=torch.float16)
    for small in ((False, True) if os.environ.get("L3_SMALL") else (False,)):
        for s in (1, 2, 3, 4, 5, 6, 8, 10):
            if s > K // 64:
                continue
            ws = torch.empty(s * M * pw.N, device=DEV, dtype=torch.float32) if s > 1 else None
            fn = lambda: ops.gemm(x, pw, out, M=M, splits=s, splitk_ws=ws, force_small=small, **kw)
            run(f"{kind} M={M} N={n} K={K} {'small' if small else 'big  '} s={s}", fn, 2.0 * M * n * K)


if __name__ == "__main__":
    ops.set_stream(None)
    for M in MS:
        if os.environ.get("L3_SET", "a") == "a":
            case("lin", 1280, 1280, M)
            case("tconv", 1280, 1280, M)
            case("conv", 1280, 1280, M)
            case("lin", 5120, 1280, M)
        else:
            case("lin", 2560, 1280, M)
            case("conv", 2560, 1280, M)
            case("lin", 1280, 3840, M)
