#!/usr/bin/env python3
"""MOCA_EP_LNFOLD consumer epilogue against the plain linear on the same operands (same A, same-shaped W): what the fold costs
per kernel.  GPU only."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from moca_video_amd import ops

DEV = "cuda"


def run(name, fn, flops, iters=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    print(f"{name:60s} {dt*1e6:8.1f} us {flops/dt/1e12:7.1f} TF/s", flush=True)


def case(M, C, n, geglu, nparts):
    x = torch.randn(M, C, device=DEV).half()
    nn = 2 * n if geglu else n
    w, b = torch.randn(nn, C, device=DEV) * C ** -0.5, torch.randn(nn, device=DEV) * 0.1
    g, be = torch.rand(C, device=DEV) + 0.5, torch.randn(C, device=DEV) * 0.1
    pack = ops.pack_geglu if geglu else ops.pack_linear
    pw = pack(w, b)
    wf, bf = ops.fold_layernorm(w, b, g, be)
    pwf = ops.finish_lnfold(pack(wf, bf))
    xf = x.float()
    part = torch.zeros(nparts, M, 2, device=DEV)
    part[0, :, 0], part[0, :, 1] = xf.sum(1), (xf * xf).sum(1)
    out = torch.empty(M, n, device=DEV, dtype=torch.float16)
    tag = f"M={M} K={C} N={nn}{' geglu' if geglu else ''}"
    run(tag + " plain", lambda: ops.gemm(x, pw, out, M=M), 2.0 * M * nn * C, iters=60)
    run(tag + " plain", lambda: ops.gemm(x, pw, out, M=M), 2.0 * M * nn * C)
    run(tag + f" lnfold({nparts})", lambda: ops.gemm(x, pwf, out, M=M, lnfold=(part, nparts, 1e-5)), 2.0 * M * nn * C)
    run(tag + " plain, folded W", lambda: ops.gemm(x, pwf, out, M=M), 2.0 * M * nn * C)


if __name__ == "__main__":
    ops.set_stream(None)
    case(81920, 320, 1280, True, 1)
    case(81920, 320, 960, False, 1)
    case(20480, 640, 2560, True, 2)
    case(20480, 640, 1920, False, 2)
    case(5120, 1280, 5120, True, 10)
    case(5120, 1280, 3840, False, 10)
