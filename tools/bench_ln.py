#!/usr/bin/env python3
"""diagnostic: linear(+res) N=320 at L0 -- tall tiling, wide tiling, wide + LayerNorm epilogue, and the separate LayerNorm kernel"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from moca_video_amd import ops
ops.set_stream(None)
DEV = "cuda"
def run(name, fn, iters=30):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters): fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / iters
    print(f"{name:50s} {dt*1e6:9.1f} us", flush=True)
N = 320
for M, K in ((81920, 320), (81920, 1280), (327680, 320), (327680, 1280)):
    a = torch.randn(M, K, device=DEV).half(); w = torch.randn(N, K, device=DEV) * K ** -0.5
    pw = ops.pack_linear(w, torch.zeros(N, device=DEV))
    res = torch.randn(M, N, device=DEV).half()
    out = torch.empty(M, N, device=DEV, dtype=torch.float16); ln = torch.empty_like(out)
    g = torch.ones(N, device=DEV); b = torch.zeros(N, device=DEV)
    os.environ["MOCA_GEMM_WIDE"] = "0"
    run(f"M={M} K={K} tall 320x160 linear+res", lambda: ops.gemm(a, pw, out, M=M, residual=res))
    os.environ["MOCA_GEMM_WIDE"] = "1"
    run(f"M={M} K={K} wide 160x320 linear+res", lambda: ops.gemm(a, pw, out, M=M, residual=res))
    run(f"M={M} K={K} wide + LN epilogue", lambda: ops.gemm(a, pw, out, M=M, residual=res, ln=(g, b, ln, 1e-5)))
    run(f"layernorm kernel M={M} C=320", lambda: ops.layernorm(out, ln, g, b, M=M, Cn=N))
