#!/usr/bin/env python3
"""Calibration only (never on the product path): what the vendor GEMM (torch.matmul -> hipBLASLt/rocBLAS) reaches on the
UNet's plain linear shapes on THIS device, random fp16 data -- a known-good reference for the hand-written kernels
(cdna_hip_programming.md rule 10: never infer a ceiling from your own attempts)."""
import time, torch
dev = "cuda"
def run(M, N, K, iters=20):
    a = torch.randn(M, K, device=dev, dtype=torch.float16); w = torch.randn(N, K, device=dev, dtype=torch.float16) * K ** -0.5
    for _ in range(3): (a @ w.t())
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters): (a @ w.t())
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / iters
    print(f"matmul M={M:6d} N={N:5d} K={K:5d}  {dt*1e6:8.1f} us  {2.0*M*N*K/dt/1e12:7.1f} TF/s", flush=True)
for M, N, K in [(81920, 320, 320), (81920, 960, 320), (81920, 2560, 320), (81920, 320, 1280), (20480, 640, 640), (20480, 1920, 640),
                (20480, 5120, 640), (20480, 640, 2560), (5120, 1280, 1280), (5120, 3840, 1280), (5120, 10240, 1280), (5120, 1280, 5120),
                (1280, 1280, 1280), (1280, 3840, 1280), (1280, 1280, 5120), (81920, 320, 2880), (20480, 640, 5760), (5120, 1280, 11520),
                (8192, 8192, 8192)]:
    run(M, N, K)
