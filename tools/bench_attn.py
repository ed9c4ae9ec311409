#!/usr/bin/env python3
"""Micro-benchmark of moca_attention_f16 / moca_temporal_attention_f16 on the UNet's shapes (B=2, T=16)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from moca_video_amd import ops
ops.set_stream(None)
DEV = "cuda"
F = 32

def run(name, fn, flops, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters): fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / iters
    print(f"{name:44s} {dt*1e6:9.1f} us  {flops/dt/1e12:8.1f} TF/s", flush=True)

for (heads, N) in ((5, 2560), (10, 640), (20, 160), (20, 40)):
    C = heads * 64
    qkv = torch.randn(F * N, 3 * C, device=DEV).half()
    out = torch.empty(F * N, C, device=DEV, dtype=torch.float16)
    run(f"spatial self  heads={heads} N={N}", lambda: ops.attention(qkv[:, :C], qkv[:, C:2*C], qkv[:, 2*C:], out, Bq=F, heads=heads, Nq=N, Nk=N, ldq=3*C, ldk=3*C, ldv=3*C, ldo=C, kv_div=1, scale=0.125), 4.0 * N * N * 64 * heads * F)
    q = torch.randn(F * N, C, device=DEV).half(); kv = torch.randn(2 * 77, 2 * C, device=DEV).half()
    run(f"spatial cross heads={heads} N={N} L=77", lambda: ops.attention(q, kv[:, :C], kv[:, C:], out, Bq=F, heads=heads, Nq=N, Nk=77, ldq=C, ldk=2*C, ldv=2*C, ldo=C, kv_div=16, scale=0.125), 4.0 * N * 77 * 64 * heads * F)
    run(f"temporal      heads={heads} HW={N}", lambda: ops.temporal_attention(qkv[:, :C], qkv[:, C:2*C], qkv[:, 2*C:], out, B=2, T=16, HW=N, heads=heads, ld_qkv=3*C, ldo=C, scale=0.125), 4.0 * 16 * 16 * 64 * heads * 2 * N)
