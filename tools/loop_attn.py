#!/usr/bin/env python3
"""the 2560-token spatial self-attention launch (attention_v4, F = 32, 5 heads) looping for LOOP_S seconds; prints the mean launch time"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from moca_video_amd import ops
ops.set_stream(None)
F, heads, N = 32, 5, 2560
C = heads * 64
qkv = torch.randn(F * N, 3 * C, device="cuda").half()
out = torch.empty(F * N, C, device="cuda", dtype=torch.float16)
t0, n = time.time(), 0
while time.time() - t0 < float(os.environ.get("LOOP_S", "12")):
    for _ in range(200):
        ops.attention(qkv[:, :C], qkv[:, C:2*C], qkv[:, 2*C:], out, Bq=F, heads=heads, Nq=N, Nk=N, ldq=3*C, ldk=3*C, ldv=3*C, ldo=C, kv_div=1, scale=0.125)
    torch.cuda.synchronize(); n += 200
print(f"attention_v4 2560 tokens: {n} launches, {(time.time() - t0) / n * 1e6:.1f} us each")
