#!/usr/bin/env python3
"""Micro-benchmark of moca_attention_f16 / moca_temporal_attention_f16 on the UNet's shapes (B=2, T=16), random q / k / v, HIP events.
Two figures per shape: "burst" = 10 back-to-back launches after 2 warm-ups from an idle chip, "sustained" = back-to-back launches for
>= 60 ms (what a launch inside the replayed 33 ms forward sees).  On the MFMA-heavy shapes the first launches from idle are the
SLOWER ones (the clock ramps up under load: 359 vs 314 us at 2560 tokens, profiles/r04_bench_attn.txt) -- quote the sustained figure."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from moca_video_amd import ops
ops.set_stream(None)
DEV = "cuda"
F = int(os.environ.get("BA_F", "32"))


def timed(fn, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


def run(name, fn, flops):
    fn(); fn()
    torch.cuda.synchronize()
    burst = timed(fn, 10)
    n = max(20, int(60e3 / burst))
    sus = timed(fn, n)
    print(f"{name:40s} burst {burst:8.1f} us {flops/burst/1e6:7.1f} TF/s   sustained ({n:4d} launches) {sus:8.1f} us {flops/sus/1e6:7.1f} TF/s", flush=True)
    torch.cuda.synchronize()
    import time; time.sleep(0.2)


for (heads, N) in ((5, 2560), (10, 640), (20, 160), (20, 40)):
    C = heads * 64
    qkv = torch.randn(F * N, 3 * C, device=DEV).half()
    out = torch.empty(F * N, C, device=DEV, dtype=torch.float16)
    run(f"spatial self  heads={heads} N={N}", lambda: ops.attention(qkv[:, :C], qkv[:, C:2*C], qkv[:, 2*C:], out, Bq=F, heads=heads, Nq=N, Nk=N, ldq=3*C, ldk=3*C, ldv=3*C, ldo=C, kv_div=1, scale=0.125), 4.0 * N * N * 64 * heads * F)
    q = torch.randn(F * N, C, device=DEV).half(); kv = torch.randn(2 * 77, 2 * C, device=DEV).half()
    run(f"spatial cross heads={heads} N={N} L=77", lambda: ops.attention(q, kv[:, :C], kv[:, C:], out, Bq=F, heads=heads, Nq=N, Nk=77, ldq=C, ldk=2*C, ldv=2*C, ldo=C, kv_div=16, scale=0.125), 4.0 * N * 77 * 64 * heads * F)
    run(f"temporal      heads={heads} HW={N}", lambda: ops.temporal_attention(qkv[:, :C], qkv[:, C:2*C], qkv[:, 2*C:], out, B=2, T=16, HW=N, heads=heads, ld_qkv=3*C, ldo=C, scale=0.125), 4.0 * 16 * 16 * 64 * heads * 2 * N)
