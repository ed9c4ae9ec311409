#!/usr/bin/env python3
"""What a read-once / write-once pass reaches on this device: torch copy (vendor elementwise kernel) against the GroupNorm
apply / LayerNorm kernels on the UNet's activation sizes.  GPU only."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from moca_video_amd import ops
from moca_video_amd import lib as L

DEV = "cuda"


def run(name, fn, nbytes, iters=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    print(f"{name:56s} {dt*1e6:8.1f} us {nbytes/dt/1e12:6.2f} TB/s", flush=True)


ops.set_stream(None)
for F, HW, C, fps in ((32, 2560, 320, 16), (32, 2560, 320, 1), (32, 640, 640, 16), (32, 160, 1280, 16), (32, 40, 1280, 16)):
    x = torch.randn(F * HW, C, device=DEV).half()
    y = torch.empty_like(x)
    nb = 2 * x.numel() * 2
    run(f"torch copy            F={F} HW={HW} C={C}", lambda: y.copy_(x), nb)
    g, b = torch.ones(C, device=DEV), torch.zeros(C, device=DEV)
    ws = torch.empty(ops.groupnorm_ws_floats(F, HW, C), dtype=torch.float32, device=DEV)
    for slab in (0, 1):
        L.set_tuning(L.MOCA_TUNE_GN_SLAB, slab)
        run(f"groupnorm slab={slab}      F={F} HW={HW} C={C} fps={fps}", lambda: ops.groupnorm(x, y, g, b, F=F, HW=HW, Cn=C, frames_per_stat=fps, eps=1e-5, silu=True, ws=ws), nb)
    rows = 320 if C < 1280 else 256
    if (fps * HW) % rows == 0:
        cs = torch.randn(F * HW // rows, 2 * C, device=DEV).abs()
        run(f"groupnorm from colsum F={F} HW={HW} C={C} fps={fps}", lambda: ops.groupnorm_colsum(x, y, g, b, cs, tile_rows=rows, F=F, HW=HW, Cn=C, frames_per_stat=fps, eps=1e-5, silu=True, ws=ws), nb)
