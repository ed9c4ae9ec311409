#!/usr/bin/env python3
"""Single-launch GroupNorm of the small (1280-channel) tensors: slab kernel (S blocks per slab, each re-reducing it) against the
slab-in-registers kernel (MOCA_GN_SLAB_REG).  GPU only."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from moca_video_amd import ops
DEV = "cuda"


def run(name, fn, iters=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    print(f"{name:60s} {(time.perf_counter() - t0) / iters * 1e6:8.1f} us", flush=True)


ops.set_stream(None)
for F, HW, C, fps in ((32, 40, 1280, 16), (32, 40, 1280, 1), (32, 160, 1280, 1), (32, 160, 1280, 16), (32, 40, 2560, 1), (128, 40, 1280, 16), (128, 160, 1280, 1)):
    x = torch.randn(F * HW, C, device=DEV).half()
    y = torch.empty_like(x)
    g, b = torch.ones(C, device=DEV), torch.zeros(C, device=DEV)
    ws = torch.empty(ops.groupnorm_ws_floats(F, HW, C), dtype=torch.float32, device=DEV)
    for reg in ("0", "1"):
        os.environ["MOCA_GN_SLAB_REG"] = reg
        run(f"groupnorm F={F} HW={HW} C={C} fps={fps} slab_reg={reg}", lambda: ops.groupnorm(x, y, g, b, F=F, HW=HW, Cn=C, frames_per_stat=fps, eps=1e-5, silu=True, ws=ws))
