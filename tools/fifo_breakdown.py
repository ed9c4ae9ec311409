#!/usr/bin/env python3
"""diagnostic: where one outer MoCA-FIFO iteration (full size, batched windows) spends its time"""
import os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import bench
from moca_video_amd.sampler import DDIMSampler
from moca_video_amd import fifo as F
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
dm = bench.build_model(dev)
T, H, W = 16, 40, 64
args = types.SimpleNamespace(num_inference_steps=64, video_length=T, lookahead_denoising=True, num_partitions=4, new_video_length=100)
s = DDIMSampler(dm); s.make_schedule(64, ddim_eta=1.0, verbose=False)
g = torch.Generator(device=dev).manual_seed(7)
Q = 72
lat = torch.randn(1, 4, Q, H, W, device=dev, generator=g)
cond = {"c_crossattn": [torch.randn(1, 77, 1024, device=dev, generator=g), torch.randn(1, 77, 1024, device=dev, generator=g)], "fps": torch.tensor([10], device=dev)}
uc = torch.randn(1, 77, 1024, device=dev, generator=g)
mask = torch.zeros(1, 1, Q, H, W, device=dev); mask[..., 10:30, 16:48] = 1.0
cimg = torch.rand(1, 4, 1, H, W, device=dev, generator=g)
run = lambda n: F.fifo_ddim_sampling(args, dm, cond, (1, 4, T, H, W), s, cfg_scale=12.0, uc_emb=uc, latents=lat, conditioned_image=cimg, masks=mask, n_iterations=n)
run(2); torch.cuda.synchronize()
# instrument
tm = {"unet": 0.0, "ddim_step": 0.0, "shift": 0.0}
def wrap(obj, name, key):
    orig = getattr(obj, name)
    def f(*a, **k):
        torch.cuda.synchronize(); t0 = time.perf_counter(); r = orig(*a, **k); torch.cuda.synchronize(); tm[key] += time.perf_counter() - t0; return r
    setattr(obj, name, f)
wrap(s, "unet_windows", "unet"); wrap(s, "ddim_step", "ddim_step"); wrap(F, "shift_latents", "shift")
torch.cuda.synchronize(); t0 = time.perf_counter(); run(4); torch.cuda.synchronize(); tot = time.perf_counter() - t0
print(f"per iteration: total {tot/4*1e3:.1f} ms; " + ", ".join(f"{k} {v/4*1e3:.1f} ms" for k, v in tm.items()) + f"; rest {(tot-sum(tm.values()))/4*1e3:.1f} ms")
