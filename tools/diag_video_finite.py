#!/usr/bin/env python3
"""diagnostic: magnitude of the emitted latents / the queue over the 148 FIFO iterations of the bench's video leg"""
import os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from moca_video_amd.fifo import base_ddim_sampling, prepare_latents
from moca_video_amd.fifo_graph import FifoEngine
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
dm = bench.build_model(dev)
zdd = bench.ZeroDataDenoiser(dm) if (len(sys.argv) < 2 or sys.argv[1] == 'zdd') else None
T, H, W, Q = 16, 40, 64, 72
args = types.SimpleNamespace(num_inference_steps=64, video_length=T, lookahead_denoising=True, num_partitions=4, new_video_length=100)
g = torch.Generator(device=dev).manual_seed(9)
c1, c2, uc_emb = (torch.randn(1, 77, 1024, device=dev, generator=g) for _ in range(3))
fps = torch.tensor([10], device=dev)
base, sampler, samples = base_ddim_sampling(dm, {"c_crossattn": [c1], "fps": fps}, [1, 4, T, H, W], 64, 1.0, 12.0, uc_emb=uc_emb)
print("base samples absmax", float(samples.abs().max()), "std", float(samples.std()))
lat = prepare_latents(args, None, sampler, initial_latents=samples)
print("queue absmax per 8 frames", [round(float(lat[:, :, i:i + 8].abs().max()), 1) for i in range(0, Q, 8)])
eng = FifoEngine(args, dm, sampler, {"c_crossattn": [c1, c2], "fps": fps}, {"c_crossattn": [uc_emb], "fps": fps}, 12.0, lat, n_slots=148, seed=9)
for i in range(148):
    eng.step()
    if i < 6 or i % 8 == 0:
        q = eng.latents()
        xp, p0 = eng.window_outputs()
        print(i, "queue absmax per 8 frames", [round(float(q[:, :, k:k + 8].abs().max()), 1) for k in range(0, Q, 8)],
              "eps absmax", round(float(eng.plan.out.abs().max()), 2), "finite", bool(torch.isfinite(q).all()), flush=True)
em = eng.emitted_frames(0, 148)
print("emitted absmax", [round(float(em[:, :, k].abs().max()), 1) for k in range(0, 148, 6)])
