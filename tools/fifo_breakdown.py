#!/usr/bin/env python3
"""Where one outer MoCA-FIFO iteration (full size: 8 windows of [1,4,16,40,64], cond = 2 prompts / 154 tokens, uncond 77,
CFG 12, mask injection, FreeInit shift) spends its time -- BEFORE (host-driven loop: two batched UNet launches + 8 ddim_step
calls + clone-based shift, `use_graph=False`) and AFTER (fifo_graph.FifoEngine: the whole iteration as one hipGraph).

    python tools/fifo_breakdown.py > gpurun_out/fifo_breakdown.txt
"""
import ctypes as C
import os
import sys
import time
import types

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from moca_video_amd import fifo as F
from moca_video_amd import lib as L
from moca_video_amd import ops
from moca_video_amd.fifo_graph import FifoEngine
from moca_video_amd.plan import _Plan
from moca_video_amd.sampler import DDIMSampler

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
lib = L.load()
dm = bench.build_model(dev)
unet = dm.model.diffusion_model
T, H, W, Q = 16, 40, 64, 72
args = types.SimpleNamespace(num_inference_steps=64, video_length=T, lookahead_denoising=True, num_partitions=4, new_video_length=100)
s = DDIMSampler(dm)
s.make_schedule(64, ddim_eta=1.0, verbose=False)
g = torch.Generator(device=dev).manual_seed(7)
lat0 = torch.randn(1, 4, Q, H, W, device=dev, generator=g)
fps = torch.tensor([10], device=dev)
cond = {"c_crossattn": [torch.randn(1, 77, 1024, device=dev, generator=g), torch.randn(1, 77, 1024, device=dev, generator=g)], "fps": fps}
uc_emb = torch.randn(1, 77, 1024, device=dev, generator=g)
uc = {"c_crossattn": [uc_emb], "fps": fps}
mask = torch.zeros(1, 1, Q, H, W, device=dev)
mask[..., 10:30, 16:48] = 1.0
cimg = torch.rand(1, 4, 1, H, W, device=dev, generator=g)


def events():
    a, b = C.c_void_p(), C.c_void_p()
    lib.moca_event_create(C.byref(a)); lib.moca_event_create(C.byref(b))
    return a, b


def elapsed(a, b):
    ms = C.c_float()
    lib.moca_event_elapsed_ms(a, b, C.byref(ms))
    return ms.value


# ---------------------------------------------------------------------------------------------------------------- BEFORE
lat = lat0.clone()
run = lambda n: F.fifo_ddim_sampling(args, dm, cond, (1, 4, T, H, W), s, cfg_scale=12.0, uc_emb=uc_emb, latents=lat,
                                     conditioned_image=cimg, masks=mask.clone(), n_iterations=n, use_graph=False)
run(2)
torch.cuda.synchronize()
t0 = time.perf_counter(); run(4); torch.cuda.synchronize(); tot_free = (time.perf_counter() - t0) / 4
tm = {"unet (2 launches: B=8 L=154, B=8 L=77)": 0.0, "ddim_step x 8": 0.0, "shift_latents": 0.0}


def wrap(obj, name, key):
    orig = getattr(obj, name)

    def f(*a, **k):
        torch.cuda.synchronize(); t0 = time.perf_counter(); r = orig(*a, **k); torch.cuda.synchronize(); tm[key] += time.perf_counter() - t0
        return r
    setattr(obj, name, f)
    return orig


o1 = wrap(s, "unet_windows", "unet (2 launches: B=8 L=154, B=8 L=77)")
o2 = wrap(s, "ddim_step", "ddim_step x 8")
o3 = wrap(F, "shift_latents", "shift_latents")
torch.cuda.synchronize(); t0 = time.perf_counter(); run(4); torch.cuda.synchronize(); tot = (time.perf_counter() - t0) / 4
s.unet_windows, s.ddim_step, F.shift_latents = o1, o2, o3
print("BEFORE  host-driven loop (round 2 structure; fifo_ddim_sampling(use_graph=False)), per outer iteration:")
print(f"  wall clock, free running                 {tot_free * 1e3:8.2f} ms")
print(f"  wall clock with a sync around every part {tot * 1e3:8.2f} ms")
for k, v in tm.items():
    print(f"    {k:40s} {v / 4 * 1e3:8.2f} ms")
print(f"    {'rest (window clones, write-back, python)':40s} {(tot - sum(tm.values()) / 4) * 1e3:8.2f} ms")
for pl in list(unet._plans.values()):
    pl.close()
unet._plans.clear()

# ---------------------------------------------------------------------------------------------------------------- AFTER
eng = FifoEngine(args, dm, s, cond, uc, 12.0, lat0.clone(), conditioned_image=cimg, masks=mask.clone(), n_slots=8, seed=1)
for _ in range(3):
    eng.step()
torch.cuda.synchronize()
N = 12
a, b = events()
h = C.c_void_p(eng.plan.stream.cuda_stream)
t0 = time.perf_counter()
lib.moca_event_record(a, h)
for _ in range(N):
    eng.step()
lib.moca_event_record(b, h)
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
t_wall = time.perf_counter() - t0
it_ms = elapsed(a, b) / N
finite = bool(torch.isfinite(eng.latents()).all())
print(f"\nAFTER   one hipGraph per iteration (fifo_graph.FifoEngine: {eng.n_unet_launches} UNet launches, B = 16 = 8 windows x 2 contexts with a shared prefix, + 12 more):")
print(f"  per iteration, HIP events around {N} replays    {it_ms:8.2f} ms   (wall {t_wall / N * 1e3:.2f} ms; host enqueue {t_host / N * 1e3:.3f} ms per iteration; queue finite: {finite})")

# the UNet launches alone: a second plan of the same signature, replayed as its own graph
plan = _Plan(unet, eng.plan.B, T, H, W, tuple(eng.plan.segs), torch.float32, dev, shared_x=eng.plan.reps > 1)
x = torch.randn(eng.plan.Bx, 4, T, H, W, device=dev, generator=g)
for _ in range(3):
    plan.run(x, eng.plan.t_rows, eng.plan.fps_rows, eng.plan.ctx)
torch.cuda.synchronize()
hp = C.c_void_p(plan.stream.cuda_stream)
lib.moca_event_record(a, hp)
for _ in range(N):
    L.check(lib.moca_graph_launch(plan.graph, hp))
lib.moca_event_record(b, hp)
torch.cuda.synchronize()
un_ms = elapsed(a, b) / N
print(f"  the UNet launches alone (same plan signature, own graph) {un_ms:8.2f} ms   -> everything else in the iteration: "
      f"{it_ms - un_ms:.2f} ms = {(it_ms - un_ms) / it_ms * 100:.1f} %")
plan.close()

# the non-UNet launches in isolation (eager, back to back)
pre, post = eng.plan.steps[0], eng.plan.steps[-1]
ops.set_stream(eng.plan.stream.cuda_stream)
REP = 20
for name, fn in (("noise (Philox) + window gather", pre), ("guidance + ddim_step x 8 + write-back, FreeInit mix (7 launches), advance", post)):
    with torch.cuda.stream(eng.plan.stream):
        fn()
        lib.moca_event_record(a, h)
        for _ in range(REP):
            fn()
        lib.moca_event_record(b, h)
    torch.cuda.synchronize()
    print(f"    {name:80s} {elapsed(a, b) / REP * 1e3:8.1f} us")
ops.set_stream(None)
print(f"\n  UNet-steps per iteration 16 -> {16 / (it_ms * 1e-3):.1f} UNet-steps/s; 148 iterations = {148 * it_ms * 1e-3:.2f} s per video before decode")
eng.close()
