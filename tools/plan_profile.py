#!/usr/bin/env python3
"""Time every recorded launch of one UNet plan (bench workload, B=2 batched CFG) in isolation with HIP events and
print the total per (op, shape).  Each step is replayed `REP` times back to back between two events, so the figure
is the steady-state launch-to-launch time of that step alone (no inter-kernel overlap, warm caches).

    python tools/plan_profile.py [B] > gpurun_out/plan_profile.txt
"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from moca_video_amd import ops, lib as L

REP = 5
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
# PP_MODE: "hot" (default) = REP back-to-back replays after one warm-up (operands L2 / Infinity-Cache resident, short bursts);
#          "sustained" = every step replayed for >= 25 ms without a pause (hot operands, but the chip sits at its sustained
#                        power state like inside the 35 ms graph);
#          "cold" = a 1 GiB streaming write between replays, each replay timed on its own (operands come from HBM like the
#                   weights of the next layer do inside the graph)
MODE = os.environ.get("PP_MODE", "hot")
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dm = bench.build_model(dev, seed=321)
unet = dm.model.diffusion_model
g = torch.Generator(device=dev).manual_seed(1)
# the plan the samplers run: B = 2n videos = n latents x (conditional, unconditional) context with the shared prefix
# (PP_SHARED=0: a plain batch of B independent videos, the round-2 structure)
SHARED = os.environ.get("PP_SHARED", "1") != "0" and B % 2 == 0
n = B // 2 if SHARED else B
x = torch.randn(n, 4, 16, 40, 64, device=dev, generator=g)
ctx = torch.randn(B, 77, 1024, device=dev, generator=g)
ts = torch.full((n,), 500, device=dev, dtype=torch.long)
with torch.no_grad():
    for _ in range(2):
        if SHARED:
            unet.forward_segments(x, ts, [ctx[:n], ctx[n:]], fps=torch.tensor([10] * n, device=dev), shared_x=True)
        else:
            unet(x, ts, ctx, fps=torch.tensor([10] * B, device=dev))
torch.cuda.synchronize()
plan = next(iter(unet._plans.values())) if hasattr(unet, "_plans") else None
assert plan is not None
st = plan.stream
ops.set_stream(st.cuda_stream)
rows = collections.OrderedDict()
tot = 0.0
flush = torch.empty(1 << 28, dtype=torch.float32, device=dev) if MODE == "cold" else None
pending = []
seq_steps = []
with torch.cuda.stream(st):
    for s in plan.steps:
        fn, kw = s.func.__name__, s.keywords
        if fn == "gemm":
            pw = s.args[1]
            mode = {0: "lin", 1: "conv", 2: "tconv"}[kw.get("mode", 0)]
            key = f"gemm {mode:5s} M={kw['M']:6d} N={pw.N:5d} K={pw.K:5d}" + (" geglu" if pw.geglu else "") + \
                  (" +res" if kw.get("residual") is not None else "") + (" +rowadd" if kw.get("rowadd") is not None else "") + \
                  (f" splits={kw['splits']}" if kw.get("splits", 1) > 1 else "") + (" f32" if kw.get("out_f32") else "") + \
                  (" +colsum" if kw.get("colsum") is not None else "") + (" +LN" if kw.get("ln") is not None else "") + \
                  (" +rowsum" if kw.get("rowsum") is not None else "") + (" lnfold" if kw.get("lnfold") is not None else "") + \
                  (" +gstat" if kw.get("gstat") is not None else "") + (" +tattn" if kw.get("tattn") is not None else "")
            flop = 2.0 * kw["M"] * pw.N * pw.w.shape[1]
            if kw.get("conv") is not None and kw["conv"][6]:
                pass
        elif fn == "groupnorm_colsum":
            key = f"groupnorm(colsum) F={kw['F']} HW={kw['HW']} C={kw['Cn']} fps={kw['frames_per_stat']}"
            flop = 0
        elif fn == "groupnorm_gstat":
            key = f"groupnorm(gstat) F={kw['F']} HW={kw['HW']} C={kw['Cn']} fps={kw['frames_per_stat']}"
            flop = 0
        elif fn == "groupnorm":
            key = f"groupnorm F={kw['F']} HW={kw['HW']} C={kw['Cn']} fps={kw['frames_per_stat']}"
            flop = 0
        elif fn == "layernorm":
            key = f"layernorm M={kw['M']} C={kw['Cn']}"
            flop = 0
        elif fn == "attention":
            key = f"attention Bq={kw['Bq']} h={kw['heads']} Nq={kw['Nq']} Nk={kw['Nk']}"
            flop = 4.0 * kw["Bq"] * kw["heads"] * kw["Nq"] * kw["Nk"] * 64
        elif fn == "temporal_attention":
            key = f"temporal_attention B={kw['B']} T={kw['T']} HW={kw['HW']} h={kw['heads']}"
            flop = 0
        else:
            key = fn
            flop = 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        if MODE in ("seq", "seqpf"):      # the forward in order, every launch once per pass (operands as cold / warm as inside the graph),
            # 3 timed passes; "seqpf": the step's weights are read once (torch sum) right before it -> W in L2 / Infinity Cache
            seq_steps.append((key, flop, s, s.args[1].w if fn == "gemm" else None))
            continue
        if MODE == "nosync":              # warm-up + 2 timed replays per step, nothing synchronised until the very end: the chip
            s()                           # never idles (like inside the graph) but every timed launch finds its operands hot
            e0.record(st); s(); s(); e1.record(st)
            pending.append((key, flop, e0, e1))
            continue
        s()
        if MODE == "cold":
            us = 0.0
            for _ in range(3):
                flush.fill_(1.0)
                e0.record(st)
                s()
                e1.record(st)
                e1.synchronize()
                us += e0.elapsed_time(e1) * 1e3 / 3
        else:
            rep = REP
            if MODE == "sustained":
                e0.record(st); s(); e1.record(st); e1.synchronize()
                rep = max(REP, int(25e3 / max(e0.elapsed_time(e1) * 1e3, 1.0)))
            e0.record(st)
            for _ in range(rep):
                s()
            e1.record(st)
            e1.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / rep
        r = rows.setdefault(key, [0, 0.0, flop])
        r[0] += 1
        r[1] += us
        tot += us
if seq_steps:
    with torch.cuda.stream(st):
        for it in range(4):
            for key, flop, s, w in seq_steps:
                if MODE == "seqpf" and w is not None:
                    w.sum()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(st); s(); e1.record(st)
                if it > 0:
                    pending.append((key, flop / 1.5, e0, e1))     # (3 timed passes; the summary below divides by 2)
    torch.cuda.synchronize()
    seq_n = 3
ops.set_stream(None)
if pending:
    torch.cuda.synchronize()
    for key, flop, e0, e1 in pending:
        us = e0.elapsed_time(e1) * 1e3 / (3 if seq_steps else 2)
        r = rows.setdefault(key, [0, 0.0, flop * (1.5 if seq_steps else 1.0)])
        r[0] += (1.0 / 3 if seq_steps else 1)
        r[1] += us
        tot += us
print(f"# B={B}{' (shared CFG prefix)' if SHARED else ''}, mode {MODE}: {len(plan.steps)} steps, sum of isolated step times {tot / 1e3:.2f} ms")
print(f"{'total_us':>9s} {'n':>3s} {'each_us':>8s} {'TF/s':>6s}  step")
for key, (n, us, flop) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
    tf = flop * n / us / 1e6 if flop else 0.0
    n = int(round(n))
    print(f"{us:9.1f} {n:3d} {us / n:8.1f} {tf:6.0f}  {key}")
