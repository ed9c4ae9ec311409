#!/usr/bin/env python3
"""Time every recorded launch of one UNet plan (bench workload: B = 2n videos = n latents x (conditional, unconditional) context with
the shared prefix) in isolation with HIP events, on the operands the graph gives it, and print the total per (op, shape).

    python tools/plan_profile.py [B] > gpurun_out/plan_profile.txt           (on the GPU box, repo root)

PP_MODE
  "true" (default) -- TRUE OPERANDS.  The plan's buffers come from a pool with stream-ordered reuse, so after a forward a step's
      input buffers hold whatever later layers wrote there; replaying a step on them times it on stale (for attention: degenerate)
      data, which runs faster than real data (cdna_hip_programming.md rule 25: softmax work collapses, the clock rises; round 3's
      "214 us = 1.25 PFLOP/s" attention figure was such a replay).  Here one eager forward is run step by step; BEFORE each step
      every tensor argument it reads is cloned (a snapshot, <= PP_SNAP_GB at a time, in chunks of consecutive steps), and each
      timed replay group of the step starts from the restored snapshot: 1 warm-up + REP back-to-back launches between two events.
      Isolated = caches warm, chip otherwise idle; the in-graph time of the same launch is in the rocprof kernel trace
      (tools/ingraph_vs_hot.sh joins the two).
  "stale"  -- round-3 behaviour (no restore), kept only to show the difference.
  "cold"   -- true operands + a 1 GiB streaming write before every timed launch (operands come from HBM).

Columns: total / count / each [us], TF/s (executed FLOPs), GB/s (operand bytes touched once: A, W, output, residual), then the
roofline floor of the step = max(FLOP / 1.7 PFLOP/s, bytes / 5.5 TB/s) -- 1.7 PFLOP/s is what the matrix pipe sustains on random
fp16 data on this chip (tools/micro/mfma_peak.hip), 5.5 TB/s a streaming kernel's HBM rate -- and the slack each - floor.
A JSON copy (PP_JSON=path) carries the per-key numbers for tools/ingraph_vs_hot.py."""
import collections
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from moca_video_amd import ops

REP = int(os.environ.get("PP_REP", "5"))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
MODE = os.environ.get("PP_MODE", "true")
SNAP_BYTES = float(os.environ.get("PP_SNAP_GB", "96")) * 2 ** 30
MFMA_SUSTAINED, HBM_STREAM = 1.7e15, 5.5e12
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dm = bench.build_model(dev, seed=321)
unet = dm.model.diffusion_model
g = torch.Generator(device=dev).manual_seed(1)
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
plan = next(iter(unet._plans.values()))
st = plan.stream
ops.set_stream(st.cuda_stream)


def describe(s):
    """(key, executed FLOP, operand bytes touched once) of a recorded step"""
    fn, kw = s.func.__name__, s.keywords
    esz = lambda t: 0 if t is None else t.numel() * t.element_size()
    if fn == "gemm":
        pw = s.args[1]
        mode = {0: "lin", 1: "conv", 2: "tconv"}[kw.get("mode", 0)]
        key = f"gemm {mode:5s} M={kw['M']:6d} N={pw.N:5d} K={pw.K:5d}" + (" geglu" if pw.geglu else "") + \
              (" +res" if kw.get("residual") is not None else "") + (" +rowadd" if kw.get("rowadd") is not None else "") + \
              (f" splits={kw['splits']}" if kw.get("splits", 1) > 1 else "") + (" f32" if kw.get("out_f32") else "") + \
              (" +colsum" if kw.get("colsum") is not None else "") + (" +LN" if kw.get("ln") is not None else "") + \
              (" +rowsum" if kw.get("rowsum") is not None else "") + (" lnfold" if kw.get("lnfold") is not None else "") + \
              (" +gstat" if kw.get("gstat") is not None else "") + (" +tattn" if kw.get("tattn") is not None else "") + \
              (f" up{kw['up_phase']}" if kw.get("up_phase") else "") + (" cat2" if kw.get("a2") is not None else "") + \
              (" slabs" if kw.get("slabs") else "")
        M = kw["M"]
        flop = 2.0 * M * pw.N * pw.K
        n_out = (pw.n_out if pw.geglu else pw.N) if kw.get("tattn") is None else pw.N // 3
        cin = kw["conv"][0] if kw.get("conv") is not None else (kw["tconv"][0] if kw.get("tconv") is not None else pw.K)
        a_rows = M if kw.get("conv") is None else s.args[0].numel() // max(cin, 1)
        byt = 2.0 * a_rows * cin + 2.0 * pw.N * pw.K + 2.0 * M * n_out + (2.0 * M * n_out if kw.get("residual") is not None else 0.0)
        if kw.get("splits", 1) > 1:
            byt += 2 * 4.0 * kw["splits"] * M * pw.N              # fp32 slabs written and read back
        return key, flop, byt
    if fn in ("groupnorm_colsum", "groupnorm_gstat", "groupnorm"):
        tag = {"groupnorm_colsum": "groupnorm(colsum)", "groupnorm_gstat": "groupnorm(gstat)", "groupnorm": "groupnorm"}[fn]
        rows = kw["F"] * kw["HW"]
        return f"{tag} F={kw['F']} HW={kw['HW']} C={kw['Cn']} fps={kw['frames_per_stat']}", 0.0, \
            2.0 * rows * kw["Cn"] * (3 if fn == "groupnorm" else 2)
    if fn == "groupnorm_gstat_cat":
        rows = kw["F"] * kw["HW"]
        return f"groupnorm(gstat, virtual cat) F={kw['F']} HW={kw['HW']} C={kw['C1']}+{kw['C2']} fps={kw['frames_per_stat']}", 0.0, \
            4.0 * rows * (kw["C1"] + kw["C2"])
    if fn == "gemm_splitk_groupnorm":
        pw = s.args[1]
        return f"splitk reduce + groupnorm M={kw['M']} N={pw.N} splits={kw['splits']} fps={kw['frames_per_stat']}" + \
            (" +x" if kw.get("write_x") else ""), 0.0, 4.0 * kw["splits"] * kw["M"] * pw.N + 2.0 * kw["M"] * pw.N * (2 if kw.get("write_x") else 1)
    if fn == "gstat_accum":
        return f"gstat_accum F={kw['F']} HW={kw['HW']} C={kw['Cn']}", 0.0, 2.0 * kw["F"] * kw["HW"] * kw["Cn"]
    if fn == "layernorm":
        return f"layernorm M={kw['M']} C={kw['Cn']}", 0.0, 4.0 * kw["M"] * kw["Cn"]
    if fn == "attention":
        flop = 4.0 * kw["Bq"] * kw["heads"] * kw["Nq"] * kw["Nk"] * 64
        byt = 2.0 * kw["Bq"] * kw["heads"] * 64 * (2 * kw["Nq"] + 2 * kw["Nk"] / max(kw["kv_div"], 1))
        return f"attention Bq={kw['Bq']} h={kw['heads']} Nq={kw['Nq']} Nk={kw['Nk']}", flop, byt
    if fn == "temporal_attention":
        return f"temporal_attention B={kw['B']} T={kw['T']} HW={kw['HW']} h={kw['heads']}", 0.0, \
            2.0 * 4 * kw["B"] * kw["T"] * kw["HW"] * kw["heads"] * 64
    byt = sum(esz(t) for t in list(s.args) + list(kw.values()) if torch.is_tensor(t))
    return fn, 0.0, float(byt)


def tensors_of(s):
    out = []

    def walk(v):
        if torch.is_tensor(v):
            out.append(v)
        elif isinstance(v, (tuple, list)):
            for u in v:
                walk(u)
    for v in list(s.args) + list(s.keywords.values()):
        walk(v)
    return out


flush = torch.empty(1 << 28, dtype=torch.float32, device=dev) if MODE == "cold" else None
rows = collections.OrderedDict()
tot = 0.0


def time_step(s, snap):
    """1 warm-up + REP timed launches from the restored operands"""
    def restore():
        if snap is not None:
            for t, c in snap:
                t.copy_(c)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    restore()
    s()
    if MODE == "cold":
        us = 0.0
        for _ in range(3):
            restore()
            flush.fill_(1.0)
            e0.record(st); s(); e1.record(st); e1.synchronize()
            us += e0.elapsed_time(e1) * 1e3 / 3
        return us
    restore()
    e0.record(st)
    for _ in range(REP):
        s()
    e1.record(st)
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / REP


steps = list(plan.steps)
with torch.cuda.stream(st), torch.no_grad():
    i = 0
    while i < len(steps):
        # ---- one eager pass from the top: snapshot the operands of steps i .. j-1 as the forward reaches them
        snaps, used, j = {}, 0.0, i
        if MODE != "stale":
            for full in plan._gstat_full:
                ops.memset_zero(full)
            if plan._gstat_used:
                ops.memset_zero(plan._gstat_buf[:plan._gstat_used])
            for k, s in enumerate(steps):
                if k >= i:
                    ts_ = tensors_of(s)
                    need = sum(t.numel() * t.element_size() for t in ts_)
                    if k > i and used + need > SNAP_BYTES:
                        break
                    snaps[k] = [(t, t.clone()) for t in ts_]
                    used += need
                    j = k + 1
                s()
            torch.cuda.synchronize()
        else:
            j = len(steps)
        for k in range(i, j):
            s = steps[k]
            key, flop, byt = describe(s)
            us = time_step(s, snaps.get(k))
            r = rows.setdefault(key, [0, 0.0, flop, byt])
            r[0] += 1
            r[1] += us
            tot += us
        snaps.clear()
        torch.cuda.empty_cache()
        i = j
ops.set_stream(None)

floor_tot = 0.0
out_rows = []
for key, (cnt, us, flop, byt) in rows.items():
    floor = max(flop / MFMA_SUSTAINED, byt / HBM_STREAM) * 1e6
    floor_tot += floor * cnt
    out_rows.append((us, cnt, us / cnt, flop * cnt / us / 1e6 if flop else 0.0, byt * cnt / us / 1e3, floor, us / cnt - floor, key,
                     "mfma" if flop / MFMA_SUSTAINED >= byt / HBM_STREAM else "hbm"))
print(f"# B={B}{' (shared CFG prefix)' if SHARED else ''}, mode {MODE}: {len(steps)} steps, sum of isolated step times {tot / 1e3:.2f} ms; "
      f"sum of per-step roofline floors (max(FLOP / 1.7 PFLOP/s, operand bytes / 5.5 TB/s)) {floor_tot / 1e3:.2f} ms")
print(f"{'total_us':>9s} {'n':>3s} {'each_us':>8s} {'TF/s':>6s} {'GB/s':>6s} {'floor':>7s} {'slack_us':>8s} bound  step")
for us, cnt, each, tf, gbs, floor, slack, key, bound in sorted(out_rows, key=lambda r: -r[0]):
    print(f"{us:9.1f} {cnt:3d} {each:8.1f} {tf:6.0f} {gbs:6.0f} {floor:7.1f} {slack * cnt:8.1f} {bound:5s}  {key}")
if os.environ.get("PP_JSON"):
    json.dump({"B": B, "mode": MODE, "total_ms": tot / 1e3,
               "rows": [dict(key=r[7], n=r[1], each_us=r[2], tflops=r[3], gbs=r[4], floor_us=r[5], bound=r[8]) for r in out_rows]},
              open(os.environ["PP_JSON"], "w"), indent=1)
