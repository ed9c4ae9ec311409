#!/usr/bin/env python3
"""Per-step HBM-side traffic of one UNet plan (bench workload, B=2 with the shared CFG prefix).

Pass 1/2 (on the GPU box, under rocprofv3 --pmc FETCH_SIZE and again --pmc WRITE_SIZE):
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 tools/plan_traffic.py run $OUT/steps.json
  runs every recorded launch of the plan ONCE, in order, eagerly, with a torch cos_ launch between steps as a separator (three in
  a row mark the start), and writes the step descriptions + their algorithmic byte counts to steps.json.
Join:
    python3 tools/plan_traffic.py join $OUT > $OUT/plan_traffic.txt
  reads the two counter_collection.csv files, cuts them at the separators and prints fetched / written bytes per step next to the
  algorithmic bytes (A once + weights once + residual once; output once).  FETCH_SIZE x2 per MI355X_MICROARCH.md (HBM)."""
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


SEP = "cos_kernel"


def describe(s):
    fn, kw = s.func.__name__, s.keywords
    rd = wr = 0
    if fn == "gemm":
        pw = s.args[1]
        M, N, K = kw["M"], pw.N, pw.w.shape[1]
        mode = {0: "lin", 1: "conv", 2: "tconv"}[kw.get("mode", 0)]
        taps = {0: 1, 1: 9, 2: 3}[kw.get("mode", 0)]
        key = f"gemm {mode:5s} M={M:6d} N={N:5d} K={K:5d}" + (" geglu" if pw.geglu else "") + \
              (" +res" if kw.get("residual") is not None else "") + (" +rowadd" if kw.get("rowadd") is not None else "") + \
              (f" splits={kw['splits']}" if kw.get("splits", 1) > 1 else "") + \
              (" +rowsum" if kw.get("rowsum") is not None else "") + (" lnfold" if kw.get("lnfold") is not None else "") + \
              (" +gstat" if kw.get("gstat") is not None else "") + (" +tattn" if kw.get("tattn") is not None else "")
        n_out = N // 2 if pw.geglu else N
        rd = 2 * M * (K // taps) + 2 * N * K + (2 * M * n_out if kw.get("residual") is not None else 0)
        wr = (4 if kw.get("out_f32") else 2) * M * n_out
    elif fn.startswith("groupnorm"):
        key = f"{fn} F={kw['F']} HW={kw['HW']} C={kw['Cn']} fps={kw['frames_per_stat']}"
        n = kw["F"] * kw["HW"] * kw["Cn"]
        rd, wr = 2 * n, 2 * n
    elif fn == "attention":
        key = f"attention Bq={kw['Bq']} h={kw['heads']} Nq={kw['Nq']} Nk={kw['Nk']}"
        rd = 2 * kw["Bq"] * kw["heads"] * 64 * (kw["Nq"] + 2 * kw["Nk"])
        wr = 2 * kw["Bq"] * kw["heads"] * 64 * kw["Nq"]
    else:
        key = fn
    return key, rd, wr


def run(out):
    import torch
    import bench
    from moca_video_amd import ops
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dm = bench.build_model(dev, seed=321)
    unet = dm.model.diffusion_model
    g = torch.Generator(device=dev).manual_seed(1)
    n = int(os.environ.get("PT_N", "1"))            # latents: B = 2 n videos with the shared CFG prefix (PT_N=8: the FIFO / configs[4] batch)
    x = torch.randn(n, 4, 16, 40, 64, device=dev, generator=g)
    ctx = torch.randn(2 * n, 77, 1024, device=dev, generator=g)
    ts = torch.full((n,), 500, device=dev, dtype=torch.long)
    with torch.no_grad():
        for _ in range(3):
            unet.forward_segments(x, ts, [ctx[:n], ctx[n:]], fps=torch.tensor([10] * n, device=dev), shared_x=True)
    torch.cuda.synchronize()
    plan = next(iter(unet._plans.values()))
    st = plan.stream
    sep = torch.zeros(977, device=dev)
    ops.set_stream(st.cuda_stream)
    desc = []
    with torch.cuda.stream(st):
        sep.cos_(); sep.cos_(); sep.cos_()
        for s in plan.steps:
            s()
            sep.cos_()
            desc.append(describe(s))
        sep.cos_(); sep.cos_()
    torch.cuda.synchronize()
    ops.set_stream(None)
    json.dump(desc, open(out, "w"))
    print(len(desc), "steps")


def per_step(root, sub, n_steps):
    """counter value per step, cut at the separator launches"""
    rows = []
    for f in glob.glob(f"{root}/{sub}/**/*counter_collection.csv", recursive=True):
        rows += list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    sep = [i for i, r in enumerate(rows) if SEP in r["Kernel_Name"]]
    start = None
    for j in range(len(sep) - 2):                # the start marker: three separators in a row
        if sep[j + 1] == sep[j] + 1 and sep[j + 2] == sep[j] + 2:
            start = sep[j + 2] + 1
            break
    assert start is not None, "start marker not found"
    out, cur, names = [], 0.0, []
    for r in rows[start:]:
        if SEP in r["Kernel_Name"]:
            out.append((cur, names))
            cur, names = 0.0, []
            if len(out) == n_steps:
                break
        else:
            cur += float(r["Counter_Value"])
            names.append(r["Kernel_Name"].split("(")[0][:40])
    assert len(out) == n_steps, (len(out), n_steps)
    return out


def join(root):
    desc = json.load(open(f"{root}/steps.json"))
    fe = per_step(root, "fetch", len(desc))
    wr = per_step(root, "write", len(desc))
    agg = collections.OrderedDict()
    for (key, rd, w), (f, names), (wv, _) in zip(desc, fe, wr):
        a = agg.setdefault(key, [0, 0.0, 0.0, 0, 0, set()])
        a[0] += 1; a[1] += 2 * f * 1024; a[2] += wv * 1024; a[3] += rd; a[4] += w; a[5].update(names)
    tf, tw, ta = sum(a[1] for a in agg.values()), sum(a[2] for a in agg.values()), sum(a[3] + a[4] for a in agg.values())
    print(f"# one B={2 * int(os.environ.get('PT_N', '1'))} shared-prefix forward, every launch once, eager: fetched {tf / 1e9:.2f} GB (FETCH_SIZE x2) + written {tw / 1e9:.2f} GB;"
          f" algorithmic (operands once) {ta / 1e9:.2f} GB")
    print(f"{'n':>3s} {'fetch MB':>9s} {'alg rd MB':>9s} {'x':>5s} {'write MB':>9s} {'alg wr MB':>9s} {'x':>5s} {'excess MB':>9s}  step")
    for key, (n, f, w, rd, aw, names) in sorted(agg.items(), key=lambda kv: -(kv[1][1] + kv[1][2] - kv[1][3] - kv[1][4])):
        print(f"{n:3d} {f / 1e6:9.1f} {rd / 1e6:9.1f} {f / max(rd, 1):5.2f} {w / 1e6:9.1f} {aw / 1e6:9.1f} {w / max(aw, 1):5.2f} "
              f"{(f + w - rd - aw) / 1e6:9.1f}  {key}")


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(sys.argv[2])
    else:
        join(sys.argv[2])
