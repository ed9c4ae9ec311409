#!/usr/bin/env python3
"""Which shader clock does the chip hold INSIDE the B = 2 DDIM-step graph, launch by launch?  (the power-cap
argument of DESIGN 4.6 rested on rocm-smi readings while ONE shape looped for seconds; rocm-smi samples at ~1 Hz and cannot see a
300 us launch inside a 33 ms step.)

Run under the kernel tracer, then summarise:
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r06/cis -- python3 tools/clock_in_step.py run gpurun_out/r06/cis
    python3 tools/clock_in_step.py report gpurun_out/r06/cis > profiles/r06_clock_in_step.txt

`run`: builds the bench's B = 2 one-graph DDIM step (bench.build_model + fifo_graph.BaseEngine), replays it, and WHILE `steps` more
replays run, 8 one-wave sampler blocks (moca_debug_clock_sampler, one per XCD) record (shader cycles, 100 MHz real time) pairs every
~8 us on a side stream.  `report`: maps the samples onto the traced launches (the sampler's own start / end in the trace calibrate its
real-time counter against the tracer's clock) and prints, per kernel class and for the largest launches, the clock held while they ran."""
import csv
import glob
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(out_dir, steps=6, nsamples=40000):
    import ctypes as C
    import numpy as np
    import torch
    import bench
    from moca_video_amd import lib as mlib
    from moca_video_amd.fifo_graph import BaseEngine
    from moca_video_amd.sampler import DDIMSampler
    device = torch.device("cuda", 0)
    torch.cuda.set_device(device)
    lib = mlib.load()
    dm = bench.build_model(device, seed=321)
    sampler = DDIMSampler(dm)
    sampler.make_schedule(50, ddim_eta=1.0, verbose=False)
    g = torch.Generator(device=device).manual_seed(321)
    x = torch.randn(1, 4, 16, 40, 64, device=device, generator=g)
    fps = torch.tensor([10], device=device)
    cond = {"c_crossattn": [torch.randn(1, 77, 1024, device=device, generator=g)], "fps": fps}
    uc = {"c_crossattn": [torch.randn(1, 77, 1024, device=device, generator=g)], "fps": fps}
    eng = BaseEngine(dm, sampler, x, cond, uc, 12.0, seed=321)
    cur = torch.cuda.current_stream(device)
    for _ in range(6):                       # eager, capture, replays (clocks settle)
        eng.step()
        cur.wait_stream(eng.plan.stream)
    torch.cuda.synchronize()
    assert eng.plan.graph is not None
    side = torch.cuda.Stream(device)
    buf = torch.zeros(8, nsamples, 2, dtype=torch.int64, device=device)
    stop = torch.zeros(1, dtype=torch.int32, device=device)
    torch.cuda.synchronize()
    mlib.check(lib.moca_debug_clock_sampler(C.c_void_p(buf.data_ptr()), 8, nsamples, C.c_void_p(stop.data_ptr()), C.c_void_p(side.cuda_stream)),
               "moca_debug_clock_sampler")
    for _ in range(steps):
        eng.step()
        cur.wait_stream(eng.plan.stream)
    cur.synchronize()
    stop.fill_(1)                            # (the sampler leaves at its next sample)
    torch.cuda.synchronize()
    os.makedirs(out_dir, exist_ok=True)
    np.save(os.path.join(out_dir, "clock_samples.npy"), buf.cpu().numpy())
    print("samples written", flush=True)


def classify(n):
    for pat, k in (("clock_sampler", None), ("attention_v4", "attention_v4 (spatial self-attention, long keys)"), ("attention_short", "attention_short (77-token context)"),
                   ("temporal_attention", "temporal attention"), ("gemm_sqp", "gemm_sqp (GEGLU, persistent 256x256)"), ("gemm_g4", "gemm_g4 / g4p (GEGLU 256x128)"),
                   ("gemm_ws", "gemm_ws (weight-stationary 320->320)"), ("gemm_glds", "gemm_glds (256-row, 1280-channel levels)"), ("splitk", "split-K reduce (+GroupNorm)"),
                   ("gemm_f16", "gemm_small")):
        if pat in n:
            return k
    if "gemm_w80s" in n:
        if re.search(r"ELi3EEE|, 3>", n):
            return "gemm_w80s 320x192 (q|k|v + temporal attention)"
        if re.search(r"ELi1EEE|, 1>", n):
            return "gemm_w80s 160x320 (linears)"
        return "gemm_w80s 320x160 (convs)"
    if "gn_" in n or "layernorm" in n or "gstat" in n:
        return "GroupNorm / LayerNorm passes"
    return "other"


def report(out_dir):
    import numpy as np
    f = sorted(glob.glob(os.path.join(out_dir, "**", "*kernel_trace.csv"), recursive=True))
    assert f, "no kernel trace under " + out_dir
    rows = list(csv.DictReader(open(f[-1])))
    ks = [(r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Grid_Size", r.get("Grid_Size_X", "?"))) for r in rows]
    smp = [k for k in ks if "clock_sampler" in k[0]]
    assert len(smp) == 1, f"{len(smp)} sampler launches in the trace"
    _, s0, s1, _ = smp[0]
    buf = np.load(os.path.join(out_dir, "clock_samples.npy"))              # [8][n][2]
    series = []
    for b in range(buf.shape[0]):
        t, r = buf[b, :, 0], buf[b, :, 1]
        n = int((r > 0).sum())
        series.append((t[:n].astype(np.float64), r[:n].astype(np.float64)))
    # calibrate: first sample ~ kernel start, last ~ kernel end; the real-time counter runs at 100 MHz (10 ns per tick)
    r_first = min(s[1][0] for s in series)
    r_last = max(s[1][-1] for s in series)
    ns_per_tick = (s1 - s0) / (r_last - r_first)
    to_ns = lambda r: s0 + (r - r_first) * ns_per_tick
    print(f"# tools/clock_in_step.py: shader clock held INSIDE the B = 2 DDIM-step graph (bench.py's workload, random-init UNet), launch by launch.")
    print(f"# 8 sampler blocks (one per XCD), {len(series[0][0])} samples each, one every {np.median(np.diff(series[0][1])) * ns_per_tick / 1e3:.1f} us; "
          f"real-time counter: {ns_per_tick:.3f} ns per tick against the tracer's clock (nominal 10).")
    print("# clock of an interval = d(s_memtime) / d(s_memrealtime) x 100 MHz; a launch's clock = cycles / time over the sample intervals whose "
          "midpoint lies inside it, mean over the 8 XCDs.  (Tracing inflates durations ~1.5 %; it does not change what the clock does.)")
    mids, clk = [], []
    for t, r in series:
        dr, dt = np.diff(r), np.diff(t)
        ok = dr > 0
        mids.append(to_ns((r[:-1] + r[1:]) / 2)[ok])
        clk.append((dt / dr * 100.0)[ok])                                   # MHz
    win = [k for k in ks if "clock_sampler" not in k[0] and k[1] >= s0 and k[2] <= s1]
    if not win:
        raise SystemExit("no launch overlaps the sampler")
    t_lo, t_hi = min(k[1] for k in win), max(k[2] for k in win)
    allm, allc = np.concatenate(mids), np.concatenate(clk)
    inside = (allm >= t_lo) & (allm <= t_hi)
    print(f"\nwhole window ({(t_hi - t_lo) / 1e6:.1f} ms of graph replays, {len(win)} launches): mean clock {allc[inside].mean():.0f} MHz, "
          f"p10 {np.percentile(allc[inside], 10):.0f}, p50 {np.percentile(allc[inside], 50):.0f}, p90 {np.percentile(allc[inside], 90):.0f}; "
          f"per XCD mean: " + " ".join(f"{c[(m >= t_lo) & (m <= t_hi)].mean():.0f}" for m, c in zip(mids, clk)))
    order = np.argsort(allm)
    am, ac = allm[order], allc[order]

    def clock_of(a, b):
        i, j = np.searchsorted(am, a), np.searchsorted(am, b)
        return (ac[i:j].mean(), j - i) if j > i else (float("nan"), 0)
    cls = {}
    for name, a, b, grid in win:
        c = classify(name)
        if c is None:
            continue
        mhz, n = clock_of(a, b)
        d = cls.setdefault(c, [0.0, 0, 0.0, 0])
        d[0] += (b - a); d[1] += 1
        if n:
            d[2] += mhz * n; d[3] += n
    print(f"\n{'kernel class':58s} {'launches':>8s} {'time ms':>8s} {'clock MHz':>10s}   (sample intervals)")
    for c, d in sorted(cls.items(), key=lambda kv: -kv[1][0]):
        print(f"{c:58s} {d[1]:8d} {d[0] / 1e6:8.2f} {d[2] / max(d[3], 1):10.0f}   ({d[3]})")
    print(f"\nthe 14 longest launches of the window:")
    for name, a, b, grid in sorted(win, key=lambda k: -(k[2] - k[1]))[:14]:
        mhz, n = clock_of(a, b)
        short = re.sub(r"\(anonymous namespace\)::|_ZN12_GLOBAL__N_1\d+", "", name)[:60]
        print(f"  {(b - a) / 1e3:8.1f} us  {mhz:6.0f} MHz ({n:3d} intervals)  grid {grid:>8s}  {short}")
    # coarse time series over one replay
    print("\nclock over the window, 0.5 ms bins (MHz):")
    nb = int((t_hi - t_lo) / 5e5) + 1
    line = []
    for i in range(nb):
        mhz, n = clock_of(t_lo + i * 5e5, t_lo + (i + 1) * 5e5)
        line.append(f"{mhz:.0f}" if n else "-")
    for i in range(0, len(line), 20):
        print(f"  {i * 0.5:6.1f} ms: " + " ".join(line[i:i + 20]))


if __name__ == "__main__":
    if len(sys.argv) >= 3 and sys.argv[1] == "run":
        run(sys.argv[2])
    elif len(sys.argv) >= 3 and sys.argv[1] == "report":
        report(sys.argv[2])
    else:
        raise SystemExit(__doc__)
