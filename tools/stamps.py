#!/usr/bin/env python3
"""Diagnostic: per-block s_memtime stamps of the direct-to-LDS GEMM kernels (glds, g4, w80s) (library built with -DMOCA_STAMPS).
   MOCA_HIP_LIB=tools/diag/libmoca_hip_stamps.so python tools/stamps.py <shape>
Prints the mean duration in shader cycles (s_memtime) of:
prologue issue, first-tile wait, main loop, epilogue stage 1, stage 2, and the gap between consecutive
blocks on the same CU."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench_gemm as B
from moca_video_amd import lib as L
B.ops.set_stream(None)
B.FILTER = []          # (bench_gemm filters by argv; this tool passes a shape key instead)
which = sys.argv[1] if len(sys.argv) > 1 else "qkv0"
shapes = {"qkv0": lambda: B.linear(0, 320, 960), "geglu0": lambda: B.linear(0, 320, 1280, geglu=True),
          "lin0": lambda: B.linear(0, 320, 320), "ff2_0": lambda: B.linear(0, 1280, 320), "conv0": lambda: B.conv(0, 320, 320),
          "conv2": lambda: B.conv(2, 1280, 1280), "conv1": lambda: B.conv(1, 640, 640), "qkv1": lambda: B.linear(1, 640, 1920),
          "geglu1": lambda: B.linear(1, 640, 2560, geglu=True), "linres0": lambda: B.linear_res(0, 320, 320),
          "ff2res0": lambda: B.linear_res(0, 1280, 320), "tconv0": lambda: B.tconv(0, 320),
          "tconv3": lambda: B.tconv(3, 1280), "conv3": lambda: B.conv(3, 1280, 1280), "tconv2": lambda: B.tconv(2, 1280), "lin3": lambda: B.linear(3, 5120, 1280),
          "tattn0": lambda: B.tattn(0, 320, 5), "tattn1": lambda: B.tattn(1, 640, 10), "tattn2": lambda: B.tattn(2, 1280, 20)}
dt = shapes[which]()
torch.cuda.synchronize()
lib = L.load()
lib.moca_debug_stamps.restype = C.c_int
lib.moca_debug_stamps.argtypes = [C.c_void_p, C.c_int]
nb = 16384
NS = int(os.environ.get('STAMP_SLOTS', '16'))
buf = np.zeros((nb, NS), dtype=np.uint64)
assert lib.moca_debug_stamps(buf.ctypes.data_as(C.c_void_p), nb) == 0
ok = buf[:, 5] > 0
LAST = 5
if ok.sum() == 0:                      # split-K blocks leave through the fp32 slab store before stamps 4 / 5: report up to the end of the main loop
    ok = buf[:, 3] > 0
    LAST = 3
s = buf[ok].astype(np.int64)
print("blocks stamped:", ok.sum(), "(split-K: prologue and main loop only)" if LAST == 3 else "")
tick_us = 1.0           # s_memtime ticks are SHADER cycles on gfx950 (MI355X_MICROARCH.md): everything below is printed in cycles
span = float(s[:, LAST].max() - s[:, 0].min())
print(f"kernel span from stamps: {span:.0f} cycles; host-timed {dt*1e6:.1f} us -> {span / (dt * 1e9):.2f} GHz if the span covers the launch")
names = ["start->issued", "issued->tile0 landed", "main loop", "stage1+sync", "stage2 (stores)"]
for i, n in enumerate(names[:LAST]):
    d = (s[:, i + 1] - s[:, i]) * tick_us
    print(f"{n:24s} mean {d.mean():8.0f} cyc  p50 {np.median(d):8.0f}  p90 {np.percentile(d, 90):8.0f}")
if NS > 8 and (s[:, 15] > 0).any():
    segn = ["LOADe reads+wait", "barrier", "MFMAe (+reads)", "barrier", "LOADo issue", "LOADo vmcnt wait", "barrier"]
    ss = s[s[:, 15] > 0]
    for i, n in enumerate(segn):
        d = (ss[:, 9 + i] - ss[:, 8 + i]).astype(np.float64)
        print(f"   seg {n:20s} mean {d.mean():8.0f} cyc  p50 {np.median(d):8.0f}  p90 {np.percentile(d, 90):8.0f}")
    if (ss[:, 1] > ss[:, 15]).all():      # (sqp: slots 1 / 2 = end of MFMAo / behind its barrier)
        for n, a, b in (("MFMAo", 15, 1), ("barrier", 1, 2)):
            d = (ss[:, b] - ss[:, a]).astype(np.float64)
            print(f"   seg {n:20s} mean {d.mean():8.0f} cyc  p50 {np.median(d):8.0f}  p90 {np.percentile(d, 90):8.0f}")
if (s[:, 11] > 0).all() and (s[:, 8] > 0).all() and (s[:, 8] < s[:, 1]).all():       # glds prologue stamps (8..11)
    for n, a, b in (("  start->tile coords", 0, 8), ("  ->row descriptors", 8, 9), ("  ->W rows, frag offsets", 9, 10), ("  ->first k-tile issued", 10, 11), ("  ->second issued", 11, 1)):
        d = (s[:, b] - s[:, a]).astype(np.float64)
        print(f"{n:24s} mean {d.mean():8.0f} cyc  p50 {np.median(d):8.0f}  p90 {np.percentile(d, 90):8.0f}")
tot = (s[:, LAST] - s[:, 0]) * tick_us
print(f"{'block total':24s} mean {tot.mean():8.0f} cyc")
# gaps between consecutive blocks on the same CU
cu = (s[:, 7] << 16) | (s[:, 6] & 0xFF00) | ((s[:, 6] >> 13) & 7) << 4   # xcc, cu_id/sh, se
gaps = []
for c in np.unique(cu):
    r = s[cu == c]
    r = r[np.argsort(r[:, 0])]
    gaps += list((r[1:, 0] - r[:-1, LAST]) * tick_us)
gaps = np.array(gaps) if len(gaps) else np.zeros(1)
print(f"CUs seen {len(np.unique(cu))}; gap end(prev)->start(next) on a CU: mean {gaps.mean():.0f} cyc  p50 {np.median(gaps):.0f}  p90 {np.percentile(gaps,90):.0f}")
