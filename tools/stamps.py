#!/usr/bin/env python3
"""Diagnostic: per-block s_memtime stamps of gemm_glds_kernel (library built with -DMOCA_STAMPS).
   MOCA_HIP_LIB=tools/diag/libmoca_hip_stamps.so python tools/stamps.py <shape>
Prints the mean duration (us at 100 MHz s_memtime ticks -> converted with the measured ratio) of:
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
          "conv2": lambda: B.conv(2, 1280, 1280), "conv1": lambda: B.conv(1, 640, 640), "qkv1": lambda: B.linear(1, 640, 1920)}
dt = shapes[which]()
torch.cuda.synchronize()
lib = L.load()
lib.moca_debug_stamps.restype = C.c_int
lib.moca_debug_stamps.argtypes = [C.c_void_p, C.c_int]
nb = 16384
buf = np.zeros((nb, 8), dtype=np.uint64)
assert lib.moca_debug_stamps(buf.ctypes.data_as(C.c_void_p), nb) == 0
ok = buf[:, 5] > 0
s = buf[ok].astype(np.int64)
print("blocks stamped:", ok.sum())
tick_us = 1e-2          # s_memtime counts at 100 MHz on gfx950 (REFCLK); verified against kernel time below
span = (s[:, 5].max() - s[:, 0].min()) * tick_us
print(f"kernel span from stamps: {span:.1f} us (host-timed {dt*1e6:.1f} us)")
names = ["start->issued", "issued->tile0 landed", "main loop", "stage1+sync", "stage2 (stores)"]
for i, n in enumerate(names):
    d = (s[:, i + 1] - s[:, i]) * tick_us
    print(f"{n:24s} mean {d.mean():7.2f} us  p50 {np.median(d):7.2f}  p90 {np.percentile(d, 90):7.2f}")
tot = (s[:, 5] - s[:, 0]) * tick_us
print(f"{'block total':24s} mean {tot.mean():7.2f} us")
# gaps between consecutive blocks on the same CU
cu = (s[:, 7] << 16) | (s[:, 6] & 0xFF00) | ((s[:, 6] >> 13) & 7) << 4   # xcc, cu_id/sh, se
gaps = []
for c in np.unique(cu):
    r = s[cu == c]
    r = r[np.argsort(r[:, 0])]
    gaps += list((r[1:, 0] - r[:-1, 5]) * tick_us)
gaps = np.array(gaps)
print(f"CUs seen {len(np.unique(cu))}; gap end(prev)->start(next) on a CU: mean {gaps.mean():.2f} us  p50 {np.median(gaps):.2f}  p90 {np.percentile(gaps,90):.2f}")
