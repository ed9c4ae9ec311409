#!/usr/bin/env python3
"""Micro-benchmark of moca_gemm_f16 on the UNet's real shapes (B=2 videos, 16 frames, 40x64 latents).
Prints TFLOP/s per shape; used to steer kernel tuning.  GPU only."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from moca_video_amd import ops, lib as L

DEV = "cuda"
ZERO = os.environ.get("BG_ZERO") == "1"      # all-zero operands: how much of a kernel's time is the power/clock limit
if ZERO:
    _randn = torch.randn
    torch.randn = lambda *a, **k: torch.zeros(*a, **k)
B, T = int(os.environ.get("BG_B", "2")), 16
for kv in os.environ.get("BG_TUNE", "").split(","):      # e.g. BG_TUNE=2:2,1:0  (MOCA_TUNE_* knob:value)
    if kv:
        L.set_tuning(int(kv.split(":")[0]), int(kv.split(":")[1]))
F = B * T
LV = {0: (40, 64), 1: (20, 32), 2: (10, 16), 3: (5, 8)}


FILTER = [a for a in sys.argv[1:]]


def run(name, fn, flops, iters=int(os.environ.get("BG_ITERS", "20"))):
    if FILTER and not any(f in name for f in FILTER):
        return 0.0
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    print(f"{name:46s} {dt*1e6:9.1f} us  {flops/dt/1e12:8.1f} TF/s", flush=True)
    return dt


def splits_for(M, pw):
    from moca_video_amd.plan import gemm_splits
    return gemm_splits(M, pw)


def conv(lv, cin, cout, splits=None):
    H, W = LV[lv]
    x = torch.randn(F * H * W, cin, device=DEV).half()
    w = torch.randn(cout, cin, 3, 3, device=DEV) * (9 * cin) ** -0.5
    pw = ops.pack_conv3x3(w, torch.zeros(cout, device=DEV))
    M = F * H * W
    out = torch.empty(M, pw.N, device=DEV, dtype=torch.float16)
    s = splits_for(M, pw) if splits is None else splits
    ws = torch.empty(s * M * pw.N, device=DEV, dtype=torch.float32) if s > 1 else None
    fn = lambda: ops.gemm(x, pw, out, M=M, mode=L.MOCA_A_CONV3X3, conv=(cin, H, W, H, W, 1, 0), splits=s, splitk_ws=ws)
    return run(f"conv3x3 L{lv} {cin}->{cout} M={M} s={s}", fn, 2.0 * M * cout * 9 * cin)


def tconv(lv, c, splits=None):
    H, W = LV[lv]
    M = F * H * W
    x = torch.randn(M, c, device=DEV).half()
    w = torch.randn(c, c, 3, 1, 1, device=DEV) * (3 * c) ** -0.5
    pw = ops.pack_tconv3(w, torch.zeros(c, device=DEV))
    out = torch.empty(M, pw.N, device=DEV, dtype=torch.float16)
    s = splits_for(M, pw) if splits is None else splits
    ws = torch.empty(s * M * pw.N, device=DEV, dtype=torch.float32) if s > 1 else None
    fn = lambda: ops.gemm(x, pw, out, M=M, mode=L.MOCA_A_TCONV3, tconv=(c, T, H * W), splits=s, splitk_ws=ws)
    return run(f"tconv3   L{lv} {c} M={M} s={s}", fn, 2.0 * M * c * 3 * c)


def linear(lv, k, n, geglu=False, splits=None):
    H, W = LV[lv]
    M = F * H * W
    x = torch.randn(M, k, device=DEV).half()
    if geglu:
        pw = ops.pack_geglu(torch.randn(2 * n, k, device=DEV) * k ** -0.5, torch.zeros(2 * n, device=DEV))
        nn = 2 * n
    else:
        pw = ops.pack_linear(torch.randn(n, k, device=DEV) * k ** -0.5, torch.zeros(n, device=DEV))
        nn = n
    out = torch.empty(M, n, device=DEV, dtype=torch.float16)
    s = splits_for(M, pw) if splits is None else splits
    ws = torch.empty(s * M * pw.N, device=DEV, dtype=torch.float32) if s > 1 else None
    fn = lambda: ops.gemm(x, pw, out, M=M, splits=s, splitk_ws=ws)
    return run(f"linear   L{lv} {k}->{nn}{' geglu' if geglu else ''} M={M} s={s}", fn, 2.0 * M * nn * k)


def tattn(lv, k, heads):
    """to_q|to_k|to_v per head + temporal attention in the epilogue (MOCA_EP_TATTN, the 320 x 192 tiling)"""
    H, W = LV[lv]; M = F * H * W
    x = torch.randn(M, k, device=DEV).half()
    C = heads * 64
    wq, wk, wv = (torch.randn(C, k, device=DEV) * k ** -0.5 for _ in range(3))
    pw = ops.pack_qkv_per_head(wq, wk, wv, heads)
    out = torch.empty(M, C, device=DEV, dtype=torch.float16)
    return run(f"qkv+tattn L{lv} {k}->{3 * C} M={M}", lambda: ops.gemm(x, pw, out, M=M, tattn=(T, H * W, 0.125)), 2.0 * M * 3 * C * k)


def linear_res(lv, k, n):
    """with residual (as in the model: attention out-proj, ff2, proj_out all add a residual)"""
    H, W = LV[lv]; M = F * H * W
    x = torch.randn(M, k, device=DEV).half()
    pw = ops.pack_linear(torch.randn(n, k, device=DEV) * k ** -0.5, torch.zeros(n, device=DEV))
    out = torch.empty(M, n, device=DEV, dtype=torch.float16); res = torch.randn(M, n, device=DEV).half()
    return run(f"linear+res L{lv} {k}->{n} M={M}", lambda: ops.gemm(x, pw, out, M=M, residual=res), 2.0 * M * n * k)


if __name__ == "__main__":
    ops.set_stream(None)
    if os.environ.get("BG_PROBE") == "tattn":  # the fused q|k|v + temporal attention launches of the four levels
        tattn(0, 320, 5); tattn(1, 640, 10); tattn(2, 1280, 20); tattn(3, 1280, 20)
        sys.exit(0)
    if os.environ.get("BG_PROBE") == "1":     # shapes outside the UNet: what the persistent register-epilogue kernel (knob 7:2) does on short-K, HBM-bound linears
        for n in (256, 512):
            linear(0, 320, n); linear_res(0, 320, n); linear_res(0, 1280, n); linear_res(1, 640, n)
        linear(0, 320, 320); linear_res(0, 320, 320); linear_res(0, 1280, 320); linear_res(1, 640, 640)
        sys.exit(0)
    conv(0, 320, 320); conv(0, 640, 320); conv(0, 960, 320)
    conv(1, 320, 640); conv(1, 640, 640); conv(1, 1280, 640); conv(1, 1920, 640)
    conv(2, 640, 1280); conv(2, 1280, 1280); conv(2, 2560, 1280); conv(2, 1920, 1280)
    conv(3, 1280, 1280); conv(3, 2560, 1280)
    tconv(0, 320); tconv(1, 640); tconv(2, 1280); tconv(3, 1280)
    linear(0, 320, 320); linear(0, 320, 960); linear(0, 320, 1280, geglu=True); linear(0, 1280, 320)
    linear(1, 640, 640); linear(1, 640, 1920); linear(1, 640, 2560, geglu=True); linear(1, 2560, 640)
    linear(2, 1280, 1280); linear(2, 1280, 3840); linear(2, 1280, 5120, geglu=True); linear(2, 5120, 1280)
    linear(3, 1280, 3840); linear(3, 1280, 5120, geglu=True); linear(3, 5120, 1280)
    linear_res(0, 320, 320); linear_res(0, 1280, 320); linear_res(1, 640, 640); linear_res(1, 2560, 640)
