#!/usr/bin/env python3
"""Run a few named GEMM shapes a handful of times (for rocprofv3 --pmc / --kernel-trace)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench_gemm as B
B.ops.set_stream(None)
which = sys.argv[1:] or ["conv0", "lin2"]
for w in which:
    if w == "conv0": B.conv(0, 320, 320)
    elif w == "conv1": B.conv(1, 640, 640)
    elif w == "conv2": B.conv(2, 1280, 1280)
    elif w == "lin2": B.linear(2, 5120, 1280)
    elif w == "geglu0": B.linear(0, 320, 1280, geglu=True)
    elif w == "lin0": B.linear(0, 320, 320)
torch.cuda.synchronize()
