"""Shared test helpers: deterministic inputs (same recipe as tools/make_golden.py) and golden loading."""
import os

import numpy as np
import torch

from moca_video_amd.weightgen import gen_state_dict, gen_tensor

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

REDUCED = dict(in_channels=4, out_channels=4, model_channels=64, attention_resolutions=[4, 2, 1], num_res_blocks=2,
               channel_mult=[1, 2, 4, 4], num_head_channels=64, transformer_depth=1, context_dim=128, use_linear=True,
               use_checkpoint=False, temporal_conv=True, temporal_attention=True, temporal_selfatt_only=True,
               use_relative_position=False, use_causal_attention=False, temporal_length=16, addition_attention=True,
               fps_cond=True)

FULL = dict(in_channels=4, out_channels=4, model_channels=320, attention_resolutions=[4, 2, 1], num_res_blocks=2,
            channel_mult=[1, 2, 4, 4], num_head_channels=64, transformer_depth=1, context_dim=1024, use_linear=True,
            use_checkpoint=True, temporal_conv=True, temporal_attention=True, temporal_selfatt_only=True,
            use_relative_position=False, use_causal_attention=False, temporal_length=16, addition_attention=True,
            fps_cond=True)


def inp(name, shape, seed=0):
    t = gen_tensor("input:" + name, (int(np.prod(shape)),), seed)
    return (t * 10.0).reshape(shape)


def golden(name):
    return np.load(os.path.join(GOLD, name + ".npz"))


def state_dict_for(module, seed):
    """weightgen parameters for a module built by OUR classes (same keys/shapes as the reference)."""
    return gen_state_dict({k: v.shape for k, v in module.state_dict().items()}, seed)


def relerr(got, ref):
    """max|got - ref| / max|ref|.  With MOCA_ERRLOG=<file> every value is appended with the running test's id (how the
    tolerances in the GPU tests were set: <= 1.5 x the largest value observed there)."""
    got = torch.as_tensor(got).float()
    ref = torch.as_tensor(ref).float()
    e = ((got - ref).abs().max() / ref.abs().max().clamp_min(1e-12)).item()
    log = os.environ.get("MOCA_ERRLOG")
    if log:
        with open(log, "a") as f:
            f.write(f"{e:.3e} {os.environ.get('PYTEST_CURRENT_TEST', '?')}\n")
    return e


def loop_sam_candidates(call, F, H, W):
    """Scripted Grounded-SAM-2 output for the `call`-th `ddim_step` of a sampling loop, one entry per frame (None = no box
    detected): tools/make_golden.py::loop_cases feeds these to the REAL reference through fake predictor objects, the tests
    hand the same lists to the oracle / the HIP path as `sam_masks`.  Cycles through: a detection, a shifted detection (IoU
    fallback), no detection, a > 80 % mask followed by a small one, and an empty list."""
    def rect(y0, y1, x0, x1):
        m = torch.zeros(H, W)
        m[y0 % H:max(y0 % H + 1, y1 % (H + 1)), x0 % W:max(x0 % W + 1, x1 % (W + 1))] = 1.0
        return m
    big = torch.ones(H, W)
    out = []
    for i in range(F):
        k = (call * 3 + i) % 5
        if k == 0:
            out.append(rect(2 + call, 10 + call, 3, 12)[None])
        elif k == 1:
            out.append(rect(2 + call, 10 + call, 4 + i, 12 + i)[None])
        elif k == 2:
            out.append(None)
        elif k == 3:
            out.append(torch.stack([big, rect(1, 5 + i, 1 + call, 6 + call)]))
        else:
            out.append(rect(11, 15, call % 4, 5 + call % 4)[None])
    return out
