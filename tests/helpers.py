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
    got = torch.as_tensor(got).float()
    ref = torch.as_tensor(ref).float()
    return ((got - ref).abs().max() / ref.abs().max().clamp_min(1e-12)).item()
