"""Deterministic synthetic weights  w = f(state-dict key, shape, seed).

No VideoCrafter2 checkpoint exists offline, and a random-init reference UNet outputs
exactly 0 (every zero_module'd conv / proj_out: openaimodel3d.py:177,266-267,531,
attention.py:256-258,326-328).  This generator gives every tensor -- zero-initialised
ones included -- a variance-preserving random value from a counter-based numpy Philox
stream keyed by the tensor's NAME, so the reference module here, the CPU oracle and the
HIP path on the GPU box all see bit-identical parameters without shipping 5.6 GB.
"""
from __future__ import annotations

import hashlib
from collections import OrderedDict

import numpy as np
import torch


def _rng(key: str, seed: int):
    h = hashlib.sha256(f"{seed}:{key}".encode()).digest()
    return np.random.Generator(np.random.Philox(key=int.from_bytes(h[:16], "little")))


def gen_tensor(key: str, shape, seed: int = 0) -> torch.Tensor:
    shape = tuple(int(s) for s in shape)
    z = _rng(key, seed).standard_normal(shape, dtype=np.float32)
    if len(shape) <= 1:
        if key.endswith("weight"):          # GroupNorm / LayerNorm scale
            z = 1.0 + 0.1 * z
        else:                               # any bias
            z = 0.1 * z
    else:
        fan_in = int(np.prod(shape[1:]))
        z = z * np.float32(fan_in ** -0.5)
    return torch.from_numpy(np.asarray(z, dtype=np.float32))


def gen_state_dict(shapes, seed: int = 0) -> "OrderedDict[str, torch.Tensor]":
    """shapes: mapping name -> shape (e.g. {k: v.shape for k, v in module.state_dict().items()})"""
    return OrderedDict((k, gen_tensor(k, s, seed)) for k, s in shapes.items())


def fill_module_(module: torch.nn.Module, seed: int = 0):
    """In-place fill of every parameter of `module` (ours or the reference's: same keys)."""
    sd = module.state_dict()
    new = gen_state_dict({k: v.shape for k, v in sd.items()}, seed)
    module.load_state_dict(new, strict=True)
    return module


@torch.no_grad()
def init_random_(module: torch.nn.Module, seed: int = 0):
    """Fast on-device random init with the same per-tensor statistics as gen_tensor (torch RNG:
    NOT bit-identical to gen_tensor; for benchmarks, where ranks then share rank 0's weights by
    RCCL broadcast).  The architecture is the reference's; VideoCrafter2 weights do not exist offline."""
    g = torch.Generator(device=next(module.parameters()).device)
    g.manual_seed(seed)
    for name, p in module.named_parameters():
        if p.dim() == 1:
            p.normal_(0.0, 0.1, generator=g)
            if name.endswith("weight"):
                p.add_(1.0)
        else:
            fan_in = 1
            for s in p.shape[1:]:
                fan_in *= int(s)
            p.normal_(0.0, fan_in ** -0.5, generator=g)
    if hasattr(module, "_invalidate"):
        module._invalidate()
    for m in module.modules():
        if hasattr(m, "_invalidate"):
            m._invalidate()
    return module
