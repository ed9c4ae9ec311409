"""MI355X mirror of `utils/freeinit_utils.py`: `freq_mix_3d` (:7-47) and `get_freq_filter`
(:51-70) with its four filter builders (:73-156) on the fp32 HIP kernels of
csrc/freeinit.hip.  The reference builds each filter with a T*H*W-iteration Python loop on
every call (funcs.py:95); here it is a closed-form kernel and results are cached per
(shape, type, n, d_s, d_t, device)."""
from __future__ import annotations

import ctypes as C

import torch

from . import lib as _l
from . import ops

_FILTER_TYPES = {"gaussian": 0, "butterworth": 1, "ideal": 2, "box": 3}
_filter_cache = {}


def _st():
    return C.c_void_p(ops.current_stream())


def get_freq_filter(shape, device, filter_type, n, d_s, d_t):
    """freeinit_utils.py:51-70.  Returns a tensor of `shape` (B, C, T, H, W) like the reference."""
    if filter_type not in _FILTER_TYPES:
        raise NotImplementedError
    device = torch.device(device)
    if device.type != "cuda":
        raise RuntimeError("moca_video_amd.get_freq_filter runs on the GPU only (no CPU path)")
    shape = tuple(int(s) for s in shape)
    key = (shape, filter_type, int(n), float(d_s), float(d_t), device.index)
    hit = _filter_cache.get(key)
    if hit is not None:
        return hit
    T, H, W = shape[-3], shape[-2], shape[-1]
    vol = torch.empty(T, H, W, dtype=torch.float32, device=device)
    _l.check(_l.load().moca_freq_filter_f32(_l.ptr(vol), T, H, W, _FILTER_TYPES[filter_type], int(n), float(d_s),
                                            float(d_t), _st()), "moca_freq_filter_f32")
    out = vol.expand(shape)          # `mask[..., t, h, w] = v` broadcasts over the leading dims (:89)
    _filter_cache[key] = out
    return out


def freq_mix_3d(x, noise, LPF):
    """freeinit_utils.py:7-47: fftn -> fftshift -> low/high-pass blend -> ifftshift -> ifftn.real,
    over the last three dims, after squeeze(0) (:24-25)."""
    original_dtype = x.dtype
    if not x.is_cuda:
        raise RuntimeError("moca_video_amd.freq_mix_3d runs on the GPU only (no CPU path)")
    xs = x.to(torch.float32).squeeze(0)
    ns = noise.to(torch.float32).squeeze(0)
    out_shape = torch.broadcast_shapes(xs.shape, tuple(LPF.shape))   # x_freq * LPF broadcasts against the 5-D filter (:34-36)
    T, H, W = xs.shape[-3:]
    lpf = LPF.to(torch.float32)
    if lpf.dim() > 3:
        # the reference filters are constant over the leading dims (built by `mask[..., t, h, w] = v`)
        lpf = lpf.reshape(-1, T, H, W)[0]
    lpf = lpf.contiguous()
    xs = xs.reshape(-1, T, H, W).contiguous()
    ns = ns.reshape(-1, T, H, W).contiguous()
    Cn = xs.shape[0]
    out = torch.empty_like(xs)
    lib = _l.load()
    ws = torch.empty(int(lib.moca_freq_mix_ws_bytes(Cn, T, H, W)) // 4, dtype=torch.float32, device=x.device)
    _l.check(lib.moca_freq_mix_3d_f32(_l.ptr(xs), _l.ptr(ns), _l.ptr(lpf), _l.ptr(out), Cn, T, H, W, _l.ptr(ws), _st()),
             "moca_freq_mix_3d_f32")
    out = out.reshape(x.to(torch.float32).squeeze(0).shape).expand(out_shape)
    if original_dtype != torch.float32:
        out = out.to(original_dtype)
    return out
