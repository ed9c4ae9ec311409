"""ORACLE (test infrastructure only): CPU restatement of `/root/reference/utils/freeinit_utils.py`
(filters :73-156 vectorised in float64 numpy, freq_mix_3d :7-47 via torch.fft) and of the queue
construction in `/root/reference/scripts/evaluation/funcs.py:21-99`.  Pinned by
tests/golden/freeinit.npz and tests/golden/fifo_queue.npz (real-reference outputs)."""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.fft as fft


def _grid(T, H, W, d_s, d_t):
    t = np.arange(T, dtype=np.float64)[:, None, None]
    h = np.arange(H, dtype=np.float64)[None, :, None]
    w = np.arange(W, dtype=np.float64)[None, None, :]
    return ((d_s / d_t) * (2 * t / T - 1)) ** 2 + (2 * h / H - 1) ** 2 + (2 * w / W - 1) ** 2


def get_freq_filter(shape, filter_type, n, d_s, d_t):
    T, H, W = shape[-3], shape[-2], shape[-1]
    if d_s == 0 or d_t == 0:
        return torch.zeros(shape)
    if filter_type == "gaussian":
        m = np.exp(-1 / (2 * d_s ** 2) * _grid(T, H, W, d_s, d_t))
    elif filter_type == "butterworth":
        m = 1 / (1 + (_grid(T, H, W, d_s, d_t) / d_s ** 2) ** n)
    elif filter_type == "ideal":
        m = (_grid(T, H, W, d_s, d_t) <= d_s * 2).astype(np.float64)
    elif filter_type == "box":
        m = np.zeros((T, H, W))
        ts_, tt_ = round(int(H // 2) * d_s), round(T // 2 * d_t)
        cf, cr, cc = T // 2, H // 2, W // 2
        m[cf - tt_:cf + tt_, cr - ts_:cr + ts_, cc - ts_:cc + ts_] = 1.0
    else:
        raise NotImplementedError
    return torch.from_numpy(m.astype(np.float32)).expand(shape).clone()


def freq_mix_3d(x, noise, LPF):
    x = x.float().squeeze(0)
    noise = noise.float().squeeze(0)
    xf = fft.fftshift(fft.fftn(x, dim=(-3, -2, -1)), dim=(-3, -2, -1))
    nf = fft.fftshift(fft.fftn(noise, dim=(-3, -2, -1)), dim=(-3, -2, -1))
    mixed = xf * LPF + nf * (1 - LPF)
    mixed = fft.ifftshift(mixed, dim=(-3, -2, -1))
    return fft.ifftn(mixed, dim=(-3, -2, -1)).real


def prepare_latents(z, ddim_alphas, num_inference_steps, video_length, lookahead, noises):
    """funcs.py:53-79"""
    out, k = [], 0
    if lookahead:
        for i in range(video_length // 2):
            alpha = torch.as_tensor(ddim_alphas[0], dtype=torch.float32)   # a 0-dim fp32 tensor in the reference
            beta = 1 - alpha
            out.append(alpha ** 0.5 * z[:, :, [0]] + beta ** 0.5 * noises[k]); k += 1
    for i in range(num_inference_steps):
        alpha = torch.as_tensor(ddim_alphas[i], dtype=torch.float32)
        frame_idx = max(0, i - (num_inference_steps - z.shape[2]))
        out.append(alpha ** 0.5 * z[:, :, [frame_idx]] + (1 - alpha) ** 0.5 * noises[k]); k += 1
    return torch.cat(out, dim=2)


def shift_latents(latents, noise, davis_data=None, encode=None):
    """funcs.py:86-118.  Prompt mode: anchor = the dequeued frame.  DAVIS mode (`davis_data = (frames, masks)`): anchor =
    `encode(last DAVIS frame, RGB)` (the caller's restatement of encode_first_stage_2DAE with its sample noise), and the mask
    queue shifts with its tail kept; returns (latents, masks) then."""
    if davis_data is None:
        anchor = latents[:, :, 0].clone().unsqueeze(2)
    else:
        frames, masks = davis_data
        anchor = frames[:, :, -1].clone().unsqueeze(2)
        if anchor.shape[1] == 4:
            anchor = anchor[:, :3]
        anchor = encode(anchor)
    latents[:, :, :-1] = latents[:, :, 1:].clone()
    lpf = get_freq_filter(anchor.shape, "gaussian", 1, 0.25, 0.25)
    latents[:, :, -1] = freq_mix_3d(anchor, noise.unsqueeze(2), lpf).squeeze(2)
    if davis_data is None:
        return latents
    masks[:, :, :-1] = masks[:, :, 1:].clone()
    masks[:, :, -1] = masks[:, :, -1].clone()
    return latents, masks
