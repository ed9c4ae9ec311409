"""ORACLE (test infrastructure only): the two sampling LOOPS of the reference composed from the oracle's pieces --
`base_ddim_sampling` -> `DDIMSampler.sample` -> `ddim_sampling` (/root/reference/scripts/evaluation/funcs.py:177-241,
lvdm/models/samplers/ddim.py:109-252) and `fifo_ddim_sampling` (funcs.py:243-373: window order, write-back slice, emitted
frame index, queue / mask shift, both the prompt-mode segmentation branch and the DAVIS-mask branch of `ddim_step`).
Pinned by tests/golden/loop_base.npz / loop_fifo.npz: tools/make_golden.py::loop_cases ran the REAL loops (real reduced-width
UNet inside the real DiffusionWrapper / apply_model, real reduced-width AutoencoderKL) with every random draw replaced by a
named tensor; `draw(kind, shape)` below hands the same tensors back in the reference's call order.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package."""
from __future__ import annotations

import numpy as np
import torch

from . import freeinit_oracle as FO
from . import sampler_oracle as SO


def base_ddim_sampling(unet, decode, buf, ctx, uctx, shape, S, eta, cfg, draw):
    """funcs.py:177-241.  unet(x, t, context) -> eps; decode(z [b,c,t,h,w]) -> images.  Returns (images, schedule, samples, x_T)."""
    sch = SO.make_schedule(buf, S, eta)
    x_T = draw("randn", tuple(shape))                                   # ddim.py:201
    img = x_T
    for i, step in enumerate(np.flip(sch["ddim_timesteps"])):          # :226-237
        index = S - i - 1
        t = torch.full((shape[0],), int(step), dtype=torch.long)
        e_c, e_u = unet(img, t, ctx), unet(img, t, uctx)               # :298-299
        img, _ = SO.p_sample_ddim(sch, img, e_c, e_u, cfg, index, draw("noise_like", tuple(shape)))
    return decode(img), sch, img, x_T


def fifo_windows(video_length, num_partitions, lookahead):
    """funcs.py:306-309: (start, mid, end) for rank = last .. 0"""
    f = video_length
    n = 2 * num_partitions if lookahead else num_partitions
    for rank in reversed(range(n)):
        start = rank * (f // 2) if lookahead else rank * f
        yield start, start + f // 2, start + f


def fifo_ddim_sampling(unet, decode, sch, args, ctx, uctx, cfg, cimg, draw, z=None, davis=None, encode=None, sam=None,
                       n_iterations=None):
    """funcs.py:243-373.  `z` = the cached base latents (`{N}.pt`), or `davis = (frames, masks)` with `encode(frames_rgb, noises)`
    = encode_first_stage_2DAE.  `sam(call, F, H, W)` -> scripted Grounded-SAM-2 candidates of the call-th ddim_step (prompt mode).
    Returns dict(queue=[after each shift], latents_emitted=[...], frames=[decoded], x_prev=[per ddim_step], pred_x0=[...],
    momentum, masks=[after each shift])."""
    f, N = args.video_length, args.num_inference_steps
    masks = None
    if davis is not None:                                               # prepare_latents, funcs.py:38-48
        frames, masks = davis
        rgb = frames[:, :3] if frames.shape[1] == 4 else frames
        z = encode(rgb, [draw("randn", None) for _ in range(rgb.shape[2])])
        masks = masks.clone()
    nz = []
    n_q = (f // 2 if args.lookahead_denoising else 0) + N
    for _ in range(n_q):
        nz.append(draw("randn_like", (z.shape[0], z.shape[1], 1) + tuple(z.shape[3:])))
    lat = FO.prepare_latents(z, sch["ddim_alphas"], N, f, args.lookahead_denoising, nz)
    timesteps, indices = sch["ddim_timesteps"], np.arange(N)
    if args.lookahead_denoising:                                        # :292-294
        timesteps = np.concatenate([np.full((f // 2,), timesteps[0]), timesteps])
        indices = np.concatenate([np.full((f // 2,), 0), indices])
    total = args.new_video_length + N - f if n_iterations is None else n_iterations
    out = dict(queue=[], latents_emitted=[], frames=[], x_prev=[], pred_x0=[], masks=[])
    b, C, _, H, W = lat.shape
    momentum = torch.zeros(b, C, f, H, W)                               # ddim.py:395-397 (persists across calls)
    call = 0
    for _ in range(total):
        for start, mid, end in fifo_windows(f, args.num_partitions, args.lookahead_denoising):
            t, idx = timesteps[start:end], indices[start:end]
            x = lat[:, :, start:end].clone()
            ts = torch.as_tensor(np.asarray(t).copy()).long()
            e_c, e_u = unet(x, ts, ctx), unet(x, ts, uctx)              # ddim.py:366-369
            eps = e_u + cfg * (e_c - e_u)                               # :372
            noises = [draw("noise_like", (b, C, 1, H, W)) for _ in range(f)]
            wm = masks[:, :, start:end].clone() if masks is not None else None
            cands = sam(call, f, H, W) if (sam is not None and masks is None) else None
            x_prev, p0 = SO.ddim_step(sch, x, eps, idx, cimg, ts, noises, momentum,
                                      davis_masks=wm, sam_masks=cands)
            out["x_prev"].append(x_prev)
            out["pred_x0"].append(p0)
            call += 1
            if args.lookahead_denoising:                                # :351-354
                lat[:, :, mid:end] = x_prev[:, :, -(f // 2):]
            else:
                lat[:, :, start:end] = x_prev
        first = f // 2 if args.lookahead_denoising else 0              # :358-360
        emitted = lat[:, :, [first]].clone()
        out["latents_emitted"].append(emitted)
        if decode is not None:
            out["frames"].append(decode(emitted))
        if davis is not None:                                           # :367-371
            anchor_nz = draw("randn", None)
            enc = lambda x, _n=anchor_nz: encode(x, [_n])
            lat, masks = FO.shift_latents(lat, draw("randn_like", (b, C, H, W)), davis_data=(frames, masks), encode=enc)
            out["masks"].append(masks.clone())
        else:
            lat = FO.shift_latents(lat, draw("randn_like", (b, C, H, W)))
        out["queue"].append(lat.clone())
    out["momentum"] = momentum
    return out
