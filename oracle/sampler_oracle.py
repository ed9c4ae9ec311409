"""ORACLE (test infrastructure only): CPU restatement of the sampler arithmetic of
`/root/reference/lvdm/models/samplers/ddim.py` and `lvdm/models/utils_diffusion.py`,
`lvdm/models/ddpm3d.py` schedule buffers.  Pinned by tests/golden/sampler_*.npz (outputs of
the real reference, tools/make_golden.py).  No plotting, no SAM: masks are inputs."""
from __future__ import annotations

import numpy as np
import torch


# ---- utils_diffusion.py:31-53, ddpm3d.py:113-165,362-376 ----------------------------------
def ddpm_buffers(timesteps=1000, linear_start=0.00085, linear_end=0.012, scale_a=1, scale_b=0.7, mid_step=400):
    betas = (torch.linspace(linear_start ** 0.5, linear_end ** 0.5, timesteps, dtype=torch.float64) ** 2).numpy()
    ac = np.cumprod(1. - betas, axis=0)
    scale_arr = np.concatenate((np.linspace(scale_a, scale_b, mid_step), np.full(timesteps, scale_b)))
    return dict(betas=torch.tensor(betas, dtype=torch.float32), alphas_cumprod=torch.tensor(ac, dtype=torch.float32),
                scale_arr=torch.tensor(scale_arr, dtype=torch.float32))


# ---- ddim.py:62-106, utils_diffusion.py:56-93 ------------------------------------------------
def make_schedule(buf, S, eta, T=1000):
    ts = np.linspace(0, T - 1, S).round().copy().astype(np.int64)
    ac = buf["alphas_cumprod"]
    alphas = ac[ts]
    alphas_prev = np.asarray([ac[0]] + ac[ts[:-1]].tolist())
    sigmas = eta * np.sqrt((1 - alphas_prev) / (1 - alphas) * (1 - alphas / alphas_prev))
    sa = buf["scale_arr"]
    return dict(ddim_timesteps=ts, ddim_sigmas=np.asarray(sigmas), ddim_alphas=np.asarray(alphas), ddim_alphas_prev=alphas_prev,
                ddim_sqrt_one_minus_alphas=np.sqrt(1. - np.asarray(alphas)), ddim_scale_arr=np.asarray(sa[ts]),
                ddim_scale_arr_prev=np.asarray([sa[0]] + sa[ts[:-1]].tolist()))


# ---- ddim.py:290-357 --------------------------------------------------------------------------
def p_sample_ddim(sch, x, e_c, e_u, cfg, index, noise, use_scale=True):
    e_t = e_u + cfg * (e_c - e_u)
    size = (x.shape[0], 1, 1, 1, 1)
    a_t = torch.full(size, sch["ddim_alphas"][index])
    a_prev = torch.full(size, sch["ddim_alphas_prev"][index])
    sigma_t = torch.full(size, sch["ddim_sigmas"][index])
    s1m = torch.full(size, sch["ddim_sqrt_one_minus_alphas"][index])
    pred_x0 = (x - s1m * e_t) / a_t.sqrt()
    dir_xt = (1. - a_prev - sigma_t ** 2).sqrt() * e_t
    nz = sigma_t * noise
    if use_scale:
        scale_t = torch.full(size, sch["ddim_scale_arr"][index])
        scale_prev = torch.full(size, sch["ddim_scale_arr_prev"][index])
        pred_x0 = pred_x0 / scale_t
        x_prev = a_prev.sqrt() * scale_prev * pred_x0 + dir_xt + nz
    else:
        x_prev = a_prev.sqrt() * pred_x0 + dir_xt + nz
    return x_prev, pred_x0


# ---- ddim.py:377-649 (arithmetic only) ---------------------------------------------------------
def _calculate_iou(masks1, masks2):
    """ddim.py:905-943"""
    masks1, masks2 = torch.as_tensor(masks1) > 0.5, torch.as_tensor(masks2) > 0.5
    ious = []
    for m1, m2 in zip(masks1, masks2):
        inter = torch.logical_and(m1, m2).sum().float()
        union = torch.logical_or(m1, m2).sum().float()
        ious.append(1.0 if union == 0 else (inter / union).item())
    return torch.tensor(ious).mean().item()


def _apply_segmentation(pred_x0, cond_image, candidate_masks, pre_masks):
    """`_apply_segmentation` (ddim.py:739-903) with the Grounded-SAM-2 output replaced by `candidate_masks` ([n,H,W], or
    None / empty for "no box detected").  Pinned: tools/make_golden.py::sampler_sam_cases runs the REAL `ddim_step` with fake
    sam2_predictor / processor / grounding_model objects returning scripted candidates (tests/golden/sampler_ddim_step_sam.npz);
    tests/test_oracle_sampler.py holds this restatement to it bit-exactly."""
    if candidate_masks is None or len(candidate_masks) == 0:            # :788-793
        if pre_masks is None:
            return pred_x0, None
        masks = pre_masks
    else:
        masks = torch.as_tensor(candidate_masks).float()
        if pre_masks is not None and _calculate_iou(masks, pre_masks) < 0.5:   # :804-807
            masks = pre_masks
    modified = pred_x0.clone()
    for mask in masks:
        if mask.sum() > 0.8 * mask.numel():                              # :820-822
            modified = pred_x0
            continue
        # The reference expands the mask to [1,C,H,W] and lets torch.where broadcast it against the 5-D pred_x0
        # [1,C,1,H,W]: the result is the injected frame REPLICATED C times along the frame axis ([1,C,C,H,W]) -- harmless
        # there because every caller discards pred_x0 (funcs.py:320,336).  Restated here without the replication: one frame.
        m = mask.reshape(1, 1, 1, *mask.shape[-2:]).expand(-1, pred_x0.shape[1], -1, -1, -1)
        ci = cond_image
        if ci is None:
            ci = torch.zeros_like(pred_x0)
        else:
            ci = ci.reshape(ci.shape[0], ci.shape[1], 1, *ci.shape[-2:])
            if ci.shape[1] != pred_x0.shape[1]:
                ci = torch.cat([ci, torch.ones_like(ci[:, :1])], dim=1)
        modified = torch.where(m > 0.5, ci * 2, modified)                 # enhancement_factor = 2 (:847,897-901)
    return modified, masks


def ddim_step(sch, sample, noise_pred, indices, cond_image, ts, noises, momentum, davis_masks=None, gamma=0.5, beta=0.9,
              reference_index_quirk=True, sam_masks=None):
    """noises: list of per-frame [b,c,1,h,w]; momentum: [b,c,f,h,w] state, updated in place. Returns (x_prev, pred_x0).
    sam_masks: optional list over frames of candidate masks [n,H,W] standing in for Grounded-SAM-2 (:592-606)."""
    b, _, f, H, W = sample.shape
    size = (b, 1, 1, 1, 1)
    x_prevs, pred_x0s = [], []
    prev_frame = None
    pre_masks = None                                                     # :391
    for i, index in enumerate(indices):
        x = sample[:, :, [i]]
        e_t = noise_pred[:, :, [i]]
        timestep = ts[i]
        a_t = torch.full(size, sch["ddim_alphas"][index])
        a_prev = torch.full(size, sch["ddim_alphas_prev"][index])
        sigma_t = torch.full(size, sch["ddim_sigmas"][index])
        s1m = torch.full(size, sch["ddim_sqrt_one_minus_alphas"][index])
        pred_x0 = (x - s1m * e_t) / a_t.sqrt()                         # :415
        dir_xt = (1. - a_prev - sigma_t ** 2).sqrt() * e_t             # :418
        mi = i
        if prev_frame is not None:
            g = pred_x0 - prev_frame                                   # :422
            g = g + 1.5 * dir_xt                                       # :423
            momentum[:, :, [i]] = beta * momentum[:, :, [i - 1]] + (1 - beta) * g   # :424-427
            correction_strength = 2 * (1.0 - timestep / 1000.0)        # :428
            pred_x0 = pred_x0 + correction_strength * momentum[:, :, [i]]    # :430,557
            if reference_index_quirk:
                mi = len(range(0, H, 4)) - 1     # plotting loops `for i in range(len(X))` clobber i (:477,502,533)
        prev_frame = pred_x0                                            # :559
        noise = sigma_t * noises[i]                                     # :561
        x_prev = a_prev.sqrt() * pred_x0 + dir_xt + noise               # :562
        if davis_masks is not None and davis_masks.shape[2] > mi:       # :565
            mask = davis_masks[:, :, mi, :, :].unsqueeze(0)
            mask = mask.expand(-1, pred_x0.shape[1], -1, -1, -1)
            enhancement_factor = 1.5 if timestep <= 300 else 1.0        # :582
            if mask.sum() != 0:                                         # :585
                pred_x0 = torch.where(mask > 0.5, cond_image * enhancement_factor, pred_x0)
        elif sam_masks is not None and timestep <= 300:                 # :592-606
            pred_x0, pre_masks = _apply_segmentation(pred_x0, cond_image, sam_masks[i] if i < len(sam_masks) else None, pre_masks)
        pred_x0 = (1 - gamma) * pred_x0 + gamma * noise                 # :609
        x_prevs.append(x_prev)
        pred_x0s.append(pred_x0)
    return torch.cat(x_prevs, dim=2), torch.cat(pred_x0s, dim=2)
