"""MI355X host-side mirror of `lvdm.models.samplers.ddim.DDIMSampler` for the hot path:
`make_schedule` (ddim.py:62-106), `sample`/`ddim_sampling` (:109-252), `p_sample_ddim`
(:274-359), `fifo_onestep` (:255-271), `unet` (:362-374) and the MoCA `ddim_step`
(:377-649).  All per-element arithmetic runs in the fp32 HIP kernels of
csrc/sampler.hip; only O(steps) scalar schedule math stays on the host (numpy, as in
the reference).

Differences from the reference, all explicit:
  * noise is an optional explicit argument everywhere (the reference draws torch.randn
    inside: ddim.py:345,561) so results are reproducible against fixtures;
  * the Grounded-SAM-2 mask producer (ddim.py:713-969) is out of scope: masks come in as
    tensors (`davis_masks`), no debug PNG/matplotlib dumps are written (:432-554,611-641);
  * the two CFG branches are evaluated as ONE batched UNet call when their contexts have
    the same length (bit-equivalent: every UNet op is per-sample).
"""
from __future__ import annotations

import ctypes as C
import math

import numpy as np
import torch

from . import lib as _l
from . import ops
from .unet import same_fps


def make_ddim_timesteps(ddim_discr_method, num_ddim_timesteps, num_ddpm_timesteps, verbose=True):
    """utils_diffusion.py:56-78"""
    if ddim_discr_method == 'uniform':
        ddim_timesteps = np.linspace(0, num_ddpm_timesteps - 1, num_ddim_timesteps).round().copy().astype(np.int64)
        steps_out = ddim_timesteps
    elif ddim_discr_method == 'quad':
        ddim_timesteps = ((np.linspace(0, np.sqrt(num_ddpm_timesteps * .8), num_ddim_timesteps)) ** 2).astype(int)
        steps_out = ddim_timesteps + 1
    else:
        raise NotImplementedError(f'There is no ddim discretization method called "{ddim_discr_method}"')
    if verbose:
        print(f'Selected timesteps for ddim sampler: {steps_out}')
    return steps_out


def make_ddim_sampling_parameters(alphacums, ddim_timesteps, eta, verbose=True):
    """utils_diffusion.py:81-93"""
    alphas = alphacums[ddim_timesteps]
    alphas_prev = np.asarray([alphacums[0]] + alphacums[ddim_timesteps[:-1]].tolist())
    sigmas = eta * np.sqrt((1 - alphas_prev) / (1 - alphas) * (1 - alphas / alphas_prev))
    return sigmas, alphas, alphas_prev


def _st():
    return C.c_void_p(ops.current_stream())


def _f32c(t):
    return t.contiguous() if t.dtype == torch.float32 else t.float().contiguous()


class DDIMSampler(object):
    def __init__(self, model, schedule="linear", use_self_attention=False, **kwargs):
        self.model = model
        self.ddpm_num_timesteps = model.num_timesteps
        self.schedule = schedule
        self.counter = 0
        self.use_self_attention = use_self_attention
        self.reference_index_quirk = True    # see ddim_step
        self.cfg_mode = "batched"            # "batched": cond+uncond as one B=2 launch; "concurrent": two B=1
                                             # hipGraphs on two streams (UNetModel.forward_concurrent)
        self.beta = 0.9                      # momentum decay (ddim.py:397)
        self._base_engine = None
        self.use_graph = True                # `sample`: one hipGraph per DDIM step (fifo_graph.BaseEngine) where the call allows it
        self.share_prefix = True             # the two CFG branches share everything before the first cross-attention (same x, same
                                             # t): computed once (UNetModel.forward_segments(shared_x=True)); False: plain B = 2 batch

    # ---- schedule (host) ---------------------------------------------------------------
    def make_schedule(self, ddim_num_steps, ddim_discretize="uniform", ddim_eta=0., verbose=True):
        self.ddim_timesteps = make_ddim_timesteps(ddim_discretize, ddim_num_steps, self.ddpm_num_timesteps, verbose=verbose)
        alphas_cumprod = self.model.alphas_cumprod
        assert alphas_cumprod.shape[0] == self.ddpm_num_timesteps, 'alphas have to be defined for each timestep'
        ac = alphas_cumprod.detach().cpu()
        self.use_scale = self.model.use_scale
        if self.use_scale:
            scale_arr = self.model.scale_arr.detach().cpu()
            self.ddim_scale_arr = scale_arr[self.ddim_timesteps]                                           # :82-83
            self.ddim_scale_arr_prev = np.asarray([scale_arr[0]] + scale_arr[self.ddim_timesteps[:-1]].tolist())  # :84-85
        sig, al, alp = make_ddim_sampling_parameters(alphacums=ac, ddim_timesteps=self.ddim_timesteps, eta=ddim_eta, verbose=verbose)
        self.ddim_sigmas, self.ddim_alphas, self.ddim_alphas_prev = sig, al, alp
        self.ddim_sqrt_one_minus_alphas = np.sqrt(1. - al)

    # ---- UNet with classifier-free guidance ----------------------------------------------
    def _cfg_eps(self, x, t, c, uc, scale, **kwargs):
        """e_u + s (e_c - e_u)  (ddim.py:290-304,362-374)"""
        if uc is None or scale == 1.:
            return self.model.apply_model(x, t, c, **kwargs)
        batched = False
        if isinstance(c, dict) and isinstance(uc, dict) and x.shape[0] * 2 <= 64:
            cc = torch.cat(c["c_crossattn"], 1)
            cu = torch.cat(uc["c_crossattn"], 1)
            batched = cc.shape == cu.shape and set(c.keys()) == set(uc.keys()) <= {"c_crossattn", "fps"}
        unet = getattr(getattr(self.model, "model", None), "diffusion_model", None)
        if batched and self.cfg_mode == "concurrent" and hasattr(unet, "forward_concurrent") and not kwargs.get("no_concurrent"):
            f_c, f_u = c.get("fps", 16), uc.get("fps", 16)
            e_c, e_u = unet.forward_concurrent([dict(x=x, timesteps=t, context=cc, fps=f_c),
                                                dict(x=x, timesteps=t, context=cu, fps=f_u)])
        elif batched and self.share_prefix and hasattr(unet, "forward_segments") and \
                getattr(self.model.model, "conditioning_key", None) == "crossattn" and same_fps([c.get("fps", 16), uc.get("fps", 16)]):
            # both branches in ONE forward that shares everything before the first cross-attention (same x, same t)
            B = x.shape[0]
            e = unet.forward_segments(x, t, [cc, cu], fps=[c.get("fps", 16), uc.get("fps", 16)], shared_x=True)
            e_c, e_u = e[:B], e[B:]
        elif batched:
            B = x.shape[0]
            cond = {"c_crossattn": [torch.cat([cc, cu], 0)]}
            if "fps" in c:
                f_c, f_u = c["fps"], uc["fps"]
                if isinstance(f_c, int):
                    cond["fps"] = f_c
                else:
                    cond["fps"] = torch.cat([torch.as_tensor(f_c).reshape(-1).expand(B), torch.as_tensor(f_u).reshape(-1).expand(B)], 0)
            tt = torch.as_tensor(t, device=x.device).reshape(-1)
            if tt.shape[0] == B:                     # uniform timestep per sample
                t2 = torch.cat([tt, tt], 0)
            else:                                    # FIFO: per-frame timesteps, B == 1 -> per-(b,t) rows
                t2 = torch.cat([tt, tt], 0)
            e = self.model.apply_model(torch.cat([x, x], 0), t2, cond, **kwargs)
            e_c, e_u = e[:B], e[B:]
        else:
            e_c = self.model.apply_model(x, t, c, **kwargs)
            e_u = self.model.apply_model(x, t, uc, **kwargs)
        e_c, e_u = _f32c(e_c), _f32c(e_u)
        out = torch.empty_like(e_c)
        _l.check(_l.load().moca_cfg_combine_f32(_l.ptr(e_c), _l.ptr(e_u), _l.ptr(out), float(scale), e_c.numel(), _st()),
                 "moca_cfg_combine_f32")
        return out

    @torch.no_grad()
    def unet(self, x, c, t, unconditional_guidance_scale=1., unconditional_conditioning=None, **kwargs):
        """ddim.py:362-374"""
        return self._cfg_eps(x, t, c, unconditional_conditioning, unconditional_guidance_scale, **kwargs)

    # ---- base DDIM step --------------------------------------------------------------------
    @torch.no_grad()
    def p_sample_ddim(self, x, c, t, index, repeat_noise=False, use_original_steps=False, quantize_denoised=False,
                      temperature=1., noise_dropout=0., score_corrector=None, corrector_kwargs=None,
                      unconditional_guidance_scale=1., unconditional_conditioning=None, uc_type=None,
                      conditional_guidance_scale_temporal=None, noise=None, **kwargs):
        """ddim.py:274-359 (uc_type None, no score corrector / temporal guidance / quantisation)"""
        if use_original_steps or quantize_denoised or score_corrector is not None or uc_type is not None \
                or conditional_guidance_scale_temporal is not None or noise_dropout > 0.:
            raise NotImplementedError("option outside the MoCA driver's call (funcs.py:223-236)")
        e_t = self._cfg_eps(x, t, c, unconditional_conditioning, unconditional_guidance_scale, **kwargs)
        return self.ddim_update(x, e_t, index, noise=noise, temperature=temperature)

    def ddim_update(self, x, e_t, index, noise=None, temperature=1.):
        """the arithmetic tail of p_sample_ddim (ddim.py:328-357) on the HIP kernel"""
        x, e_t = _f32c(x), _f32c(e_t)
        if noise is None:
            noise = torch.randn(x.shape, device=x.device)          # noise_like, common.py:31-34
        noise = _f32c(noise)
        if temperature != 1.:
            noise = noise * temperature
        x_prev, pred_x0 = torch.empty_like(x), torch.empty_like(x)
        f32 = np.float32
        use_scale = bool(self.use_scale)
        _l.check(_l.load().moca_ddim_update_f32(
            _l.ptr(x), _l.ptr(e_t), _l.ptr(noise), _l.ptr(x_prev), _l.ptr(pred_x0),
            f32(self.ddim_alphas[index]), f32(self.ddim_alphas_prev[index]), f32(self.ddim_sigmas[index]),
            f32(self.ddim_sqrt_one_minus_alphas[index]), 1 if use_scale else 0,
            f32(self.ddim_scale_arr[index]) if use_scale else f32(1), f32(self.ddim_scale_arr_prev[index]) if use_scale else f32(1),
            x.numel(), _st()), "moca_ddim_update_f32")
        return x_prev, pred_x0

    def release(self):
        """free the cached step graph of `sample` (plan buffers + hipGraph)"""
        if self._base_engine is not None:
            self._base_engine[1].close()
            self._base_engine = None

    @torch.no_grad()
    def sample(self, S, batch_size, shape, conditioning=None, eta=0., x_T=None, verbose=False,
               unconditional_guidance_scale=1., unconditional_conditioning=None, latents_dir=None, noises=None, **kwargs):
        """ddim.py:109-181 -> ddim_sampling (:183-252): the base 'N DDIM steps, single prompt' loop"""
        self.make_schedule(ddim_num_steps=S, ddim_eta=eta, verbose=verbose)
        size = (batch_size,) + tuple(shape)
        device = self.model.betas.device
        img = torch.randn(size, device=device) if x_T is None else x_T
        if latents_dir is not None:
            torch.save(img, f"{latents_dir}/0.pt")                    # :233-234
        time_range = np.flip(self.ddim_timesteps)
        total_steps = self.ddim_timesteps.shape[0]
        kwargs.pop("temporal_length", None); kwargs.pop("conditional_guidance_scale_temporal", None)
        fps_kwargs = {}
        from .fifo_graph import BaseEngine
        if self.use_graph and self.share_prefix and BaseEngine.supported(self.model, img, conditioning, unconditional_conditioning,
                                                                         unconditional_guidance_scale):
            # the loop body as one hipGraph per step (fifo_graph.BaseEngine): latents, schedule and noise stay on the device
            seed = int(torch.randint(0, 2 ** 62, (1,)).item())                  # the host generator seeds the device stream
            cu, cn = conditioning["c_crossattn"], unconditional_conditioning["c_crossattn"]
            key = (tuple(img.shape), sum(c.shape[1] for c in cu), sum(c.shape[1] for c in cn), float(unconditional_guidance_scale),
                   self.ddim_timesteps.tobytes(), np.asarray(self.ddim_sigmas).tobytes(), str(img.device))
            if self._base_engine is not None and self._base_engine[0] != key:
                self._base_engine[1].close()
                self._base_engine = None
            if self._base_engine is None:
                self._base_engine = (key, BaseEngine(self.model, self, img, conditioning, unconditional_conditioning,
                                                     unconditional_guidance_scale, seed=seed))
            else:
                self._base_engine[1].reset(img, conditioning, unconditional_conditioning, seed)
            eng = self._base_engine[1]
            for i in range(total_steps):
                eng.step(noise=None if noises is None else noises[i])
            img = eng.latents().to(img.dtype)
            time_range = []
        for i, step in enumerate(time_range):
            index = total_steps - i - 1
            ts = torch.full((batch_size,), int(step), device=device, dtype=torch.long)
            img, pred_x0 = self.p_sample_ddim(img, conditioning, ts, index=index,
                                              unconditional_guidance_scale=unconditional_guidance_scale,
                                              unconditional_conditioning=unconditional_conditioning,
                                              noise=None if noises is None else noises[i], **fps_kwargs)
        if latents_dir is not None:
            torch.save(img, f"{latents_dir}/{total_steps}.pt")        # :249-250
        return img, {}

    @torch.no_grad()
    def unet_windows(self, windows, c, ts_list, unconditional_guidance_scale=1., unconditional_conditioning=None, **kwargs):
        """CFG noise prediction for several independent FIFO windows in ONE batched UNet call.

        The 2n windows of one outer FIFO iteration read only pre-iteration frames (funcs.py:305-352: rank r
        writes [8r+8,8r+16) and rank r-1 reads [8r-8,8r+8)), so their UNet evaluations are independent; the
        reference's own multi-GPU variant relies on the same fact (funcs_mp.py:207-221).  Windows become the
        batch dimension with per-(window, frame) timesteps; with classifier-free guidance of equal context
        length the unconditional branch doubles the batch (2 x 2n = 16 videos per launch).  Returns the list of
        per-window eps tensors, equal to `self.unet(w, c, ts, ...)` per window."""
        W = len(windows)
        x = torch.cat(windows, 0)
        t = torch.cat([torch.as_tensor(np.asarray(ts).copy(), device=x.device).to(torch.long).reshape(-1) for ts in ts_list], 0)
        uc, scale = unconditional_conditioning, unconditional_guidance_scale
        cc = torch.cat(c["c_crossattn"], 1).expand(W, -1, -1)
        fps_c = c.get("fps", 16)

        def fps_rows(f, n):
            return f if isinstance(f, int) else torch.as_tensor(f, device=x.device).reshape(-1)[:1].expand(n)

        if uc is None or scale == 1.:
            e = self.model.apply_model(x, t, {"c_crossattn": [cc], "fps": fps_rows(fps_c, W)}, **kwargs)
            return list(e.split(1, 0))
        cu = torch.cat(uc["c_crossattn"], 1).expand(W, -1, -1)
        fps_u = uc.get("fps", fps_c)
        unet = getattr(getattr(self.model, "model", None), "diffusion_model", None)
        if self.share_prefix and hasattr(unet, "forward_segments") and getattr(self.model.model, "conditioning_key", None) == "crossattn" \
                and same_fps([fps_c, fps_u]):
            fr_c = fps_rows(fps_c, W)
            e = unet.forward_segments(x, t, [cc, cu], fps=[fr_c, fr_c if fps_u is fps_c else fps_rows(fps_u, W)], shared_x=True)
            e_c, e_u = e[:W], e[W:]
        elif cc.shape == cu.shape and isinstance(fps_c, int) == isinstance(fps_u, int):
            fps2 = fps_c if isinstance(fps_c, int) else torch.cat([fps_rows(fps_c, W), fps_rows(fps_u, W)], 0)
            e = self.model.apply_model(torch.cat([x, x], 0), torch.cat([t, t], 0),
                                       {"c_crossattn": [torch.cat([cc, cu], 0)], "fps": fps2}, **kwargs)
            e_c, e_u = e[:W], e[W:]
        else:   # e.g. 154 conditional tokens (two prompts) vs 77 unconditional: two batched calls
            e_c = self.model.apply_model(x, t, {"c_crossattn": [cc], "fps": fps_rows(fps_c, W)}, **kwargs)
            e_u = self.model.apply_model(x, t, {"c_crossattn": [cu], "fps": fps_rows(fps_u, W)}, **kwargs)
        e_c, e_u = _f32c(e_c), _f32c(e_u)
        out = torch.empty_like(e_c)
        _l.check(_l.load().moca_cfg_combine_f32(_l.ptr(e_c), _l.ptr(e_u), _l.ptr(out), float(scale), e_c.numel(), _st()),
                 "moca_cfg_combine_f32")
        return list(out.split(1, 0))

    # ---- MoCA FIFO step -----------------------------------------------------------------------
    @torch.no_grad()
    def fifo_onestep(self, cond, shape, latents=None, timesteps=None, indices=None, unconditional_guidance_scale=1.,
                     unconditional_conditioning=None, cond_image=None, target=None, use_self_attention=False,
                     davis_masks=None, noise=None, sam_masks=None, sam_masks_fn=None, **kwargs):
        """ddim.py:255-271"""
        device = self.model.betas.device
        ts = torch.as_tensor(np.asarray(timesteps).copy(), device=device).to(dtype=torch.long)
        noise_pred = self.unet(latents, cond, ts, unconditional_guidance_scale=unconditional_guidance_scale,
                               unconditional_conditioning=unconditional_conditioning, **kwargs)
        return self.ddim_step(latents, noise_pred, indices, cond_image, target, ts, use_self_attention=use_self_attention,
                              davis_masks=davis_masks, noise=noise, sam_masks=sam_masks, sam_masks_fn=sam_masks_fn)

    @staticmethod
    def calculate_iou(masks1, masks2):
        """ddim.py:905-943: mean IoU over zip(masks1, masks2) of the masks binarised at 0.5 (both empty -> 1.0)"""
        m1, m2 = torch.as_tensor(masks1) > 0.5, torch.as_tensor(masks2) > 0.5
        ious = []
        for a, b in zip(m1, m2):
            b = b.to(a.device)
            inter, union = torch.logical_and(a, b).sum().float(), torch.logical_or(a, b).sum().float()
            ious.append(torch.tensor(1.0) if union == 0 else (inter / union).cpu())
        return torch.stack(ious).mean().item() if ious else 0.0

    def select_sam_masks(self, sam_masks, ts, H, W, device):
        """The mask bookkeeping of `_apply_segmentation` (ddim.py:739-903) for PRECOMPUTED candidate masks (the
        Grounded-SAM-2 producer itself is out of scope).  `sam_masks[i]` = the masks SAM would return for frame i, shape
        [n_i, H, W] (None / empty = no box detected).  Per frame, in order, with `pre_masks` starting at None (:391):
          * only frames with timestep <= 300 are touched (:592);
          * no detection -> previous masks, or nothing if there are none yet (:788-793);
          * IoU(new, previous) < 0.5 -> previous masks (:804-807);
          * masks are applied in order; one covering > 80 % of the frame RESETS what the earlier ones injected (:820-822).
        Returns (effective [f, H*W] float mask, frame mask index [f] with -1 = no injection)."""
        f = len(ts)
        eff = torch.zeros(f, H * W, dtype=torch.float32, device=device)
        midx = np.full((f,), -1, dtype=np.int32)
        pre = None
        for i in range(f):
            if float(ts[i]) > 300:
                continue
            cand = sam_masks[i] if i < len(sam_masks) else None
            if cand is not None:
                cand = torch.as_tensor(cand, device=device).float().reshape(-1, H, W)
            if cand is None or cand.shape[0] == 0:
                if pre is None:
                    continue                                   # returns (pred_x0, None): pre_masks stays None
                masks = pre
            else:
                masks = cand
                if pre is not None and self.calculate_iou(masks, pre) < 0.5:
                    masks = pre
            pre = masks
            cur = torch.zeros(H, W, dtype=torch.bool, device=device)
            for m in masks:
                if m.sum() > 0.8 * m.numel():
                    cur.zero_()                                # modified_pred_x0 = pred_x0 (:821)
                    continue
                cur |= m > 0.5
            eff[i] = cur.reshape(-1).float()
            midx[i] = i
        return eff, midx

    def step_tables(self, indices, ts_np, H, Fm):
        """Host-side tables of one `ddim_step` call (they depend on the window slot only, not on the data): coef [f][6] =
        {sqrt(a_t), sqrt(a_prev), sigma_t, sqrt(1-a_t), sqrt(1-a_prev-sigma^2), 2(1-t/1000)} evaluated in fp32 exactly as the
        reference's 0-dim tensors are (ddim.py:409-428), enh [f] (:582) and the mask frame consulted for each frame (:565-567
        incl. the clobbered loop variable, -1 = none; `Fm` = mask frames available)."""
        f32 = np.float32
        f = len(indices)
        coef = np.zeros((f, 6), dtype=np.float32)
        enh = np.ones((f,), dtype=np.float32)
        midx = np.full((f,), -1, dtype=np.int32)
        for i, index in enumerate(indices):
            a_t, a_prev = f32(self.ddim_alphas[index]), f32(self.ddim_alphas_prev[index])
            sigma, s1m = f32(self.ddim_sigmas[index]), f32(self.ddim_sqrt_one_minus_alphas[index])
            coef[i, 0] = np.sqrt(a_t)                                  # a_t.sqrt()            :415
            coef[i, 1] = np.sqrt(a_prev)                               # a_prev.sqrt()         :562
            coef[i, 2] = sigma
            coef[i, 3] = s1m
            coef[i, 4] = np.sqrt(f32(1.) - a_prev - sigma * sigma)     # :418
            coef[i, 5] = f32(2) * (f32(1.0) - f32(ts_np[i]) / f32(1000.0))   # correction_strength :428
            enh[i] = 1.5 if ts_np[i] <= 300 else 1.0                   # :582
            mi = i
            if self.reference_index_quirk and i >= 1:
                mi = len(range(0, H, 4)) - 1                            # clobbered `i` (:477,502,533)
            if Fm > mi:                                                 # :565
                midx[i] = mi
        return coef, enh, midx

    @torch.no_grad()
    def ddim_step(self, sample, noise_pred, indices, cond_image, target, ts, gamma=0.5, use_self_attention=False,
                  davis_masks=None, noise=None, sam_masks=None, sam_masks_fn=None):
        """ddim.py:377-649.  Returns (x_prev, pred_x0); `self.momentum` persists across calls (:395-397).

        Mask semantics follow the reference bit for bit, quirk included: its plotting loops
        reuse the loop variable `i` (ddim.py:477,502,533), so for every frame i >= 1 the mask
        frame consulted at :565-567 is ceil(H/4)-1, not i (frame 0 uses mask frame 0).  Set
        `self.reference_index_quirk = False` for the per-frame mask the code evidently meant.
        Without `davis_masks` the reference calls Grounded-SAM-2 per frame (out of scope); pass what it would
        return as `sam_masks` (list over frames of [n,H,W] candidate masks) to get that branch's behaviour
        (`select_sam_masks`: t <= 300 only, IoU fallback, > 80 % reset, factor 2); with neither, no injection.
        `sam_masks_fn(pred_x0_frame [1,C,1,H,W], target, frame) -> [n,H,W] or None` is the mask PRODUCER interface (what
        `_apply_segmentation` does with Grounded-SAM-2, ddim.py:745-801): it is shown the momentum-corrected pred_x0 of every frame
        with t <= 300 (a first pass of the same kernel without injection and gamma = 0 yields it) and its answers go through the
        same bookkeeping as `sam_masks`."""
        if sam_masks_fn is not None and sam_masks is None and davis_masks is None:
            _, p0 = self.ddim_step(sample, noise_pred, indices, cond_image, target, ts, gamma=0.0, use_self_attention=True, noise=noise
                                   if noise is not None else torch.zeros_like(sample))
            ts_h = np.asarray(ts.detach().cpu()) if torch.is_tensor(ts) else np.asarray(ts)
            sam_masks = [sam_masks_fn(p0[:, :, [i]], target, i) if ts_h[i] <= 300 else None for i in range(sample.shape[2])]
        b, Cc, f, H, W = sample.shape
        device = sample.device
        sample, noise_pred = _f32c(sample), _f32c(noise_pred)
        if noise is None:
            noise = torch.randn(sample.shape, device=device)          # per-frame noise_like draws, :561
        noise = _f32c(noise)
        if not hasattr(self, 'momentum') or self.momentum.shape != sample.shape:
            self.momentum = torch.zeros_like(sample)                   # :395-397
        f32 = np.float32
        ts_np = np.asarray(ts.detach().cpu()) if torch.is_tensor(ts) else np.asarray(ts)
        Fm = int(davis_masks.shape[2]) if davis_masks is not None else 0
        coef, enh, midx = self.step_tables(indices, ts_np, H, Fm)
        x_prev, pred_x0 = torch.empty_like(sample), torch.empty_like(sample)
        coef_d = torch.from_numpy(coef).to(device)
        mask_d = cond_d = midx_d = enh_d = ws = None
        if davis_masks is None and sam_masks is not None and not use_self_attention:
            if b != 1:
                raise ValueError("the segmentation branch squeezes the batch axis (ddim.py:746-747): batch must be 1")
            eff, midx = self.select_sam_masks(sam_masks, ts_np, H, W, device)
            davis_masks = eff.reshape(1, 1, f, H, W)            # same injection kernel, other factor / frame selection
            Fm = f
            enh[:] = 2.0                                        # enhancement_factor = 2 (:847)
        if davis_masks is not None:
            mask_d = _f32c(davis_masks.to(device)).reshape(b, 1, Fm, H * W)
            if cond_image is None:
                cond_d = torch.zeros(b, Cc, H * W, device=device)     # :573-574
            else:
                ci = cond_image.to(device)
                if ci.shape[1] != Cc:
                    if ci.shape[1] == 3:                               # :575-578
                        ci = torch.cat([ci, torch.ones_like(ci[:, :1])], dim=1)
                    else:
                        raise ValueError(f"Conditional image must have 3 or 4 channels, got {ci.shape[1]}")
                cond_d = _f32c(ci).reshape(-1, Cc, H * W).expand(b, Cc, H * W).contiguous()
            midx_d = torch.from_numpy(midx).to(device)
            enh_d = torch.from_numpy(enh).to(device)
            ws = torch.empty(max(Fm, 1), dtype=torch.float32, device=device)
        _l.check(_l.load().moca_fifo_ddim_step_f32(
            _l.ptr(sample), _l.ptr(noise_pred), _l.ptr(noise), _l.ptr(self.momentum), _l.ptr(x_prev), _l.ptr(pred_x0),
            _l.ptr(coef_d), _l.ptr(mask_d), _l.ptr(cond_d), _l.ptr(midx_d), _l.ptr(enh_d), _l.ptr(ws),
            b, Cc, f, Fm, H * W, f32(self.beta), f32(1 - self.beta), f32(gamma), f32(1 - gamma), _st()),
            "moca_fifo_ddim_step_f32")
        return x_prev, pred_x0
