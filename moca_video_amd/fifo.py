"""Latent-queue side of the FIFO / MoCA loop (`scripts/evaluation/funcs.py`): `prepare_latents`
(:21-82), `shift_latents` (:86-99), `base_ddim_sampling` (:177-241) and the window scheduling of
`fifo_ddim_sampling` (:243-373).  The queue stays resident in HBM; VAE decoding of emitted frames
(funcs.py:359-365) is outside the hot path: the loop hands each emitted latent to `emit`.
Noise is an optional explicit argument wherever the reference calls torch.randn*."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import lib as _l
from . import ops
from .fifo_graph import FifoEngine, fifo_windows
from .freeinit import freq_mix_3d, get_freq_filter
from .sampler import DDIMSampler


def prepare_latents(args, input_path, sampler, model=None, data=None, initial_latents=None, noises=None):
    """funcs.py:21-82: frame j of the queue = sqrt(a_j) z[frame_idx] + sqrt(1-a_j) eps, frame_idx = max(0, j - (N - z.shape[2]));
    with lookahead the first f/2 frames all use a_0.  z = the cached base-sampling latents (`{N}.pt`), `initial_latents`, or --
    DAVIS mode, `data=(frames [b,3|4,t,H,W], masks)` -- the VAE encoding of the frames (`model.encode_first_stage_2DAE`)."""
    if data is not None:                                      # DAVIS branch, funcs.py:38-48
        frames, _masks = data
        frames = frames.to("cuda")
        if frames.shape[1] == 4:                              # RGBA -> RGB (:44-45)
            frames = frames[:, :3]
        initial_latents = model.encode_first_stage_2DAE(frames)
    elif initial_latents is None:
        initial_latents = torch.load(input_path + f"/{args.num_inference_steps}.pt")
    z = initial_latents.to("cuda", torch.float32).contiguous()
    b, c, tz, h, w = z.shape
    n_look = args.video_length // 2 if args.lookahead_denoising else 0
    N = args.num_inference_steps
    Q = n_look + N
    # per queue frame: which base frame it starts from and its noise level (:55-77) -- O(Q) host scalars, evaluated like the
    # reference's 0-dim fp32 tensors (alpha ** 0.5, (1 - alpha) ** 0.5); the arithmetic over the latents is one kernel
    alphas = torch.as_tensor(np.asarray(sampler.ddim_alphas), dtype=torch.float32)
    j_alpha = torch.cat([alphas[:1].expand(n_look), alphas[:N]])
    coef_z, coef_n = j_alpha ** 0.5, (1 - j_alpha) ** 0.5
    fidx = torch.tensor([0] * n_look + [max(0, i - (N - tz)) for i in range(N)], dtype=torch.int32)
    if noises is None:
        noise = torch.randn(b, c, Q, h, w, device=z.device)          # the Q torch.randn_like draws of :62,72, one tensor
    else:
        noise = torch.cat([n.to(z.device, torch.float32).reshape(b, c, 1, h, w) for n in noises[:Q]], dim=2).contiguous()
    out = torch.empty(b, c, Q, h, w, dtype=torch.float32, device=z.device)
    cz, cn, fi = coef_z.to(z.device), coef_n.to(z.device), fidx.to(z.device)      # (named: a temporary's block would be reused by the next)
    _l.check(_l.load().moca_fifo_prepare_queue_f32(_l.ptr(z), _l.ptr(noise), _l.ptr(out), _l.ptr(cz), _l.ptr(cn), _l.ptr(fi), b * c, tz, Q, h * w,
                                                   C.c_void_p(ops.current_stream())), "moca_fifo_prepare_queue_f32")
    return out


def shift_latents(latents, davis_data=None, model=None, noise=None, anchor_noise=None):
    """funcs.py:86-118: dequeue frame 0, shift left, enqueue FreeInit-mixed noise.

    Prompt mode (`davis_data` None, :88-99): the FreeInit anchor is the dequeued frame.  Returns `latents`.
    DAVIS mode (`davis_data = (frames [b,3|4,t,H,W], masks [b,1,t,h,w])`, :101-118): the anchor is the VAE encoding
    (`model.encode_first_stage_2DAE`, a fresh posterior sample per call; `anchor_noise` [b,4,1,h,w] fixes the draw) of the
    LAST DAVIS frame, and the mask queue is shifted with its tail refilled from its own last frame.  Returns
    `(latents, (frames, masks))` like the reference."""
    if davis_data is None:
        anchor_frame = latents[:, :, 0].clone().unsqueeze(2)
    else:
        frames, masks = davis_data
        anchor_frame = frames[:, :, -1].clone().unsqueeze(2)
        if anchor_frame.shape[1] == 4:                        # RGBA -> RGB (:104-105)
            anchor_frame = anchor_frame[:, :3]
        anchor_frame = model.encode_first_stage_2DAE(anchor_frame.to(latents.device), noise=anchor_noise)
    latents[:, :, :-1] = latents[:, :, 1:].clone()
    new_noise = (torch.randn_like(latents[:, :, -1]) if noise is None else noise.to(latents.device, latents.dtype).reshape(latents[:, :, -1].shape)).unsqueeze(2)
    freq_filter = get_freq_filter(anchor_frame.shape, latents.device, "gaussian", 1, 0.25, 0.25)
    latents[:, :, -1] = freq_mix_3d(anchor_frame, new_noise, freq_filter).squeeze(2)
    if davis_data is None:
        return latents
    masks[:, :, :-1] = masks[:, :, 1:].clone()
    masks[:, :, -1] = masks[:, :, -1].clone()                 # :116 (the tail keeps the last DAVIS mask)
    return latents, (frames, masks)


def uncond_embedding(model, c_emb, uc_emb):
    """The unconditional context of funcs.py:199-208 / :268-270: `model.uncond_type == "empty_seq"` (the YAML's value) is
    the text encoding of the EMPTY PROMPT -- it needs the text encoder, so the caller must pass it (`uc_emb`, e.g.
    `embed_text("")`); only `"zero_embed"` is zeros.  Passing nothing under "empty_seq" raises instead of silently
    sampling with a different unconditional branch than the reference."""
    if uc_emb is not None:
        return uc_emb
    utype = getattr(model, "uncond_type", "empty_seq")
    if utype == "zero_embed":
        return torch.zeros_like(c_emb)
    if utype == "empty_seq":
        if getattr(model, "cond_stage_model", None) is not None and hasattr(model, "empty_prompt_tokens"):
            return model.get_learned_conditioning(model.empty_prompt_tokens.expand(c_emb.shape[0], -1))
        raise ValueError('uncond_type == "empty_seq": classifier-free guidance needs the text encoding of the empty prompt; '
                         'pass uc_emb=embed_text("") (funcs.py:199-203)')
    raise NotImplementedError(f"uncond_type {utype!r}")


def base_ddim_sampling(model, cond, noise_shape, ddim_steps=50, ddim_eta=1.0, cfg_scale=1.0, uc_emb=None,
                       latents_dir=None, x_T=None, noises=None, use_graph=True):
    """funcs.py:177-241: returns (batch_images, ddim_sampler, samples) like the reference; batch_images is the VAE
    decode of the samples (:239) when the model was built with `first_stage_config`, else None.  `uc_emb` replaces
    model.get_learned_conditioning([""]) (the text encoder is out of scope)."""
    sampler = DDIMSampler(model)
    sampler.use_graph = use_graph            # one hipGraph per DDIM step (fifo_graph.BaseEngine) vs host-issued p_sample_ddim
    uc = None
    if cfg_scale != 1.0:
        c_emb = cond["c_crossattn"][0] if isinstance(cond, dict) else cond
        uc_emb = uncond_embedding(model, c_emb, uc_emb)   # model.uncond_type (:199-206)
        if isinstance(cond, dict):
            uc = {key: cond[key] for key in cond.keys()}
            uc.update({'c_crossattn': [uc_emb]})
        else:
            uc = uc_emb
    samples, _ = sampler.sample(S=ddim_steps, conditioning=cond, batch_size=noise_shape[0], shape=noise_shape[1:],
                                verbose=False, unconditional_guidance_scale=cfg_scale, unconditional_conditioning=uc,
                                eta=ddim_eta, x_T=x_T, latents_dir=latents_dir, noises=noises)
    sampler.release()                        # the step graph's buffers: the FIFO stage that follows builds its own plan
    images = model.decode_first_stage_2DAE(samples) if getattr(model, "first_stage_model", None) is not None else None
    return images, sampler, samples


def fifo_ddim_sampling(args, model, conditioning, noise_shape, ddim_sampler, cfg_scale=1.0, uc_emb=None,
                       latents=None, latents_dir=None, conditioned_image=None, masks=None, gamma=0.5, emit=None,
                       n_iterations=None, batch_windows=True, noises=None, shift_noises=None, decode=False, decode_batch=8,
                       davis_data=None, anchor_noises=None, sam_masks=None, sam_masks_fn=None, targets=None, use_graph=True,
                       seed=None, sam_capacity=None, **kwargs):
    """funcs.py:243-373: returns the list of emitted latent frames [B,4,1,h,w]; with decode=True (and a model built with
    `first_stage_config`) the list of decoded frames [B,3,1,8h,8w] instead -- `model.decode_first_stage_2DAE` of funcs.py:360,
    run on `decode_batch` emitted frames at a time rather than once per iteration.

    Injection masks, as in the reference: with `davis_data = (frames, masks)` (DAVIS-video mode) the masks come from it and every
    queue shift takes the DAVIS branch of `shift_latents` (funcs.py:101-118,368-369); `masks` [B,1,Q,h,w] handed in directly play
    the same role (`davis_masks` of ddim_step: factor 1.5 / 1.0, every timestep).  WITHOUT either the reference's `ddim_step`
    takes its segmentation branch (ddim.py:592-606: only frames with t <= 300, IoU fallback, > 80 % reset, factor 2) and asks
    Grounded-SAM-2 for masks -- out of scope here, so they come in as `sam_masks[i][w]` (iteration i, window w in the reference's
    call order: the list over frames of [n,h,w] candidate masks; or a callable (i, w) -> that list) or from `sam_masks_fn(pred_x0_frame, targets, frame) -> [n,h,w]`
    (see DDIMSampler.ddim_step); with neither, nothing is injected.  Candidate LISTS run inside the iteration graph
    (`moca_sam_select_masks_f32`; `sam_capacity` = the most candidate masks one iteration may hold, default: counted when
    `sam_masks` is a list, else 4 per window frame); a producer CALLBACK needs pred_x0 on the host frame by frame and stays on the
    host-driven loop.

    batch_windows=True evaluates the 2n windows of an iteration as one batched UNet launch; with `use_graph` (and a
    call `FifoEngine.supported` accepts) the WHOLE iteration -- gather, UNet, guidance, ddim_step of all windows, write-back,
    emission, FreeInit mix, shift -- is one hipGraph on a device-resident ring queue (fifo_graph.py) and the host never
    synchronises inside the loop; `latents` / `masks` are updated in place at the end like the reference's tensors.
    `noises[i][w]` / `shift_noises[i]` optionally fix the per-window DDIM noise and the enqueued noise (else: device Philox
    stream keyed by `seed` on the graph path, torch.randn on the host path)."""
    kwargs.update({"clean_cond": True})
    cond = conditioning
    uc = None
    if cfg_scale != 1.0:
        uc_emb = uncond_embedding(model, cond["c_crossattn"][0], uc_emb)       # :268-270
        uc = {key: cond[key] for key in cond.keys()}
        uc.update({'c_crossattn': [uc_emb]})
    if davis_data is not None:
        davis_data = (davis_data[0].to(model.device), davis_data[1].to(model.device))     # :296-300
        masks = davis_data[1]
    if latents is None:
        latents = prepare_latents(args, latents_dir, ddim_sampler, model=model, data=davis_data)
    f = args.video_length
    timesteps = ddim_sampler.ddim_timesteps
    indices = np.arange(args.num_inference_steps)
    if args.lookahead_denoising:
        timesteps = np.concatenate([np.full((f // 2,), timesteps[0]), timesteps])
        indices = np.concatenate([np.full((f // 2,), 0), indices])
    total = args.new_video_length + args.num_inference_steps - f if n_iterations is None else n_iterations
    frames, pending = [], []

    sam_in_graph = sam_masks is not None and masks is None                      # (with masks the DAVIS branch wins, ddim.py:565)
    if (batch_windows and use_graph and FifoEngine.supported(model, cond, latents, davis_data=davis_data, sam_masks_fn=sam_masks_fn) and
            (masks is None or masks.shape[2] == latents.shape[2])):
        if seed is None:
            seed = int(torch.randint(0, 2 ** 62, (1,)).item())                # follows torch.manual_seed like the randn draws it replaces
        moments = None
        if davis_data is not None:                       # :101-108: the anchor frame never changes -> its posterior moments, once
            last = davis_data[0][:, :, -1]
            if last.shape[1] == 4:
                last = last[:, :3]
            moments = model.first_stage_model.encode(last.to(latents.device)).parameters
        sam_of = (lambda i: None) if not sam_in_graph else \
            (lambda i: [sam_masks(i, w) for w in range(len(list(fifo_windows(args))))]) if callable(sam_masks) else (lambda i: sam_masks[i])
        if sam_in_graph and sam_capacity is None:
            n_wf = len(list(fifo_windows(args))) * f
            # (a list: the exact maximum over the iterations, an iteration / window / frame entry may be None; a callable: a guess of 4
            #  candidate masks per window frame -- pass `sam_capacity` explicitly when the producer can return more: an iteration that
            #  exceeds the pool raises ValueError from FifoEngine._upload_sam, after earlier frames have been emitted)
            sam_capacity = 4 * n_wf if callable(sam_masks) else max(
                [1] + [sum(0 if c is None else int(torch.as_tensor(c).reshape(-1, latents.shape[-2], latents.shape[-1]).shape[0])
                           for cw in it if cw is not None for c in cw) for it in sam_masks[:total] if it is not None])
        eng = FifoEngine(args, model, ddim_sampler, cond, uc, cfg_scale, latents, conditioned_image=conditioned_image, masks=masks,
                         n_slots=decode_batch if decode else max(total, 1), seed=seed, anchor_moments=moments,
                         scale_factor=getattr(model, "scale_factor", 1.0), sam_capacity=sam_capacity if sam_in_graph else 0)
        try:
            for i in range(total):
                sm = sam_of(i)
                if noises is not None or shift_noises is not None or anchor_noises is not None:
                    nz = noises[i] if noises is not None else [torch.randn(noise_shape, device=latents.device) for _ in eng.wins]
                    sn = shift_noises[i] if shift_noises is not None else torch.randn_like(latents[:, :, -1])
                    an = None
                    if moments is not None:
                        an = anchor_noises[i] if anchor_noises is not None else torch.randn_like(latents[:, :, -1])
                    eng.step(noise=nz, shift_noise=sn, anchor_noise=an, sam_masks=sm)
                else:
                    eng.step(sam_masks=sm)
                if decode and ((i + 1) % decode_batch == 0 or i + 1 == total):
                    i0 = (i // decode_batch) * decode_batch
                    img = model.decode_first_stage_2DAE(eng.emitted_frames(i0, i + 1))
                    frames.extend(img[:, :, [k]] for k in range(img.shape[2]))
            if not decode and total > 0:
                z = eng.emitted_frames(0, total)
                frames = [z[:, :, [k]].clone() if emit is None else emit(z[:, :, [k]].clone()) for k in range(total)]
            latents.copy_(eng.latents().to(latents.dtype))
            if masks is not None:
                masks.copy_(eng.mask_queue().to(masks.dtype))
        finally:
            eng.close()
        return frames

    for i in range(total):
        wins = list(fifo_windows(args))
        eps_list = None
        if batch_windows:
            # N2: all windows of this iteration as one batched UNet launch (they are independent, see unet_windows)
            eps_list = ddim_sampler.unet_windows([latents[:, :, s0:e0].clone() for s0, _, e0 in wins], cond,
                                                 [timesteps[s0:e0] for s0, _, e0 in wins],
                                                 unconditional_guidance_scale=cfg_scale, unconditional_conditioning=uc, **kwargs)
        for wi, (start, mid, end) in enumerate(wins):
            t, idx = timesteps[start:end], indices[start:end]
            input_latents = latents[:, :, start:end].clone()
            input_masks = masks[:, :, start:end].clone() if masks is not None else None
            noise = None if noises is None else noises[i][wi]
            sam = dict(sam_masks=None if sam_masks is None else (sam_masks(i, wi) if callable(sam_masks) else sam_masks[i][wi]),
                       sam_masks_fn=sam_masks_fn)
            if eps_list is not None:
                ts_t = torch.as_tensor(np.asarray(t).copy(), device=latents.device).to(torch.long)
                output_latents, _ = ddim_sampler.ddim_step(input_latents, eps_list[wi], idx, conditioned_image, targets, ts_t,
                                                           davis_masks=input_masks, noise=noise, **sam)
            else:
                output_latents, _ = ddim_sampler.fifo_onestep(cond=cond, shape=noise_shape, latents=input_latents, timesteps=t,
                                                              indices=idx, unconditional_guidance_scale=cfg_scale,
                                                              unconditional_conditioning=uc, cond_image=conditioned_image,
                                                              target=targets, davis_masks=input_masks, noise=noise, **sam, **kwargs)
            if args.lookahead_denoising:
                latents[:, :, mid:end] = output_latents[:, :, -(f // 2):]
            else:
                latents[:, :, start:end] = output_latents
        first = f // 2 if args.lookahead_denoising else 0
        frame = latents[:, :, [first]].clone()
        if decode:
            pending.append(frame)
            if len(pending) == decode_batch:
                frames.extend(decode_frames(model, pending))
                pending = []
        else:
            frames.append(frame if emit is None else emit(frame))
        sn = None if shift_noises is None else shift_noises[i]
        if davis_data is not None:                                                  # :367-369
            latents, davis_data = shift_latents(latents, davis_data, model, noise=sn,
                                                anchor_noise=None if anchor_noises is None else anchor_noises[i])
        else:
            latents = shift_latents(latents, noise=sn)
            if masks is not None:            # masks handed in directly (prompt mode + precomputed masks): keep them aligned
                masks[:, :, :-1] = masks[:, :, 1:].clone()
    if pending:
        frames.extend(decode_frames(model, pending))
    return frames


def decode_frames(model, latent_frames):
    """`model.decode_first_stage_2DAE` (ddpm3d.py:556-562) over a list of emitted latent frames [B,4,1,h,w]:
    one recorded decoder launch for all of them; returns the list of [B,3,1,8h,8w] tensors."""
    z = torch.cat(latent_frames, dim=2)
    img = model.decode_first_stage_2DAE(z)
    return [img[:, :, [i]] for i in range(img.shape[2])]


def tensor2image(batch_tensors):
    """funcs.py:630-640: [1,3,1,H,W] in [-1,1] -> uint8 [H,W,3] (a PIL image when Pillow is importable)"""
    img = torch.squeeze(batch_tensors).detach().cpu()
    img = torch.clamp(img.float(), -1., 1.)
    img = (img + 1.0) / 2.0
    arr = (img * 255).to(torch.uint8).permute(1, 2, 0).numpy()
    try:
        from PIL import Image
        return Image.fromarray(arr)
    except ImportError:
        return arr
