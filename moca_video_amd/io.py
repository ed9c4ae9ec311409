"""Prompt-file / hand-off formats either side of the hot path: the reference's prompt CSV
(`scripts/evaluation/funcs.py:506-535`), its rank striding (`videocrafter_main.py:179-181`), the output/latent
directory convention (`videocrafter_main.py:25-55`; the `{0,N}.pt` latent cache itself is written by
`DDIMSampler.ddim_sampling` and read by `prepare_latents`, sampler.py / fifo.py), the frame/GIF writers
(`funcs.py:614-640`) and the prompt-mode driver loop (`videocrafter_main.py:176-232`) with the two external models --
text encoder and Grounded-SAM-2 -- replaced by caller-supplied callables (both are out of scope).
Host-side Python, as in the reference; nothing here computes on the GPU except through the drop-in classes.
"""
from __future__ import annotations

import csv
import os

import torch

from .fifo import base_ddim_sampling, fifo_ddim_sampling, fifo_windows, prepare_latents, tensor2image
from .sampler import DDIMSampler

_FIELDS = ("prompt", "conditioned_object", "conditioned_image_path", "conditioned_prompt", "gamma")


def load_prompts(prompt_file, prompt_index=None):
    """funcs.py:506-535: rows of `prompt,conditioned_object,conditioned_image_path,conditioned_prompt,gamma`;
    `conditioned_prompt` gets a trailing '.', `gamma` is a float; `prompt_index` selects one row (ValueError past the end)."""
    def row_to_dict(row):
        return {"prompt": row["prompt"].strip(), "conditioned_object": row["conditioned_object"].strip(),
                "conditioned_image_path": row["conditioned_image_path"].strip(),
                "conditioned_prompt": row["conditioned_prompt"].strip() + ".", "gamma": float(row["gamma"].strip())}
    with open(prompt_file, "r") as f:
        reader = csv.DictReader(f)
        if prompt_index is not None:
            for i, row in enumerate(reader):
                if i == prompt_index:
                    return [row_to_dict(row)]
            raise ValueError(f"Prompt index {prompt_index} exceeds number of available prompts")
        return [row_to_dict(row) for row in reader]


def shard_indices(num_samples, rank, num_processes):
    """videocrafter_main.py:179-181: `indices[rank::num_processes]`"""
    return list(range(num_samples))[rank::num_processes]


def set_directory(args, prompt, conditioned_image_path=None, root="."):
    """videocrafter_main.py:25-55 (prompt mode): returns (output_dir, latents_dir), both created"""
    p = prompt[:100]
    if getattr(args, "output_dir", None) is None:
        out = f"results/videocraft_v2_fifo/random_noise/{'self_attention' if getattr(args, 'use_self_attention', False) else 'sam2'}/{p}"
        if args.eta != 1.0:
            out += f"/eta{args.eta}"
        if args.new_video_length != 100:
            out += f"/{args.new_video_length}frames"
        if not args.lookahead_denoising:
            out = out.replace(p, f"{p}/no_lookahead_denoising")
        if args.num_partitions != 4:
            out = out.replace(p, f"{p}/n={args.num_partitions}")
        if args.video_length != 16:
            out = out.replace(p, f"{p}/f={args.video_length}")
    else:
        out = args.output_dir
    lat = f"results/videocraft_v2_fifo/latents/{args.num_inference_steps}steps/{p}/eta{args.eta}"
    out, lat = os.path.join(root, out), os.path.join(root, lat)
    os.makedirs(out, exist_ok=True)
    os.makedirs(lat, exist_ok=True)
    return out, lat


def load_cond_image(path, height, width, device="cuda"):
    """videocrafter_main.py:89-98: the conditioning image is NOT VAE-encoded in prompt mode -- it is the RGBA image
    resized to the latent grid (PIL bilinear, as torchvision.transforms.Resize does on a PIL image; the CenterCrop to
    the same size is a no-op), scaled to [0,1] (ToTensor) and used as a 4-channel latent frame [1,4,1,h/8,w/8]."""
    import numpy as np
    from PIL import Image
    h, w = height // 8, width // 8
    img = Image.open(path).convert("RGBA").resize((w, h), Image.BILINEAR)
    t = torch.from_numpy(np.asarray(img, dtype=np.uint8).copy()).permute(2, 0, 1).float().div(255.0)
    return t.unsqueeze(1).unsqueeze(0).to(device)


def load_masks(mask_dir, num_frames, height, width, device="cuda", ext=("png", "npy")):
    """Mask-producer interface: precomputed Grounded-SAM-2 / DAVIS masks from disk instead of the
    segmentation models.  `mask_dir/{i}.png` (8-bit, > 127 = object) or `{i}.npy` ([n,H,W] or [H,W]) per frame i; a
    missing file = no detection for that frame.  Masks are resized (nearest) to the latent grid.  Returns the list the
    `sam_masks=` argument of `DDIMSampler.ddim_step` takes; `torch.stack` of single-mask frames gives `davis_masks`."""
    import numpy as np
    from PIL import Image
    h, w = height // 8, width // 8
    out = []
    for i in range(num_frames):
        m = None
        for e in ext:
            path = os.path.join(mask_dir, f"{i}.{e}")
            if os.path.exists(path):
                if e == "npy":
                    a = np.load(path).astype(np.float32)
                    a = a[None] if a.ndim == 2 else a
                    m = torch.stack([torch.from_numpy(np.asarray(Image.fromarray((x > 0.5).astype(np.uint8) * 255)
                                                                .resize((w, h), Image.NEAREST), dtype=np.float32) / 255.0) for x in a])
                else:
                    im = Image.open(path).convert("L").resize((w, h), Image.NEAREST)
                    m = (torch.from_numpy(np.asarray(im, dtype=np.float32)) > 127).float()[None]
                break
        out.append(None if m is None else m.to(device))
    return out


def load_davis_data(video_name, davis_root, frame_stride=1, video_size=(40, 64), video_frames=16, sampling_strategy="first",
                    seed=None):
    """funcs.py:643-735: frames `JPEGImages/480p/<video>/*.jpg` and masks `Annotations/480p/<video>/*.png` of a DAVIS
    sequence -> (frames [1,4,T,8h,8w] RGBA in [-1,1], masks [1,1,T,h,w] in {0,1}); `video_size` = the LATENT grid (h, w).
    Frames are resized with Pillow's LANCZOS (the reference uses cv2.INTER_LANCZOS4; cv2 is not in this image), masks with
    nearest neighbour; "first" / "random" / "uniform" pick the frame indices as upstream."""
    import numpy as np
    from PIL import Image
    frames_dir = os.path.join(davis_root, "JPEGImages", "480p", video_name)
    masks_dir = os.path.join(davis_root, "Annotations", "480p", video_name)
    frame_files = sorted(f for f in os.listdir(frames_dir) if f.endswith(".jpg"))
    mask_files = sorted(f for f in os.listdir(masks_dir) if f.endswith(".png"))
    total = len(frame_files)
    if sampling_strategy == "first":
        idxs = list(range(min(video_frames, total)))
    elif sampling_strategy == "random":
        rng = np.random.default_rng(seed)
        idxs = sorted(rng.choice(total, size=min(video_frames, total), replace=False).tolist())
    elif sampling_strategy == "uniform":
        idxs = list(range(total)) if total <= video_frames else list(range(0, total, max(1, total // video_frames)))[:video_frames]
    else:
        raise ValueError(f"Unknown sampling strategy: {sampling_strategy}")
    H, W = video_size[0] * 8, video_size[1] * 8
    frames, masks = [], []
    for i in idxs:
        fr = Image.open(os.path.join(frames_dir, frame_files[i])).convert("RGBA")
        if fr.size != (W, H):
            fr = fr.resize((W, H), Image.LANCZOS)
        t = torch.from_numpy(np.asarray(fr, dtype=np.uint8).copy()).permute(2, 0, 1).float()
        frames.append((t / 255. - 0.5) * 2)
        mk = Image.open(os.path.join(masks_dir, mask_files[i])).convert("L")
        if mk.size != (video_size[1], video_size[0]):
            mk = mk.resize((video_size[1], video_size[0]), Image.NEAREST)
        masks.append(np.asarray(mk, dtype=np.uint8))
    frames = torch.stack(frames).unsqueeze(0).permute(0, 2, 1, 3, 4).contiguous()
    masks = torch.tensor(np.stack(masks)).unsqueeze(1).float().unsqueeze(0).permute(0, 2, 1, 3, 4)
    return frames, (masks > 0).float().contiguous()


def frames_to_uint8(batch_tensors):
    """funcs.py:614-622: [1,3,f,H,W] in [-1,1] -> uint8 [f,H,W,3]"""
    video = torch.squeeze(batch_tensors, 0) if batch_tensors.dim() == 5 else batch_tensors
    video = torch.clamp(video.detach().cpu().float(), -1., 1.).permute(1, 0, 2, 3)
    video = (video + 1.0) / 2.0
    return (video * 255).to(torch.uint8).permute(0, 2, 3, 1).numpy()


def save_gif(batch_tensors, savedir, name, duration_ms=100):
    """funcs.py:614-628 (imageio.mimsave -> Pillow, which this image has)"""
    from PIL import Image
    arr = frames_to_uint8(batch_tensors)
    imgs = [Image.fromarray(a) for a in arr]
    path = os.path.join(savedir, f"{name}.gif")
    imgs[0].save(path, save_all=True, append_images=imgs[1:], duration=duration_ms, loop=0)
    return path


def save_frames(images, savedir, ext="png"):
    """`{i}.png` per FIFO iteration (funcs.py:362-364); `images` = PIL images or uint8 arrays"""
    from PIL import Image
    os.makedirs(savedir, exist_ok=True)
    for i, im in enumerate(images):
        (im if hasattr(im, "save") else Image.fromarray(im)).save(os.path.join(savedir, f"{i}.{ext}"))


def _empty_prompt_embedding(model, embed_text, uc_emb):
    """the unconditional context of the drivers: `uncond_type == "empty_seq"` -> the text encoding of "" (funcs.py:199-203)"""
    if uc_emb is None and getattr(model, "uncond_type", "empty_seq") == "empty_seq":
        return embed_text("")
    return uc_emb


def run_prompts(args, model, embed_text, cond_image_fn=None, mask_fn=None, root=".", uc_emb=None, decode=True, rank=None,
                num_processes=None, n_iterations=None, sam_masks_fn=None):
    """The prompt-mode loop of videocrafter_main.py:176-232 on the drop-in classes.

    embed_text(str) -> [1,77,1024] replaces `model.get_learned_conditioning` (OpenCLIP, out of scope);
    cond_image_fn(row) -> conditioned-image latents [1,4,1,h,w].  Prompt mode reaches `ddim_step` WITHOUT DAVIS masks, i.e. its
    segmentation branch (ddim.py:592-606 -> `_apply_segmentation` :739-903: frames with t <= 300 only, IoU fallback, > 80 % reset,
    factor 2); the Grounded-SAM-2 producer it calls is out of scope and comes in as either
      sam_masks_fn(pred_x0_frame [1,4,1,h,w], target, frame) -> candidate masks [n,h,w] or None  (the producer interface), or
      mask_fn(row, shape) -> [1,1,Q,h,w]: one precomputed object mask per queue frame (e.g. `load_masks`), moving with the queue;
      the mask of a window frame is offered as that frame's single candidate.
    Rows go to `rank` by `indices[rank::num_processes]`.  Returns {row index: output path}."""
    rows = load_prompts(args.prompt_file, getattr(args, "prompt_index", None))
    uc_emb = _empty_prompt_embedding(model, embed_text, uc_emb)
    rank = getattr(args, "rank", 0) if rank is None else rank
    nproc = getattr(args, "num_processes", 1) if num_processes is None else num_processes
    h, w = args.height // 8, args.width // 8
    done = {}
    for idx in shard_indices(len(rows), rank, nproc):
        data = rows[idx]
        out_dir, lat_dir = set_directory(args, data["prompt"], data["conditioned_image_path"], root=root)
        noise_shape = [1, 4, args.video_length, h, w]
        fps = torch.tensor([args.fps], device=model.device).long()
        cond = {"c_crossattn": [embed_text(data["prompt"])], "fps": fps}
        cached = os.path.exists(f"{lat_dir}/{args.num_inference_steps}.pt") and os.path.exists(f"{lat_dir}/0.pt")
        if cached:
            sampler = DDIMSampler(model)
            sampler.make_schedule(ddim_num_steps=args.num_inference_steps, ddim_eta=args.eta, verbose=False)
        else:
            base, sampler, _ = base_ddim_sampling(model, cond, noise_shape, args.num_inference_steps, args.eta,
                                                  args.unconditional_guidance_scale, uc_emb=uc_emb, latents_dir=lat_dir)
            if decode and base is not None:
                save_gif(base, out_dir, "origin")
        if data["conditioned_prompt"]:
            cond["c_crossattn"].append(embed_text(data["conditioned_prompt"]))
        Q = args.num_inference_steps + (args.video_length // 2 if args.lookahead_denoising else 0)
        cimg = cond_image_fn(data) if cond_image_fn is not None else None
        sam = None
        if mask_fn is not None:
            mq = mask_fn(data, (1, 1, Q, h, w))[0, 0]
            wins = list(fifo_windows(args))

            def sam(i, wi, _mq=mq, _wins=wins):       # the mask queue moves with the latent queue: one shift per iteration, tail kept
                s0, _, e0 = _wins[wi]
                return [_mq[min(s0 + j + i, _mq.shape[0] - 1)][None] for j in range(e0 - s0)]
        frames = fifo_ddim_sampling(args, model, cond, noise_shape, sampler, args.unconditional_guidance_scale, uc_emb=uc_emb,
                                    latents_dir=lat_dir, conditioned_image=cimg, gamma=data["gamma"], sam_masks=sam,
                                    sam_masks_fn=sam_masks_fn, targets=(data.get("conditioned_object") or "") + ".",
                                    decode=decode, n_iterations=n_iterations)
        keep = frames[-args.new_video_length // 2:]                          # videocrafter_main.py:228-230 (verbatim: -N//2 floors)
        if decode:
            path = save_gif(torch.cat(keep, dim=2), out_dir, "fifo", duration_ms=int(1000 / args.output_fps))
        else:
            path = os.path.join(out_dir, "fifo_latents.pt")
            torch.save(torch.cat(keep, dim=2).cpu(), path)
        done[idx] = path
    return done


def run_davis(args, model, embed_text, prompt, cond_image=None, root=".", uc_emb=None, decode=True, n_iterations=None):
    """The DAVIS-video mode of videocrafter_main.py:102-175 on the drop-in classes: frames + annotation masks of
    `args.video_name` -> VAE-encoded 72-frame queue (`prepare_latents(data=...)`, funcs.py:38-48) -> MoCA FIFO sampling with the
    DAVIS masks as injection masks -> the FIRST `new_video_length // 2` emitted frames as a GIF (:170-175).  `prompt` replaces
    `get_davis_prompt(video_name) + " cat."` (annotation file absent offline); `embed_text` as in run_prompts."""
    h, w = args.height // 8, args.width // 8
    f = args.video_length
    Q = args.num_inference_steps + (f // 2 if args.lookahead_denoising else 0)
    frames, masks = load_davis_data(args.video_name, args.davis_root, frame_stride=getattr(args, "frame_stride", 1),
                                    video_size=(h, w), video_frames=Q, sampling_strategy=getattr(args, "sampling_strategy", "uniform"))
    out_dir, lat_dir = set_directory(args, args.video_name, getattr(args, "conditioned_image_path", None), root=root)
    fps = torch.tensor([args.fps], device=model.device).long()
    cond = {"c_crossattn": [embed_text(prompt)], "fps": fps}
    sampler = DDIMSampler(model)
    sampler.make_schedule(ddim_num_steps=args.num_inference_steps, ddim_eta=args.eta, verbose=False)
    uc_emb = _empty_prompt_embedding(model, embed_text, uc_emb)
    masks = masks.to(model.device)
    if masks.shape[2] < Q:                                       # short clips: the remaining queue frames carry no mask
        masks = torch.cat([masks, torch.zeros(1, 1, Q - masks.shape[2], h, w, device=masks.device)], dim=2)
    # davis_data goes into the FIFO loop as in videocrafter_main.py:147-163: the queue is the VAE encoding of the frames
    # (prepare_latents, funcs.py:38-48) and every shift takes the DAVIS branch of shift_latents (funcs.py:101-118)
    out = fifo_ddim_sampling(args, model, cond, [1, 4, f, h, w], sampler, args.unconditional_guidance_scale, uc_emb=uc_emb,
                             latents_dir=lat_dir, conditioned_image=cond_image, gamma=getattr(args, "gamma", 0.5),
                             decode=decode, n_iterations=n_iterations, davis_data=(frames.to(model.device), masks))
    keep = out[:args.new_video_length // 2]
    if decode:
        return save_gif(torch.cat(keep, dim=2), out_dir, args.video_name, duration_ms=int(1000 / args.output_fps))
    path = os.path.join(out_dir, f"{args.video_name}_latents.pt")
    torch.save(torch.cat(keep, dim=2).cpu(), path)
    return path
