#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REAL reference (imported read-only from
/root/reference) on CPU.  Runs only in the build container; the reference never
travels, only these small input-recipe/output fixtures do.

    python tools/make_golden.py [--full]      (--full adds the full-width 1.41 B-param UNet cases)

Inputs and parameters are not stored: they are regenerated bit-identically from
moca_video_amd.weightgen (numpy Philox keyed by tensor name), so a fixture is
{recipe metadata, expected outputs}.
"""
import argparse
import os
import sys
import tempfile
import time
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.dont_write_bytecode = True
sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")


def stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def import_reference():
    """Harness-side stubs for packages the image lacks and the hot path never calls."""
    os.environ.setdefault("MPLBACKEND", "Agg")
    sys.path.insert(0, REF)
    stub("cv2")
    stub("pytorch_lightning", LightningModule=torch.nn.Module)
    tv = stub("torchvision")
    tv.utils = stub("torchvision.utils", make_grid=None, save_image=None)
    tv.transforms = stub("torchvision.transforms")
    tr = stub("transformers", AutoProcessor=None, AutoModelForZeroShotObjectDetection=None)
    stub("sam2")
    stub("sam2.build_sam", build_sam2_video_predictor=None, build_sam2=None)
    stub("sam2.sam2_image_predictor", SAM2ImagePredictor=None)
    stub("decord", VideoReader=None, cpu=None)
    stub("imageio")
    from lvdm.modules.networks import openaimodel3d  # noqa
    from lvdm.modules import attention  # noqa
    return openaimodel3d, attention


def inp(name, shape, seed=0):
    from moca_video_amd.weightgen import gen_tensor
    # 2-D+ tensors from gen_tensor are scaled by fan_in^-0.5; inputs want unit variance
    t = gen_tensor("input:" + name, (int(np.prod(shape)),), seed)   # 1-D, 'input:..' does not end in 'weight' -> 0.1*z
    return (t * 10.0).reshape(shape)


def fill(module, seed=0):
    from moca_video_amd.weightgen import fill_module_
    return fill_module_(module, seed)


def save(name, **arrays):
    os.makedirs(GOLD, exist_ok=True)
    path = os.path.join(GOLD, name + ".npz")
    np.savez_compressed(path, **{k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in arrays.items()})
    print(f"wrote {path} ({os.path.getsize(path) / 1024:.1f} KiB)")


# --------------------------------------------------------------------------------------
def blocks(om, att):
    torch.manual_seed(0)
    with torch.no_grad():
        # ResBlock 64 -> 128 with TemporalConvBlock, b=2, t=4, 6x10
        rb = fill(om.ResBlock(64, 256, 0.0, out_channels=128, dims=2, use_checkpoint=False, use_temporal_conv=True).eval(), 1)
        x = inp("rb.x", (8, 64, 6, 10)); emb = inp("rb.emb", (8, 256))
        save("block_resblock", y=rb(x, emb, batch_size=2), meta=np.array([2, 4, 64, 128, 6, 10, 256]))
        rb2 = fill(om.ResBlock(64, 256, 0.0, out_channels=64, dims=2, use_checkpoint=False, use_temporal_conv=True).eval(), 2)
        save("block_resblock_same", y=rb2(x, emb, batch_size=2))
        # SpatialTransformer 128 ch, 2 heads, ctx 96 dims, 77 tokens
        st = fill(att.SpatialTransformer(128, 2, 64, depth=1, context_dim=96, use_linear=True, use_checkpoint=False).eval(), 3)
        x = inp("st.x", (4, 128, 6, 10)); ctx = inp("st.ctx", (4, 77, 96))
        save("block_spatial_transformer", y=st(x, ctx))
        # TemporalTransformer (linear) 128 ch, and the init_attn flavour (conv1d proj, 8 heads on 64 ch)
        tt = fill(att.TemporalTransformer(128, 2, 64, depth=1, use_linear=True, use_checkpoint=False, only_self_att=True,
                                          relative_position=False, temporal_length=16).eval(), 4)
        x5 = inp("tt.x", (2, 128, 8, 3, 5))
        save("block_temporal_transformer", y=tt(x5))
        ti = fill(att.TemporalTransformer(64, 8, 64, depth=1, use_checkpoint=False, only_self_att=True,
                                          relative_position=False, temporal_length=16).eval(), 5)
        x5 = inp("ti.x", (1, 64, 16, 3, 5))
        save("block_init_attn", y=ti(x5))
        # Downsample / Upsample
        dn = fill(om.Downsample(64, True, dims=2, out_channels=64).eval(), 6)
        up = fill(om.Upsample(64, True, dims=2, out_channels=64).eval(), 7)
        x = inp("ud.x", (3, 64, 6, 10))
        save("block_down_up", down=dn(x), up=up(x))
        # timestep_embedding
        from lvdm.models.utils_diffusion import timestep_embedding
        t = torch.tensor([0, 1, 17, 250, 500, 999])
        save("timestep_embedding", t=t, y=timestep_embedding(t, 320))


REDUCED = dict(in_channels=4, out_channels=4, model_channels=64, attention_resolutions=[4, 2, 1], num_res_blocks=2,
               channel_mult=[1, 2, 4, 4], num_head_channels=64, transformer_depth=1, context_dim=128, use_linear=True,
               use_checkpoint=False, temporal_conv=True, temporal_attention=True, temporal_selfatt_only=True,
               use_relative_position=False, use_causal_attention=False, temporal_length=16, addition_attention=True,
               fps_cond=True)


def unet_cases(om, params, tag, shape_x, ctx_dim, cases):
    t0 = time.time()
    model = om.UNetModel(**params).eval()
    fill(model, 11)
    print(f"[{tag}] reference UNet built+filled in {time.time() - t0:.1f}s "
          f"({sum(p.numel() for p in model.parameters()) / 1e6:.1f} M params)")
    out = {}
    with torch.no_grad():
        for name, B, tvals, L, fps in cases:
            x = inp(f"{tag}.{name}.x", (B,) + shape_x)
            ctx = inp(f"{tag}.{name}.ctx", (B, L, ctx_dim))
            t = torch.tensor(tvals, dtype=torch.long)
            f = fps if isinstance(fps, int) else torch.tensor(fps, dtype=torch.long)
            t1 = time.time()
            y = model(x, t, context=ctx, fps=f, clean_cond=True, gamma=0.5)   # unknown kwargs are swallowed (:534)
            print(f"[{tag}] {name}: forward {time.time() - t1:.1f}s, out std {y.std():.4f}")
            out[name] = y
            out[name + "__t"] = t
            out[name + "__fps"] = np.asarray(fps)
            out[name + "__L"] = np.asarray(L)
    save(f"unet_{tag}", **out)
    del model


def unet_reduced(om):
    T = 8
    unet_cases(om, REDUCED, "reduced", (4, T, 16, 16), 128, [
        ("uniform", 1, [500], 77, 16),
        ("fifo", 1, [int(v) for v in np.linspace(999, 0, T).round()], 154, [10]),
        ("batch2", 2, [981, 20], 77, [10, 24]),
    ])


def unet_full(om, only_cfgN=False):
    import yaml
    with open(os.path.join(REF, "configs/inference_t2v_512_v2.0.yaml")) as f:
        params = yaml.safe_load(f)["model"]["params"]["unet_config"]["params"]
    params = dict(params)
    params["use_checkpoint"] = False   # no effect on results under no_grad (common.py:80-94)
    if not only_cfgN:
        unet_cases(om, params, "full", (4, 8, 32, 32), 1024, [
            ("cfg0_uniform", 1, [500], 77, [10]),
            ("cfg0_fifo", 1, [int(v) for v in np.linspace(999, 0, 8).round()], 154, [10]),
        ])
    # cfgN: the headline shape, FIFO window call
    unet_cases(om, params, "full_cfgN", (4, 16, 40, 64), 1024, [
        ("fifo16", 1, [int(v) for v in np.linspace(999, 0, 16).round()], 77, [10]),
        ("fifo16_154", 1, [int(v) for v in np.linspace(999, 0, 16).round()], 154, [10]),     # the MoCA FIFO call: two prompts
        ("uniform_77", 1, [500], 77, [10]),                                                     # the base-sampling call (configs[1])
    ])


def unet_full_b16(om):
    """The FIFO-iteration operating point (funcs.py:305-355): the windows of one outer iteration, each run through the REAL
    reference UNet twice as `DDIMSampler.unet` does (ddim.py:362-374) -- 154-token two-prompt context and 77-token
    unconditional context on the SAME latents / per-frame timesteps.  Two distinct windows (their own latents and 16
    consecutive timesteps of the 50-step schedule); the B = 16 test tiles them over the 8 window rows of the batched plan."""
    import yaml
    with open(os.path.join(REF, "configs/inference_t2v_512_v2.0.yaml")) as f:
        params = dict(yaml.safe_load(f)["model"]["params"]["unet_config"]["params"])
    params["use_checkpoint"] = False
    t0 = time.time()
    model = om.UNetModel(**params).eval()
    fill(model, 11)
    print(f"[full_b16] reference UNet built+filled in {time.time() - t0:.1f}s")
    sched = np.linspace(0, 999, 50).round().astype(np.int64)[::-1]          # descending, as the queue holds them
    out = {}
    ctxs = {154: inp("full_b16.ctx154", (1, 154, 1024)), 77: inp("full_b16.ctx77", (1, 77, 1024))}
    with torch.no_grad():
        for w, lo in (("w0", 4), ("w1", 30)):
            x = inp(f"full_b16.{w}.x", (1, 4, 16, 40, 64))
            t = torch.from_numpy(np.ascontiguousarray(sched[lo:lo + 16]))
            out[f"{w}__t"] = t
            for L, ctx in ctxs.items():
                t1 = time.time()
                y = model(x, t, context=ctx, fps=torch.tensor([10]), clean_cond=True, gamma=0.5)
                print(f"[full_b16] {w} L={L}: forward {time.time() - t1:.1f}s, out std {y.std():.4f}")
                out[f"{w}_{L}"] = y
    out["fps"] = np.asarray([10])
    save("unet_full_b16", **out)
    del model


# --------------------------------------------------------------------------------------
class FakeModel:
    """What DDIMSampler reads from LatentDiffusion; apply_model returns queued eps tensors."""

    def __init__(self):
        from lvdm.models.utils_diffusion import make_beta_schedule
        betas = make_beta_schedule("linear", 1000, linear_start=0.00085, linear_end=0.012)
        ac = np.cumprod(1. - betas, axis=0)
        self.num_timesteps = 1000
        self.betas = torch.tensor(betas, dtype=torch.float32)
        self.alphas_cumprod = torch.tensor(ac, dtype=torch.float32)
        self.alphas_cumprod_prev = torch.tensor(np.append(1., ac[:-1]), dtype=torch.float32)
        self.use_scale = True
        self.scale_arr = torch.tensor(np.concatenate((np.linspace(1, 0.7, 400), np.full(1000, 0.7))), dtype=torch.float32)
        self.device = torch.device("cpu")
        self.parameterization = "eps"
        self.queue = []

    def apply_model(self, x, t, c, **kw):
        return self.queue.pop(0)


def sampler_cases():
    from lvdm.models.samplers import ddim as D
    D.DDIMSampler.register_buffer = lambda self, name, attr: setattr(self, name, attr)   # drop hard-coded .to("cuda") (:53-60)
    fm = FakeModel()
    out = {}
    for S in (10, 50, 64):
        s = D.DDIMSampler(fm, use_self_attention=True)
        s.make_schedule(S, ddim_eta=1.0, verbose=False)
        for k in ("ddim_timesteps", "ddim_sigmas", "ddim_alphas", "ddim_alphas_prev", "ddim_sqrt_one_minus_alphas",
                  "ddim_scale_arr", "ddim_scale_arr_prev"):
            out[f"S{S}_{k}"] = np.asarray(getattr(s, k))
    save("sampler_schedule", **out)

    # p_sample_ddim with CFG, captured noise
    s = D.DDIMSampler(fm, use_self_attention=True)
    s.make_schedule(50, ddim_eta=1.0, verbose=False)
    shape = (1, 4, 8, 16, 24)
    res = {}
    for index in (49, 20, 0):
        x = inp(f"ps.x{index}", shape); e_c = inp(f"ps.ec{index}", shape); e_u = inp(f"ps.eu{index}", shape)
        nz = inp(f"ps.nz{index}", shape)
        fm.queue = [e_c.clone(), e_u.clone()]
        D.noise_like = lambda shp, dev, repeat=False, _n=nz: _n.clone()
        t = torch.full((1,), int(s.ddim_timesteps[index]), dtype=torch.long)
        xp, p0 = s.p_sample_ddim(x, {"c_crossattn": [None]}, t, index, unconditional_guidance_scale=12.0,
                                 unconditional_conditioning={"c_crossattn": [None]})
        res[f"i{index}_x_prev"] = xp; res[f"i{index}_pred_x0"] = p0
    save("sampler_p_sample_ddim", **res)

    # ddim_step (MoCA FIFO step, DAVIS-mask branch), two consecutive calls (momentum state persists)
    cwd = os.getcwd()
    tmp = tempfile.mkdtemp()
    os.chdir(tmp)   # it writes ./visualizations/**
    try:
        for tag, (C, F, H, W), S in (("small", (4, 6, 16, 16), 64), ("cfgN", (4, 16, 40, 64), 64)):
            s = D.DDIMSampler(fm, use_self_attention=True)
            s.make_schedule(S, ddim_eta=1.0, verbose=False)
            res = {}
            # window of the FIFO queue: indices / timesteps as funcs.py:290-312 builds them
            ts_all = np.concatenate([np.full((F // 2,), s.ddim_timesteps[0]), s.ddim_timesteps])
            idx_all = np.concatenate([np.full((F // 2,), 0), np.arange(S)])
            for call, start in enumerate((0, 24)):
                idx = idx_all[start:start + F]; tsn = ts_all[start:start + F]
                ts = torch.Tensor(tsn.copy()).to(dtype=torch.long)
                shape = (1, C, F, H, W)
                x = inp(f"ds.{tag}.x{call}", shape); e = inp(f"ds.{tag}.e{call}", shape)
                noises = [inp(f"ds.{tag}.nz{call}.{i}", (1, C, 1, H, W)) for i in range(F)]
                q = list(noises)
                D.noise_like = lambda shp, dev, repeat=False: q.pop(0).clone()
                cond = (inp(f"ds.{tag}.cond", (1, C, 1, H, W)) * 0.25 + 0.5).clamp(0, 1)
                mask = (inp(f"ds.{tag}.mask", (1, 1, F, H, W)) > 0.5).float()
                mask[:, :, 1] = 0.0   # an all-zero mask frame exercises `mask.sum() != 0`
                t0 = time.time()
                xp, p0 = s.ddim_step(x, e, idx, cond, None, ts, use_self_attention=True, davis_masks=mask)
                print(f"ddim_step {tag} call {call}: {time.time() - t0:.1f}s")
                res[f"c{call}_x_prev"] = xp; res[f"c{call}_pred_x0"] = p0; res[f"c{call}_momentum"] = s.momentum.clone()
                res[f"c{call}_indices"] = idx; res[f"c{call}_ts"] = tsn
            save(f"sampler_ddim_step_{tag}", **res)
    finally:
        os.chdir(cwd)



def sam_candidates(F, H, W, seq):
    """Scripted Grounded-SAM-2 outputs, one entry per frame (None = no box detected), exercising every rule of
    `_apply_segmentation` (ddim.py:739-903); tests/test_sampler_gpu.py::_sam_candidates builds the same lists."""
    def rect(y0, y1, x0, x1):
        m = torch.zeros(H, W)
        m[y0:y1, x0:x1] = 1.0
        return m
    big = torch.ones(H, W)
    if seq == 0:
        cands = [None, rect(2, 10, 3, 12)[None], rect(2, 10, 4, 12)[None], rect(11, 15, 0, 5)[None], None,
                 torch.stack([big, rect(1, 5, 1, 6)]), torch.stack([rect(1, 5, 1, 6), big])]
    else:
        cands = [torch.stack([rect(2, 10, 3, 12), big]), None, rect(2, 10, 4, 12)[None]]
    return (cands + [None] * F)[:F]


def sampler_sam_cases():
    """The segmentation branch of the REAL `ddim_step` (use_self_attention=False, davis_masks=None; ddim.py:592-606 ->
    apply_cond_img -> _apply_segmentation :739-903) with the three external-model attributes replaced by fakes that
    return scripted boxes / masks: the IoU fallback, the > 80 % reset, the factor-2 injection, the t <= 300 gate and the
    `torch.where` broadcast (an injected frame comes back replicated C times along the frame axis) are the reference's
    own code.  The hard-coded `.to("cuda", ...)` of :769-773 is patched to a dtype-only conversion (CPU harness)."""
    from lvdm.models.samplers import ddim as D
    D.DDIMSampler.register_buffer = lambda self, name, attr: setattr(self, name, attr)
    fm = FakeModel()
    real_to = torch.Tensor.to

    def to_nocuda(self, *a, **k):
        if a and isinstance(a[0], str) and a[0] == "cuda":
            return real_to(self, **k) if k else self
        return real_to(self, *a, **k)

    class FakeProcessor:
        def __init__(self, cands):
            self.cands, self.k = cands, 0

        def __call__(self, images=None, text=None, return_tensors="pt"):
            return {"input_ids": torch.zeros(1, 4, dtype=torch.int64), "pixel_values": torch.zeros(1, 3, 2, 2)}

        def post_process_grounded_object_detection(self, outputs, input_ids, box_threshold=0.4, text_threshold=0.3, target_sizes=None):
            c = self.cands[self.k]
            n = 0 if c is None else c.shape[0]
            return [{"boxes": torch.zeros(n, 4)}]

    class FakePredictor:
        def __init__(self, proc):
            self.proc = proc

        def set_image(self, img):
            assert img.ndim == 3 and img.shape[2] == 3 and img.dtype == np.uint8

        def predict(self, point_coords=None, point_labels=None, box=None, multimask_output=False):
            c = self.proc.cands[self.proc.k]
            assert box.shape[0] == c.shape[0]
            return c.numpy().copy(), None, None

    cwd = os.getcwd()
    tmp = tempfile.mkdtemp()
    os.chdir(tmp)
    torch.Tensor.to = to_nocuda
    try:
        C, F, H, W, S = 4, 8, 16, 16, 64
        res = {}
        for seq in (0, 1):
            for rng_name, lo in (("low", 3), ("high", 40)):           # t <= 300 everywhere / t > 300 everywhere
                if rng_name == "high" and seq == 1:
                    continue
                s = D.DDIMSampler(fm, use_self_attention=True)         # True: skips initialize_segmentation_models (:49-50)
                s.make_schedule(S, ddim_eta=1.0, verbose=False)
                cands = sam_candidates(F, H, W, seq)
                proc = FakeProcessor(cands)
                s.processor, s.sam2_predictor, s.grounding_model = proc, FakePredictor(proc), (lambda **kw: None)
                idx = np.arange(lo, lo + F)
                tsn = np.asarray(s.ddim_timesteps)[idx]
                ts = torch.Tensor(tsn.copy()).to(dtype=torch.long)
                shape = (1, C, F, H, W)
                x = inp("ds.small.x0", shape); e = inp("ds.small.e0", shape)
                noises = [inp(f"ds.small.nz0.{i}", (1, C, 1, H, W)) for i in range(F)]
                q = list(noises)
                D.noise_like = lambda shp, dev, repeat=False: q.pop(0).clone()
                cond = (inp("ds.small.cond", (1, C, 1, H, W)) * 0.25 + 0.5).clamp(0, 1)
                # frame counter for the fakes: _apply_segmentation is entered once per frame with t <= 300, in frame order
                orig = s._apply_segmentation
                frames_seen = []

                def counted(pred_x0, cond_image, target, step, pre_masks, _o=orig, _p=proc, _fs=frames_seen):
                    _p.k = len(_fs)
                    out = _o(pred_x0, cond_image, target, step, pre_masks)
                    _fs.append(int(out[0].shape[2]))
                    return out
                s._apply_segmentation = counted
                xp, p0 = s.ddim_step(x, e, idx, cond, "object", ts, use_self_attention=False, davis_masks=None)
                key = f"seq{seq}_{rng_name}"
                res[key + "_x_prev"] = xp; res[key + "_pred_x0"] = p0; res[key + "_indices"] = idx; res[key + "_ts"] = tsn
                res[key + "_frames_out"] = np.asarray(frames_seen if frames_seen else [0])
                print(f"sam {key}: x_prev {tuple(xp.shape)} pred_x0 {tuple(p0.shape)} frames_out {frames_seen}")
        save("sampler_ddim_step_sam", **res)
    finally:
        torch.Tensor.to = real_to
        os.chdir(cwd)


def freeinit_cases():
    from utils.freeinit_utils import freq_mix_3d, get_freq_filter
    out = {}
    for shp in ((1, 4, 1, 40, 64), (1, 4, 16, 40, 64), (1, 2, 3, 5, 7), (1, 4, 8, 32, 32)):
        tag = "x".join(map(str, shp[2:]))
        for ft, (ds, dt) in (("gaussian", (0.25, 0.25)), ("butterworth", (0.25, 0.25)), ("ideal", (0.25, 0.25)),
                             ("box", (0.25, 0.25)), ("gaussian", (0.3, 0.6)), ("box", (0.5, 0.5))):
            lpf = get_freq_filter(shp, "cpu", ft, 4, ds, dt)
            out[f"{tag}_{ft}_{ds}_{dt}_lpf"] = lpf[0, 0]
            x = inp(f"fi.x.{tag}", shp); nz = inp(f"fi.n.{tag}", shp)
            out[f"{tag}_{ft}_{ds}_{dt}_mix"] = freq_mix_3d(x, nz, lpf)
    save("freeinit", **out)


def fifo_cases():
    """prepare_latents / shift_latents (funcs.py:21-118).  They hard-code .to("cuda") and
    torch.randn*: the harness feeds a CPU tensor through a patched Tensor.to and a recorded
    randn_like so the queue construction can be pinned exactly."""
    stub_t = sys.modules["torchvision"]
    sys.modules["torchvision.transforms"] = stub_t.transforms
    from scripts.evaluation import funcs as Fn
    from lvdm.models.samplers import ddim as D
    D.DDIMSampler.register_buffer = lambda self, name, attr: setattr(self, name, attr)   # drop hard-coded .to("cuda") (:53-60)
    fm = FakeModel()
    s = D.DDIMSampler(fm, use_self_attention=True)
    s.make_schedule(64, ddim_eta=1.0, verbose=False)
    args = types.SimpleNamespace(num_inference_steps=64, video_length=16, lookahead_denoising=True)
    z = inp("fifo.z", (1, 4, 16, 8, 12))
    tmp = tempfile.mkdtemp()
    torch.save(z, os.path.join(tmp, "64.pt"))
    noises = []
    real_randn_like = torch.randn_like
    real_to = torch.Tensor.to

    def rec_randn_like(t, *a, **k):
        n = inp(f"fifo.nz{len(noises)}", tuple(t.shape))
        noises.append(n)
        return n.clone()

    def to_nocuda(self, *a, **k):
        if a and isinstance(a[0], str) and a[0] == "cuda":
            return self
        return real_to(self, *a, **k)

    torch.randn_like = rec_randn_like
    torch.Tensor.to = to_nocuda
    try:
        lat = Fn.prepare_latents(args, tmp, s)
        n_prep = len(noises)
        shifted = Fn.shift_latents(lat.clone())
    finally:
        torch.randn_like = real_randn_like
        torch.Tensor.to = real_to
    save("fifo_queue", prepared=lat, shifted=shifted, n_noise_prepare=np.asarray(n_prep), n_noise_total=np.asarray(len(noises)))

    # DAVIS branch of shift_latents (funcs.py:101-118): anchor = encode_first_stage_2DAE of the LAST DAVIS frame (a fresh
    # posterior sample, drawn with torch.randn inside DiagonalGaussianDistribution.sample), mask tail refill.  The model is
    # the real ddpm3d method on a holder with the real (reduced-width) AutoencoderKL.
    from lvdm.models.autoencoder import AutoencoderKL
    from lvdm.models import ddpm3d
    ae = AutoencoderKL(ddconfig=dict(VAE_DD, ch=64), lossconfig={"target": "torch.nn.Identity"}, embed_dim=4).eval()
    fill(ae, seed=5)

    class Holder:
        first_stage_model = ae
        scale_factor = 0.18215
        get_first_stage_encoding = ddpm3d.LatentDiffusion.get_first_stage_encoding
        encode_first_stage_2DAE = ddpm3d.LatentDiffusion.encode_first_stage_2DAE
    h, w, Q = 8, 8, 72
    frames = (inp("fifo.davis.frames", (1, 4, 3, 8 * h, 8 * w)) * 0.5).clamp(-1, 1)          # RGBA, 3 frames
    masks0 = (inp("fifo.davis.masks", (1, 1, Q, h, w)) > 0.3).float()
    lat0 = inp("fifo.davis.lat", (1, 4, Q, h, w))
    real_randn = torch.randn
    draws = []

    def rec_randn(*shape, **k):
        shp = tuple(shape[0]) if len(shape) == 1 and isinstance(shape[0], (tuple, list, torch.Size)) else tuple(shape)
        n = inp(f"fifo.davis.anchor_nz{len(draws)}", shp)
        draws.append(n)
        return n.clone()
    nz_like = []

    def rec_randn_like2(t, *a, **k):
        n = inp(f"fifo.davis.nz{len(nz_like)}", tuple(t.shape))
        nz_like.append(n)
        return n.clone()
    torch.randn = rec_randn
    torch.randn_like = rec_randn_like2
    try:
        with torch.no_grad():
            lat1, (fr1, mk1) = Fn.shift_latents(lat0.clone(), (frames.clone(), masks0.clone()), Holder())
            lat2, (fr2, mk2) = Fn.shift_latents(lat1.clone(), (fr1, mk1.clone()), Holder())
    finally:
        torch.randn = real_randn
        torch.randn_like = real_randn_like
    save("fifo_davis_shift", lat1=lat1, masks1=mk1, lat2=lat2, masks2=mk2, n_anchor_draws=np.asarray(len(draws)),
         n_noise_draws=np.asarray(len(nz_like)), anchor_noise_shape=np.asarray(draws[0].shape))



VAE_DD = dict(double_z=True, z_channels=4, resolution=512, in_channels=3, out_ch=3, ch=128, ch_mult=[1, 2, 4, 4],
              num_res_blocks=2, attn_resolutions=[], dropout=0.0)          # configs/inference_t2v_512_v2.0.yaml:56-70


def vae_cases():
    """AutoencoderKL.decode / decode_first_stage_2DAE of the real reference (autoencoder.py:104-107, ddpm3d.py:556-562):
    a reduced-width decoder (ch=64) on 3 frames of 8x8 latents and the full-width YAML decoder on one 8x8 latent."""
    from lvdm.models.autoencoder import AutoencoderKL
    with torch.no_grad():
        for tag, ch, shape in (("vae_reduced", 64, (1, 4, 3, 8, 8)), ("vae_full_small", 128, (1, 4, 1, 8, 8))):
            dd = dict(VAE_DD, ch=ch)
            ae = AutoencoderKL(ddconfig=dd, lossconfig={"target": "torch.nn.Identity"}, embed_dim=4).eval()
            fill(ae, seed=5)
            z = inp(tag + ":z", shape, seed=5)
            scale = 0.18215
            t0 = time.time()
            zz = 1.0 / scale * z
            out = torch.cat([ae.decode(zz[:, :, i]).unsqueeze(2) for i in range(zz.shape[2])], dim=2)   # ddpm3d.py:559-560
            print(f"{tag}: {time.time() - t0:.1f}s out {tuple(out.shape)} std {out.std():.3f}")
            # encode side (DAVIS mode, funcs.py:47-48): distribution parameters, the mode and a sample with fixed noise
            xs = (1, 3, shape[2], 8 * shape[3], 8 * shape[4])
            img = (inp(tag + ":img", xs, seed=5) * 0.5).clamp(-1, 1)
            nz = inp(tag + ":enc_noise", shape, seed=5)
            moms, modes, samples = [], [], []
            for i in range(xs[2]):
                post = ae.encode(img[:, :, i])
                moms.append(post.parameters.unsqueeze(2)); modes.append(post.mode().unsqueeze(2))
                samples.append((scale * post.sample(noise=nz[:, :, i])).unsqueeze(2))               # ddpm3d.py:465
            save(tag, ch=np.asarray(ch), z_shape=np.asarray(shape), scale_factor=np.asarray(scale), out=out,
                 enc_moments=torch.cat(moms, 2), enc_mode=torch.cat(modes, 2), enc_sample=torch.cat(samples, 2))



def loop_sam_candidates(call, F, H, W):
    """Scripted Grounded-SAM-2 output for the `call`-th ddim_step of a sampling loop (tests/helpers.py builds the same lists)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import loop_sam_candidates as f
    return f(call, F, H, W)


def loop_cases():
    """The REAL sampling loops, end to end, on the real reduced-width UNet inside the real `DiffusionWrapper`, driven through the
    real `LatentDiffusion.apply_model` / `decode_first_stage_2DAE` / `encode_first_stage_2DAE` methods on a holder with the real
    reduced-width `AutoencoderKL`:
      loop_base.npz  `base_ddim_sampling` -> `DDIMSampler.sample` -> `ddim_sampling` (funcs.py:177-241, ddim.py:109-252), S = 10,
                     eta 1, CFG 12, use_scale, `latents_dir` caches 0.pt / 10.pt, decode of the samples;
      loop_fifo.npz  `fifo_ddim_sampling` (funcs.py:243-373), 3 outer iterations, queue of 20 frames (f = 8, 2 partitions,
                     lookahead), cond = two prompts (154 tokens), CFG 12:
                       prompt mode -- no masks: `ddim_step` takes the segmentation branch (ddim.py:592-606) with fake
                       Grounded-SAM-2 objects returning scripted masks, queue from the cached `16.pt`;
                       DAVIS mode -- `davis_data = (frames, masks)`: queue from the VAE encoding of the frames, DAVIS masks in
                       `ddim_step`, DAVIS branch of `shift_latents`.
    Every torch.randn / randn_like / noise_like draw is replaced by a named weightgen tensor (recorded order = the reference's
    call order), `.to("cuda")` is a no-op, `trange` is cut to 3 iterations, the text encoder is a table of named embeddings."""
    import contextlib
    import io
    stub_t = sys.modules["torchvision"]
    sys.modules["torchvision.transforms"] = stub_t.transforms
    from scripts.evaluation import funcs as Fn
    from lvdm.models.samplers import ddim as D
    from lvdm.models import ddpm3d
    from lvdm.models.autoencoder import AutoencoderKL
    D.DDIMSampler.register_buffer = lambda self, name, attr: setattr(self, name, attr)
    D.DDIMSampler.initialize_segmentation_models = lambda self: None      # DDIMSampler(model) would load Grounded-SAM-2 (:49-50)

    unet_cfg = {"target": "lvdm.modules.networks.openaimodel3d.UNetModel", "params": dict(REDUCED)}
    wrapper = ddpm3d.DiffusionWrapper(unet_cfg, "crossattn").eval()
    fill(wrapper.diffusion_model, 11)
    ae = AutoencoderKL(ddconfig=dict(VAE_DD, ch=64), lossconfig={"target": "torch.nn.Identity"}, embed_dim=4).eval()
    fill(ae, seed=5)
    text = {"a prompt": inp("loop.ctx1", (1, 77, 128)), "a conditioned prompt": inp("loop.ctx2", (1, 77, 128)),
            "": inp("loop.uctx", (1, 77, 128))}

    class LoopModel(FakeModel):
        uncond_type = "empty_seq"
        scale_factor = 0.18215
        first_stage_model = ae
        model = wrapper
        apply_model = ddpm3d.LatentDiffusion.apply_model
        decode_first_stage_2DAE = ddpm3d.LatentDiffusion.decode_first_stage_2DAE
        encode_first_stage_2DAE = ddpm3d.LatentDiffusion.encode_first_stage_2DAE
        get_first_stage_encoding = ddpm3d.LatentDiffusion.get_first_stage_encoding

        def get_learned_conditioning(self, prompts):
            return torch.cat([text[p] for p in prompts], 0).clone()

    model = LoopModel()
    real = dict(randn=torch.randn, randn_like=torch.randn_like, to=torch.Tensor.to, noise_like=D.noise_like, trange=Fn.trange)
    counters = {}

    def named(kind, shape):
        k = counters.get(kind, 0)
        counters[kind] = k + 1
        return inp(f"{counters['tag']}.{kind}{k}", tuple(shape))

    def rec_randn(*shape, **k):
        shp = tuple(shape[0]) if len(shape) == 1 and isinstance(shape[0], (tuple, list, torch.Size)) else tuple(shape)
        return named("randn", shp)

    def to_nocuda(self, *a, **k):
        if a and isinstance(a[0], str) and a[0] == "cuda":
            return real["to"](self, **k) if k else self
        return real["to"](self, *a, **k)

    def patch(tag):
        counters.clear()
        counters["tag"] = tag
        torch.randn = rec_randn
        torch.randn_like = lambda t, *a, **k: named("randn_like", t.shape)
        torch.Tensor.to = to_nocuda
        D.noise_like = lambda shp, dev, repeat=False: named("noise_like", shp)
        Fn.trange = lambda n, **k: range(min(n, 3))

    def unpatch():
        torch.randn, torch.randn_like, torch.Tensor.to = real["randn"], real["randn_like"], real["to"]
        D.noise_like, Fn.trange = real["noise_like"], real["trange"]

    cwd = os.getcwd()
    tmp = tempfile.mkdtemp()
    os.chdir(tmp)
    fps = torch.tensor([10])
    shape = [1, 4, 8, 16, 16]
    try:
        # ---------------- base sampling, 10 steps
        patch("loop.base")
        t0 = time.time()
        with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
            cond = {"c_crossattn": [model.get_learned_conditioning(["a prompt"])], "fps": fps}
            lat_dir = os.path.join(tmp, "base")
            os.makedirs(lat_dir)
            images, sampler, samples = Fn.base_ddim_sampling(model, cond, shape, ddim_steps=10, ddim_eta=1.0, cfg_scale=12.0,
                                                             latents_dir=lat_dir, verbose=False)
        unpatch()
        x0pt, xNpt = torch.load(os.path.join(lat_dir, "0.pt")), torch.load(os.path.join(lat_dir, "10.pt"))
        print(f"loop base: {time.time() - t0:.1f}s, samples std {samples.std():.3f} max {samples.abs().max():.2f}, "
              f"draws {dict((k, v) for k, v in counters.items() if k != 'tag')}")
        save("loop_base", samples=samples, images=images, pt0=x0pt, ptN=xNpt, n_randn=np.asarray(counters.get("randn", 0)),
             n_noise_like=np.asarray(counters.get("noise_like", 0)))

        # ---------------- FIFO loops
        args = types.SimpleNamespace(num_inference_steps=16, video_length=8, lookahead_denoising=True, num_partitions=2,
                                     new_video_length=10, save_frames=False)
        cimg = (inp("loop.cimg", (1, 4, 1, 16, 16)) * 0.25 + 0.5).clamp(0, 1)
        out = {}
        for mode in ("prompt", "davis"):
            patch("loop.fifo." + mode)
            lat_dir = os.path.join(tmp, mode)
            os.makedirs(lat_dir)
            torch.save(inp("loop.z16", (1, 4, 8, 16, 16)), os.path.join(lat_dir, "16.pt"))
            s = D.DDIMSampler(model)                                         # videocrafter_main.py:202-204 (cached latents)
            s.make_schedule(ddim_num_steps=16, ddim_eta=1.0, verbose=False)
            cond = {"c_crossattn": [model.get_learned_conditioning(["a prompt"]),
                                    model.get_learned_conditioning(["a conditioned prompt"])], "fps": fps}
            steps, shifts, frames = [], [], []
            orig_step, orig_shift, orig_t2i = s.ddim_step, Fn.shift_latents, Fn.tensor2image

            class Proc:
                cands, k = None, 0

                def __call__(self, images=None, text=None, return_tensors="pt"):
                    return {"input_ids": torch.zeros(1, 4, dtype=torch.int64), "pixel_values": torch.zeros(1, 3, 2, 2)}

                def post_process_grounded_object_detection(self, outputs, input_ids, box_threshold=0.4, text_threshold=0.3, target_sizes=None):
                    c = self.cands[self.k]
                    return [{"boxes": torch.zeros(0 if c is None else c.shape[0], 4)}]

            class Pred:
                def set_image(self, img):
                    assert img.ndim == 3 and img.shape[2] == 3 and img.dtype == np.uint8

                def predict(self, point_coords=None, point_labels=None, box=None, multimask_output=False):
                    c = proc.cands[proc.k]
                    assert box.shape[0] == c.shape[0]
                    return c.numpy().copy(), None, None
            proc = Proc()
            s.processor, s.sam2_predictor, s.grounding_model = proc, Pred(), (lambda **kw: None)
            orig_seg = s._apply_segmentation
            seen = []

            def counted(pred_x0, cond_image, target, step, pre_masks):
                # entered once per frame with t <= 300, in frame order; `seen[-1]` = frames of this ddim_step call so far
                proc.k = seen[-1]["frames"][len(seen[-1]["out"])]
                res = orig_seg(pred_x0, cond_image, target, step, pre_masks)
                seen[-1]["out"].append(int(res[0].shape[2]))
                return res
            s._apply_segmentation = counted

            def spy_step(sample, noise_pred, indices, cond_image, target, ts, **kw):
                call = len(steps)
                F = sample.shape[2]
                proc.cands = loop_sam_candidates(call, F, sample.shape[3], sample.shape[4])
                seen.append({"frames": [i for i in range(F) if int(ts[i]) <= 300], "out": []})
                xp, p0 = orig_step(sample, noise_pred, indices, cond_image, target, ts, **kw)
                # fold the C-fold frame replication of injected frames (torch.where broadcast, see sampler_sam_cases)
                counts = [1] * F
                if kw.get("davis_masks") is None:
                    for fr, n in zip(seen[-1]["frames"], seen[-1]["out"]):
                        counts[fr] = n
                parts, o = [], 0
                for n in counts:
                    blk = p0[:, :, o:o + n]
                    for j in range(1, n):
                        assert torch.equal(blk[:, :, [j]], blk[:, :, [0]])
                    parts.append(blk[:, :, [0]])
                    o += n
                assert o == p0.shape[2]
                steps.append((xp.clone(), torch.cat(parts, 2), np.asarray(counts), s.momentum.clone()))
                return xp, p0
            s.ddim_step = spy_step

            def spy_shift(latents, *a, **k):
                r = orig_shift(latents, *a, **k)
                shifts.append((r[0] if isinstance(r, tuple) else r).clone())
                if isinstance(r, tuple):
                    out.setdefault(mode + "_masks_after", []).append(r[1][1].clone())
                return r

            def spy_t2i(frame_tensor):
                frames.append(frame_tensor.clone())
                return orig_t2i(frame_tensor)
            Fn.shift_latents, Fn.tensor2image = spy_shift, spy_t2i
            davis = None
            if mode == "davis":
                dframes = (inp("loop.davis.frames", (1, 4, 3, 128, 128)) * 0.5).clamp(-1, 1)          # RGBA, 3 frames
                dmasks = (inp("loop.davis.masks", (1, 1, 20, 16, 16)) > 0.3).float()
                dmasks[:, :, 7] = 0.0
                davis = (dframes, dmasks)
            t0 = time.time()
            try:
                with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
                    Fn.fifo_ddim_sampling(args, model, cond, shape, s, cfg_scale=12.0, output_dir=tmp, latents_dir=lat_dir,
                                          save_frames=False, conditioned_image=cimg, targets="object.", gamma=0.5,
                                          use_self_attention=False, davis_data=davis)
            finally:
                Fn.shift_latents, Fn.tensor2image = orig_shift, orig_t2i
                unpatch()
            draws = dict((k, v) for k, v in counters.items() if k != "tag")
            print(f"loop fifo {mode}: {time.time() - t0:.1f}s, {len(steps)} ddim_step calls, {len(shifts)} shifts, draws {draws}, "
                  f"queue max {shifts[-1].abs().max():.2f}")
            for k, v in draws.items():
                out[f"{mode}_n_{k}"] = np.asarray(v)
            out[mode + "_queue"] = torch.stack(shifts)                       # the queue after each iteration's shift
            out[mode + "_frames"] = torch.stack(frames)                      # decoded emitted frame of each iteration
            out[mode + "_x_prev"] = torch.stack([a for a, _, _, _ in steps])
            out[mode + "_pred_x0"] = torch.stack([b for _, b, _, _ in steps])
            out[mode + "_frame_copies"] = np.stack([c for _, _, c, _ in steps])
            out[mode + "_momentum_last"] = steps[-1][3]
            if mode + "_masks_after" in out:
                out[mode + "_masks_after"] = torch.stack(out[mode + "_masks_after"])
        save("loop_fifo", **out)
    finally:
        unpatch()
        os.chdir(cwd)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--full", action="store_true")
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    torch.set_num_threads(8)
    om, att = import_reference()
    todo = a.only.split(",") if a.only else ["blocks", "reduced", "sampler", "sam", "freeinit", "fifo", "vae", "loop"] + (["full"] if a.full else [])
    if "blocks" in todo: blocks(om, att)
    if "reduced" in todo: unet_reduced(om)
    if "sampler" in todo: sampler_cases()
    if "sam" in todo: sampler_sam_cases()
    if "freeinit" in todo: freeinit_cases()
    if "fifo" in todo: fifo_cases()
    if "vae" in todo: vae_cases()
    if "loop" in todo: loop_cases()
    if "full" in todo: unet_full(om)
    if "fullN" in todo: unet_full(om, only_cfgN=True)
    if "fullB16" in todo: unet_full_b16(om)


if __name__ == "__main__":
    main()
