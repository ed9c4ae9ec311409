"""Drop-in boundary above the UNet: `DiffusionWrapper.forward` (lvdm/models/ddpm3d.py:702-763,
`crossattn` branch :710-712) and the slice of `LatentDiffusion` the samplers touch:
`apply_model` (:512-527), the DDPM schedule buffers (`register_schedule`, :113-165) and
`scale_arr` (:362-376).  VAE / text encoder / training scaffolding are out of scope."""
from __future__ import annotations

import importlib
from functools import partial

import numpy as np
import torch
import torch.nn as nn

from .unet import UNetModel

# `target:` strings of the reference YAML resolve to our classes, so an unmodified
# configs/inference_t2v_512_v2.0.yaml instantiates the MI355X path.
def _vae_cls(**kw):
    from .vae import AutoencoderKL
    return AutoencoderKL(**kw)


_TARGET_ALIASES = {
    "lvdm.modules.networks.openaimodel3d.UNetModel": UNetModel,
    "lvdm.models.autoencoder.AutoencoderKL": _vae_cls,
    "lvdm.modules.encoders.condition.FrozenOpenCLIPEmbedder": lambda **kw: _clip_cls(**kw),
}


def _clip_cls(**kw):
    from .clip_text import FrozenOpenCLIPEmbedder
    return FrozenOpenCLIPEmbedder(**kw)


def get_obj_from_str(string):
    if string in _TARGET_ALIASES:
        return _TARGET_ALIASES[string]
    module, cls = string.rsplit(".", 1)
    return getattr(importlib.import_module(module, package=None), cls)


def instantiate_from_config(config):
    """utils/utils.py:27-34"""
    if "target" not in config:
        if config == '__is_first_stage__':
            return None
        elif config == "__is_unconditional__":
            return None
        raise KeyError("Expected key `target` to instantiate.")
    return get_obj_from_str(config["target"])(**config.get("params", dict()))


def load_unet_config(yaml_path):
    """unet_config block of configs/inference_t2v_512_v2.0.yaml (:22-50) as a plain dict."""
    import yaml
    with open(yaml_path) as f:
        cfg = yaml.safe_load(f)
    return cfg["model"]["params"]["unet_config"], cfg["model"]["params"]


class DiffusionWrapper(nn.Module):
    """lvdm/models/ddpm3d.py:696-763 -- only the conditioning modes reachable from the YAML
    (`crossattn`; `None` kept for completeness) are implemented, the rest raise like upstream."""

    def __init__(self, diff_model_config, conditioning_key):
        super().__init__()
        self.diffusion_model = instantiate_from_config(diff_model_config)
        self.conditioning_key = conditioning_key

    def forward(self, x, t, c_concat: list = None, c_crossattn: list = None, c_adm=None, s=None, mask=None, **kwargs):
        if self.conditioning_key == 'crossattn':
            cc = torch.cat(c_crossattn, 1)                                   # :711
            out = self.diffusion_model(x, t, context=cc, **kwargs)           # :712
        else:
            raise NotImplementedError(f"conditioning_key={self.conditioning_key!r} is outside the MoCA hot path")
        return out


def make_beta_schedule(schedule, n_timestep, linear_start=1e-4, linear_end=2e-2):
    """utils_diffusion.py:31-53 (the 'linear' branch the YAML uses)"""
    if schedule != "linear":
        raise NotImplementedError(schedule)
    betas = torch.linspace(linear_start ** 0.5, linear_end ** 0.5, n_timestep, dtype=torch.float64, device="cpu") ** 2
    return betas.numpy()


class DenoiseModel(nn.Module):
    """What `DDIMSampler` needs from `LatentDiffusion`: `apply_model`, `num_timesteps`, `betas`,
    `alphas_cumprod(_prev)`, `use_scale`/`scale_arr`, `device` (ddpm3d.py:83-165,362-376,512-527)."""

    def __init__(self, unet_config, timesteps=1000, linear_start=0.00085, linear_end=0.012, conditioning_key="crossattn",
                 use_scale=True, scale_a=1, scale_b=0.7, mid_step=400, fix_scale_bug=False, parameterization="eps",
                 uncond_type="empty_seq", first_stage_config=None, scale_factor=1.0, cond_stage_config=None, **ignored):
        super().__init__()
        self.parameterization = parameterization
        self.uncond_type = uncond_type
        self.model = DiffusionWrapper(unet_config, conditioning_key)
        self.num_timesteps = int(timesteps)
        # register_schedule, ddpm3d.py:113-165
        betas = make_beta_schedule("linear", timesteps, linear_start=linear_start, linear_end=linear_end)
        alphas = 1. - betas
        alphas_cumprod = np.cumprod(alphas, axis=0)
        alphas_cumprod_prev = np.append(1., alphas_cumprod[:-1])
        to_torch = partial(torch.tensor, dtype=torch.float32)
        self.register_buffer('betas', to_torch(betas))
        self.register_buffer('alphas_cumprod', to_torch(alphas_cumprod))
        self.register_buffer('alphas_cumprod_prev', to_torch(alphas_cumprod_prev))
        self.register_buffer('sqrt_alphas_cumprod', to_torch(np.sqrt(alphas_cumprod)))
        self.register_buffer('sqrt_one_minus_alphas_cumprod', to_torch(np.sqrt(1. - alphas_cumprod)))
        # scale_arr, ddpm3d.py:362-376 (the "bug" branch: length mid_step + num_timesteps = 1400)
        self.use_scale = use_scale
        if use_scale:
            scale_step = self.num_timesteps - mid_step if fix_scale_bug else self.num_timesteps
            scale_arr = np.concatenate((np.linspace(scale_a, scale_b, mid_step), np.full(scale_step, scale_b)))
            self.register_buffer('scale_arr', to_torch(scale_arr))
        # first stage (ddpm3d.py:383,386,431-437): only the decode side is on the MoCA path (funcs.py:360)
        self.scale_factor = scale_factor
        self.first_stage_model = instantiate_from_config(first_stage_config) if first_stage_config is not None else None
        self.cond_stage_model = instantiate_from_config(cond_stage_config) if cond_stage_config is not None else None

    def get_learned_conditioning(self, c):
        """ddpm3d.py:440-455 (cond_stage_forward is None: `self.cond_stage_model.encode(c)`); `c` = token ids [B, 77] -- the BPE
        tokenizer of open_clip is the caller's (not available offline)"""
        if self.cond_stage_model is None:
            raise RuntimeError("DenoiseModel was built without cond_stage_config")
        return self.cond_stage_model.encode(c)

    @torch.no_grad()
    def decode_first_stage_2DAE(self, z, **kwargs):
        """ddpm3d.py:556-562: z [b,c,t,h,w] latents -> [b,3,t,8h,8w]; the reference decodes one frame per call, here
        all b*t frames go through the recorded decoder in groups of `AutoencoderKL.max_frames_per_launch`."""
        if self.first_stage_model is None:
            raise RuntimeError("DenoiseModel was built without first_stage_config")
        b, c, t, h, w = z.shape
        zf = (1. / self.scale_factor * z).permute(0, 2, 1, 3, 4).reshape(b * t, c, h, w)
        out = self.first_stage_model.decode(zf, **kwargs)
        return out.reshape(b, t, *out.shape[1:]).permute(0, 2, 1, 3, 4).contiguous()

    def get_first_stage_encoding(self, encoder_posterior, noise=None):
        """ddpm3d.py:458-465"""
        from .vae import DiagonalGaussianDistribution
        if isinstance(encoder_posterior, DiagonalGaussianDistribution):
            z = encoder_posterior.sample(noise=noise)
        elif isinstance(encoder_posterior, torch.Tensor):
            z = encoder_posterior
        else:
            raise NotImplementedError(f"encoder_posterior of type '{type(encoder_posterior)}' not yet implemented")
        return self.scale_factor * z

    @torch.no_grad()
    def encode_first_stage_2DAE(self, x, noise=None):
        """ddpm3d.py:496-502: x [b,3,t,H,W] -> latents [b,4,t,H/8,W/8]; the reference encodes (and samples) frame by
        frame, here all frames go through the recorded encoder together (`noise` [b,4,t,h,w] optionally fixes the draw)."""
        if self.first_stage_model is None:
            raise RuntimeError("DenoiseModel was built without first_stage_config")
        b, c, t, H, W = x.shape
        post = self.first_stage_model.encode(x.permute(0, 2, 1, 3, 4).reshape(b * t, c, H, W))
        if noise is not None:
            noise = noise.permute(0, 2, 1, 3, 4).reshape(b * t, *noise.shape[1:2], *noise.shape[3:])
        z = self.get_first_stage_encoding(post, noise=noise)
        return z.reshape(b, t, *z.shape[1:]).permute(0, 2, 1, 3, 4).contiguous()

    @property
    def device(self):
        return self.betas.device

    def apply_model(self, x_noisy, t, cond, **kwargs):
        """ddpm3d.py:512-527"""
        if isinstance(cond, dict):
            pass
        else:
            if not isinstance(cond, list):
                cond = [cond]
            key = 'c_concat' if self.model.conditioning_key == 'concat' else 'c_crossattn'
            cond = {key: cond}
        x_recon = self.model(x_noisy, t, **cond, **kwargs)
        if isinstance(x_recon, tuple):
            return x_recon[0]
        return x_recon
