"""moca_video_amd -- MI355X (gfx950) native denoising hot path of MoCA-Video / VideoCrafter2.

Public surface mirrors the reference interfaces for this path only:
  UNetModel            lvdm/modules/networks/openaimodel3d.py:279-578
  DiffusionWrapper     lvdm/models/ddpm3d.py:696-763
  DenoiseModel         the slice of LatentDiffusion the samplers use (apply_model + schedule buffers)
  DDIMSampler          lvdm/models/samplers/ddim.py (make_schedule, p_sample_ddim, unet, ddim_step, fifo_onestep)
  freq_mix_3d, get_freq_filter   utils/freeinit_utils.py
  prepare_latents, shift_latents, fifo_ddim_sampling, base_ddim_sampling   scripts/evaluation/funcs.py
  instantiate_from_config       utils/utils.py:27-42
  AutoencoderKL                 lvdm/models/autoencoder.py:13-107 + lvdm/modules/networks/ae_modules.py:364-579
  FrozenOpenCLIPEmbedder        lvdm/modules/encoders/condition.py:174-235 (text tower on token ids)
  SimpleTokenizer               what open_clip.tokenize does (condition.py:207): byte-level BPE, needs the CLIP merges file
Importing the package loads libmoca_hip.so and fails loudly if it has not been built.
"""
from . import lib as _lib

_lib.load()

from .unet import UNetModel  # noqa: E402
from .wrapper import DiffusionWrapper, DenoiseModel, instantiate_from_config, load_unet_config  # noqa: E402
from .vae import AutoencoderKL  # noqa: E402
from .clip_text import FrozenOpenCLIPEmbedder  # noqa: E402
from .tokenizer import SimpleTokenizer  # noqa: E402

__all__ = ["UNetModel", "DiffusionWrapper", "DenoiseModel", "AutoencoderKL", "FrozenOpenCLIPEmbedder", "SimpleTokenizer", "instantiate_from_config", "load_unet_config"]
