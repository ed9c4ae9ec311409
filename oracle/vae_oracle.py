"""TEST INFRASTRUCTURE ONLY -- CPU fp32 restatement of the reference VAE decode path.

Follows `AutoencoderKL.decode` (lvdm/models/autoencoder.py:104-107), `Decoder.forward`
(lvdm/modules/networks/ae_modules.py:541-579), `ResnetBlock.forward` (:188-207), `AttnBlock.forward` (:51-78),
`Upsample.forward` (:117-121) and `LatentDiffusion.decode_first_stage_2DAE` (lvdm/models/ddpm3d.py:556-562) in plain
functional PyTorch, keyed by the reference state-dict names.  Pinned by tests/golden/vae_*.npz, which
tools/make_golden.py captured from the real reference modules.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this package; the product path never does.
"""
import torch
import torch.nn.functional as F


def _norm(sd, pre, x):
    return F.group_norm(x, 32, sd[pre + ".weight"], sd[pre + ".bias"], eps=1e-6)      # Normalize(), ae_modules.py:15-16


def _swish(x):
    return x * torch.sigmoid(x)                                                          # nonlinearity(), :10-12


def _conv(sd, pre, x, padding):
    return F.conv2d(x, sd[pre + ".weight"], sd[pre + ".bias"], padding=padding)


def resnet_block(sd, pre, x):
    h = _conv(sd, pre + ".conv1", _swish(_norm(sd, pre + ".norm1", x)), 1)
    h = _conv(sd, pre + ".conv2", _swish(_norm(sd, pre + ".norm2", h)), 1)               # dropout p=0, temb None
    if pre + ".nin_shortcut.weight" in sd:
        x = _conv(sd, pre + ".nin_shortcut", x, 0)
    return x + h


def attn_block(sd, pre, x):
    h = _norm(sd, pre + ".norm", x)
    q, k, v = (_conv(sd, pre + "." + n, h, 0) for n in ("q", "k", "v"))
    b, c, hh, ww = q.shape
    q = q.reshape(b, c, hh * ww).permute(0, 2, 1)
    k = k.reshape(b, c, hh * ww)
    w_ = torch.bmm(q, k) * (int(c) ** (-0.5))
    w_ = F.softmax(w_, dim=2)
    v = v.reshape(b, c, hh * ww)
    h = torch.bmm(v, w_.permute(0, 2, 1)).reshape(b, c, hh, ww)
    return x + _conv(sd, pre + ".proj_out", h, 0)


def decoder_forward(sd, z, pre="decoder", num_resolutions=4, num_res_blocks=2):
    h = _conv(sd, pre + ".conv_in", z, 1)
    h = resnet_block(sd, pre + ".mid.block_1", h)
    h = attn_block(sd, pre + ".mid.attn_1", h)
    h = resnet_block(sd, pre + ".mid.block_2", h)
    for lvl in reversed(range(num_resolutions)):
        for i in range(num_res_blocks + 1):
            h = resnet_block(sd, f"{pre}.up.{lvl}.block.{i}", h)
        if lvl != 0:
            h = F.interpolate(h, scale_factor=2.0, mode="nearest")
            h = _conv(sd, f"{pre}.up.{lvl}.upsample.conv", h, 1)
    h = _swish(_norm(sd, pre + ".norm_out", h))
    return _conv(sd, pre + ".conv_out", h, 1)


def decode(sd, z, **kw):
    """AutoencoderKL.decode: post_quant_conv then Decoder"""
    return decoder_forward(sd, _conv(sd, "post_quant_conv", z, 0), **kw)


def decode_first_stage_2DAE(sd, z, scale_factor, **kw):
    """ddpm3d.py:556-562: z [b,c,t,h,w] -> [b,3,t,H,W], one frame at a time"""
    z = 1.0 / scale_factor * z
    return torch.cat([decode(sd, z[:, :, i], **kw).unsqueeze(2) for i in range(z.shape[2])], dim=2)


def encoder_forward(sd, x, pre="encoder", num_resolutions=4, num_res_blocks=2):
    """Encoder.forward, ae_modules.py:429-464 (attn_resolutions = [], temb None)"""
    h = _conv(sd, pre + ".conv_in", x, 1)
    for lvl in range(num_resolutions):
        for i in range(num_res_blocks):
            h = resnet_block(sd, f"{pre}.down.{lvl}.block.{i}", h)
        if lvl != num_resolutions - 1:                                   # Downsample.forward, :98-103
            h = F.pad(h, (0, 1, 0, 1), mode="constant", value=0)
            h = F.conv2d(h, sd[f"{pre}.down.{lvl}.downsample.conv.weight"], sd[f"{pre}.down.{lvl}.downsample.conv.bias"], stride=2)
    h = resnet_block(sd, pre + ".mid.block_1", h)
    h = attn_block(sd, pre + ".mid.attn_1", h)
    h = resnet_block(sd, pre + ".mid.block_2", h)
    h = _swish(_norm(sd, pre + ".norm_out", h))
    return _conv(sd, pre + ".conv_out", h, 1)


def encode_moments(sd, x, **kw):
    """AutoencoderKL.encode up to the distribution parameters (autoencoder.py:98-101)"""
    return _conv(sd, "quant_conv", encoder_forward(sd, x, **kw), 0)


def sample_posterior(moments, noise=None):
    """DiagonalGaussianDistribution.sample / .mode (lvdm/distributions.py:24-40,66-67)"""
    mean, logvar = torch.chunk(moments, 2, dim=1)
    if noise is None:
        return mean
    return mean + torch.exp(0.5 * torch.clamp(logvar, -30.0, 20.0)) * noise


def encode_first_stage_2DAE(sd, x, scale_factor, noises=None, **kw):
    """ddpm3d.py:496-502: per frame encode -> sample -> * scale_factor"""
    out = []
    for i in range(x.shape[2]):
        z = sample_posterior(encode_moments(sd, x[:, :, i], **kw), None if noises is None else noises[:, :, i])
        out.append((scale_factor * z).unsqueeze(2))
    return torch.cat(out, dim=2)
