"""ORACLE (test infrastructure, never shipped, never on the product path).

CPU fp32 restatement, in plain functional PyTorch, of the reference VideoCrafter2 3D-UNet
forward that MoCA-Video drives (`/root/reference/lvdm/modules/networks/openaimodel3d.py`,
`lvdm/modules/attention.py`, `lvdm/basics.py`, `lvdm/models/utils_diffusion.py`).  It is
keyed by the REFERENCE state-dict names and follows the reference's own NCHW / einsum
formulation (XFORMERS_IS_AVAILBLE=False path), i.e. it shares no code and no data
layout with the HIP implementation it checks.

Pinned: tests/golden/unet_*.npz hold outputs of the REAL reference modules (imported from
/root/reference in the build container by tools/make_golden.py) on seeded inputs and
weightgen parameters; tests/test_oracle_golden.py checks this file against them.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


# ---- utils_diffusion.py:8-28 ---------------------------------------------------------
def timestep_embedding(timesteps, dim, max_period=10000):
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(start=0, end=half, dtype=torch.float32) / half)
    args = timesteps[:, None].float() * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb


def _lin(sd, p, x):
    return F.linear(x, sd[p + ".weight"], sd.get(p + ".bias"))


def _gn(sd, p, x, eps):
    # GroupNormSpecific (basics.py:76-78) upcasts to fp32; everything is fp32 here already
    return F.group_norm(x, 32, sd[p + ".weight"], sd[p + ".bias"], eps)


# ---- openaimodel3d.py:237-276 --------------------------------------------------------
def temporal_conv_block(sd, p, x):
    """x: [b, c, t, h, w]"""
    identity = x
    for name, idx in (("conv1", 2), ("conv2", 3), ("conv3", 3), ("conv4", 3)):
        x = F.silu(_gn(sd, f"{p}.{name}.0", x, 1e-5))
        x = F.conv3d(x, sd[f"{p}.{name}.{idx}.weight"], sd[f"{p}.{name}.{idx}.bias"], padding=(1, 0, 0))
    return x + identity


# ---- openaimodel3d.py:208-234 --------------------------------------------------------
def res_block(sd, p, x, emb, batch_size):
    """x: [(b t), c, h, w]; emb: [(b t), emb_ch]"""
    h = F.silu(_gn(sd, p + ".in_layers.0", x, 1e-5))
    h = F.conv2d(h, sd[p + ".in_layers.2.weight"], sd[p + ".in_layers.2.bias"], padding=1)
    emb_out = _lin(sd, p + ".emb_layers.1", F.silu(emb))
    h = h + emb_out[:, :, None, None]
    h = F.silu(_gn(sd, p + ".out_layers.0", h, 1e-5))
    h = F.conv2d(h, sd[p + ".out_layers.3.weight"], sd[p + ".out_layers.3.bias"], padding=1)
    if p + ".skip_connection.weight" in sd:
        x = F.conv2d(x, sd[p + ".skip_connection.weight"], sd[p + ".skip_connection.bias"])
    h = x + h
    if p + ".temopral_conv.conv1.0.weight" in sd and batch_size:
        bt, c, hh, ww = h.shape
        h5 = h.view(batch_size, bt // batch_size, c, hh, ww).permute(0, 2, 1, 3, 4)
        h5 = temporal_conv_block(sd, p + ".temopral_conv", h5)
        h = h5.permute(0, 2, 1, 3, 4).reshape(bt, c, hh, ww)
    return h


# ---- attention.py:76-127 -------------------------------------------------------------
def cross_attention(sd, p, x, context, heads):
    q = F.linear(x, sd[p + ".to_q.weight"])
    context = x if context is None else context
    k = F.linear(context, sd[p + ".to_k.weight"])
    v = F.linear(context, sd[p + ".to_v.weight"])
    b, n, _ = q.shape
    d = q.shape[-1] // heads

    def split(t):
        return t.view(t.shape[0], t.shape[1], heads, d).permute(0, 2, 1, 3).reshape(t.shape[0] * heads, t.shape[1], d)

    q, k, v = split(q), split(k), split(v)
    sim = torch.einsum('b i d, b j d -> b i j', q, k) * (d ** -0.5)
    sim = sim.softmax(dim=-1)
    out = torch.einsum('b i j, b j d -> b i d', sim, v)
    out = out.view(b, heads, n, d).permute(0, 2, 1, 3).reshape(b, n, heads * d)
    return _lin(sd, p + ".to_out.0", out)


# ---- attention.py:216-220, 376-403 ---------------------------------------------------
def basic_transformer_block(sd, p, x, context, heads):
    def ln(name, t):
        return F.layer_norm(t, (t.shape[-1],), sd[f"{p}.{name}.weight"], sd[f"{p}.{name}.bias"], 1e-5)

    x = cross_attention(sd, p + ".attn1", ln("norm1", x), None, heads) + x
    x = cross_attention(sd, p + ".attn2", ln("norm2", x), context, heads) + x
    y = _lin(sd, p + ".ff.net.0.proj", ln("norm3", x))
    a, gate = y.chunk(2, dim=-1)
    y = a * F.gelu(gate)
    x = _lin(sd, p + ".ff.net.2", y) + x
    return x


def _n_blocks(sd, p):
    n = 0
    while f"{p}.transformer_blocks.{n}.norm1.weight" in sd:
        n += 1
    return n


# ---- attention.py:262-278 ------------------------------------------------------------
def spatial_transformer(sd, p, x, context, heads):
    b, c, h, w = x.shape
    x_in = x
    x = _gn(sd, p + ".norm", x, 1e-6)
    use_linear = sd[p + ".proj_in.weight"].dim() == 2
    if not use_linear:
        x = F.conv2d(x, sd[p + ".proj_in.weight"], sd[p + ".proj_in.bias"])
    x = x.permute(0, 2, 3, 1).reshape(b, h * w, x.shape[1])
    if use_linear:
        x = _lin(sd, p + ".proj_in", x)
    for i in range(_n_blocks(sd, p)):
        x = basic_transformer_block(sd, f"{p}.transformer_blocks.{i}", x, context, heads)
    if use_linear:
        x = _lin(sd, p + ".proj_out", x)
    x = x.view(b, h, w, x.shape[-1]).permute(0, 3, 1, 2)
    if not use_linear:
        x = F.conv2d(x, sd[p + ".proj_out.weight"], sd[p + ".proj_out.bias"])
    return x + x_in


# ---- attention.py:331-373 (only_self_att=True, no causal mask, no relative position) ----
def temporal_transformer(sd, p, x, heads):
    """x: [b, c, t, h, w]"""
    b, c, t, h, w = x.shape
    x_in = x
    x = _gn(sd, p + ".norm", x, 1e-6)
    x = x.permute(0, 3, 4, 1, 2).reshape(b * h * w, c, t)
    use_linear = sd[p + ".proj_in.weight"].dim() == 2
    if not use_linear:
        x = F.conv1d(x, sd[p + ".proj_in.weight"], sd[p + ".proj_in.bias"])
    x = x.permute(0, 2, 1)
    if use_linear:
        x = _lin(sd, p + ".proj_in", x)
    for i in range(_n_blocks(sd, p)):
        x = basic_transformer_block(sd, f"{p}.transformer_blocks.{i}", x, None, heads)
    if use_linear:
        x = _lin(sd, p + ".proj_out", x)
        x = x.view(b, h, w, t, c).permute(0, 4, 3, 1, 2)
    else:
        x = x.permute(0, 2, 1)
        x = F.conv1d(x, sd[p + ".proj_out.weight"], sd[p + ".proj_out.bias"])
        x = x.view(b, h, w, c, t).permute(0, 3, 4, 1, 2)
    return x + x_in


def _run_sequential(sd, p, h, emb, context, b, head_ch):
    """TimestepEmbedSequential.forward (openaimodel3d.py:36-48).  The layer kinds are recovered
    from which keys exist; inside one block the reference always appends the SpatialTransformer
    before the TemporalTransformer (openaimodel3d.py:410-425,456-470,498-513)."""
    i, n_tr = 0, 0
    while True:
        q = f"{p}.{i}"
        if q + ".in_layers.0.weight" in sd:
            h = res_block(sd, q, h, emb, b)
            n_tr = 0
        elif q + ".transformer_blocks.0.norm1.weight" in sd:
            heads = sd[q + ".transformer_blocks.0.attn1.to_q.weight"].shape[0] // head_ch
            if n_tr == 0:
                h = spatial_transformer(sd, q, h, context, heads)
            else:
                bt, c, hh, ww = h.shape
                h5 = h.view(b, bt // b, c, hh, ww).permute(0, 2, 1, 3, 4)
                h5 = temporal_transformer(sd, q, h5, heads)
                h = h5.permute(0, 2, 1, 3, 4).reshape(bt, c, hh, ww)
            n_tr += 1
        elif q + ".op.weight" in sd:        # Downsample, :51-77
            h = F.conv2d(h, sd[q + ".op.weight"], sd[q + ".op.bias"], stride=2, padding=1)
        elif q + ".conv.weight" in sd:      # Upsample, :80-106
            h = F.interpolate(h, scale_factor=2, mode='nearest')
            h = F.conv2d(h, sd[q + ".conv.weight"], sd[q + ".conv.bias"], padding=1)
        elif q + ".weight" in sd:           # plain conv (input_blocks.0.0)
            h = F.conv2d(h, sd[q + ".weight"], sd[q + ".bias"], padding=1)
        else:
            break
        i += 1
    return h


# ---- openaimodel3d.py:534-578 --------------------------------------------------------
def unet_forward(sd, x, timesteps, context, fps=16, head_ch=64, model_channels=None):
    """sd: reference state dict (fp32 CPU tensors); x [b,c,t,h,w]; timesteps int64 [b] or [t]."""
    model_channels = model_channels or sd["time_embed.0.weight"].shape[1]
    is_fifo = x.shape[0] != timesteps.shape[0]
    emb = _lin(sd, "time_embed.2", F.silu(_lin(sd, "time_embed.0", timestep_embedding(timesteps, model_channels))))
    if "fps_embedding.0.weight" in sd:
        if type(fps) == int:
            fps = torch.full_like(timesteps, fps)
        fe = timestep_embedding(fps, model_channels)
        emb = emb + _lin(sd, "fps_embedding.2", F.silu(_lin(sd, "fps_embedding.0", fe)))
    b, _, t, _, _ = x.shape
    context = context.repeat_interleave(repeats=t, dim=0)
    if not is_fifo:
        emb = emb.repeat_interleave(repeats=t, dim=0)
    h = x.permute(0, 2, 1, 3, 4).reshape(b * t, x.shape[1], x.shape[3], x.shape[4]).float()
    hs = []
    i = 0
    while f"input_blocks.{i}.0.weight" in sd or f"input_blocks.{i}.0.in_layers.0.weight" in sd or f"input_blocks.{i}.0.op.weight" in sd:
        h = _run_sequential(sd, f"input_blocks.{i}", h, emb, context, b, head_ch)
        if i == 0 and "init_attn.0.norm.weight" in sd:
            heads = sd["init_attn.0.transformer_blocks.0.attn1.to_q.weight"].shape[0] // head_ch
            bt, c, hh, ww = h.shape
            h5 = h.view(b, t, c, hh, ww).permute(0, 2, 1, 3, 4)
            h5 = temporal_transformer(sd, "init_attn.0", h5, heads)
            h = h5.permute(0, 2, 1, 3, 4).reshape(bt, c, hh, ww)
        hs.append(h)
        i += 1
    h = _run_sequential(sd, "middle_block", h, emb, context, b, head_ch)
    i = 0
    while f"output_blocks.{i}.0.in_layers.0.weight" in sd:
        h = torch.cat([h, hs.pop()], dim=1)
        h = _run_sequential(sd, f"output_blocks.{i}", h, emb, context, b, head_ch)
        i += 1
    y = F.silu(_gn(sd, "out.0", h, 1e-5))
    y = F.conv2d(y, sd["out.2.weight"], sd["out.2.bias"], padding=1)
    return y.view(b, t, y.shape[1], y.shape[2], y.shape[3]).permute(0, 2, 1, 3, 4)
