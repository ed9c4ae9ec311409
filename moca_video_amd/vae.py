"""MI355X-native `AutoencoderKL.decode`: the per-frame VAE decoder that turns every emitted
FIFO frame `[1,4,40,64]` into pixels `[1,3,320,512]` (`scripts/evaluation/funcs.py:360`,
`lvdm/models/ddpm3d.py:556-562`, `lvdm/models/autoencoder.py:104-107`, `lvdm/modules/networks/ae_modules.py:466-579`).

Same constructor arguments and `state_dict()` names/shapes as the reference `AutoencoderKL` (encoder parameters are
used by `encode`, the DAVIS-video mode's entry, `scripts/evaluation/funcs.py:47-48`; prompt mode only decodes).  Nothing of the reference's execution survives: activations are channels-last
fp16 `[frame·H·W][C]`, every conv is the implicit-GEMM kernel (the nearest-x2 upsample is folded into the following
conv's gather), GroupNorm+swish is the HBM-bound norm kernel, and the single-head 512-channel mid attention is three
GEMMs around a row-softmax kernel:

    S = Q·Kᵀ        gemm(A = q rows of the frame, W-operand = k rows of the frame)      fp32 out
    P = softmax(S / sqrt(C))                                                             fp16
    Vᵀ = W_v·Xᵀ      gemm(A = W_v, W-operand = normalised tokens)  -> [C][tokens], K-contiguous for the next product
    O = P·V + b_v   gemm(A = P, W-operand = Vᵀ, bias = b_v)       (rows of P sum to 1, so the bias moves behind P)

The launch sequence is recorded once per (frames, h, w) signature and replayed as a hipGraph, like the UNet's.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import lib as _l
from . import ops
from .plan import _PlanBase
from .unet import _FMap, _Param

__all__ = ["AutoencoderKL", "DiagonalGaussianDistribution"]


class _ResnetBlock(nn.Module):
    """ae_modules.py:150-207 with temb_channels = 0 (no temb_proj)"""

    def __init__(self, cin, cout):
        super().__init__()
        self.cin, self.cout = cin, cout
        self.norm1 = _Param((cin,), kind="norm")
        self.conv1 = _Param((cout, cin, 3, 3), kind="conv")
        self.norm2 = _Param((cout,), kind="norm")
        self.conv2 = _Param((cout, cout, 3, 3), kind="conv")
        if cin != cout:
            self.nin_shortcut = _Param((cout, cin, 1, 1), kind="conv")


class _AttnBlock(nn.Module):
    """ae_modules.py:26-78"""

    def __init__(self, c):
        super().__init__()
        self.c = c
        self.norm = _Param((c,), kind="norm")
        for n in ("q", "k", "v", "proj_out"):
            setattr(self, n, _Param((c, c, 1, 1), kind="conv"))


class _Resample(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv = _Param((c, c, 3, 3), kind="conv")


def _mid(c):
    m = nn.Module()
    m.block_1 = _ResnetBlock(c, c)
    m.attn_1 = _AttnBlock(c)
    m.block_2 = _ResnetBlock(c, c)
    return m


class _Encoder(nn.Module):
    """parameter tree of ae_modules.py:364-427 (declared for strict checkpoint loading only)"""

    def __init__(self, *, ch, ch_mult, num_res_blocks, in_channels, z_channels, double_z=True, **_):
        super().__init__()
        self.in_channels = in_channels
        self.conv_in = _Param((ch, in_channels, 3, 3), kind="conv")
        in_mult = (1,) + tuple(ch_mult)
        self.down = nn.ModuleList()
        block_in = ch
        for i, mult in enumerate(ch_mult):
            lvl = nn.Module()
            lvl.block = nn.ModuleList()
            lvl.attn = nn.ModuleList()
            block_in, block_out = ch * in_mult[i], ch * mult
            for _ in range(num_res_blocks):
                lvl.block.append(_ResnetBlock(block_in, block_out))
                block_in = block_out
            if i != len(ch_mult) - 1:
                lvl.downsample = _Resample(block_in)
            self.down.append(lvl)
        self.mid = _mid(block_in)
        self.norm_out = _Param((block_in,), kind="norm")
        self.conv_out = _Param(((2 if double_z else 1) * z_channels, block_in, 3, 3), kind="conv")


class _Decoder(nn.Module):
    """ae_modules.py:466-531"""

    def __init__(self, *, ch, out_ch, ch_mult, num_res_blocks, z_channels, attn_resolutions=(), resolution=0,
                 give_pre_end=False, tanh_out=False, **_):
        super().__init__()
        if give_pre_end or tanh_out:
            raise NotImplementedError("Decoder(give_pre_end/tanh_out) is not used by inference_t2v_512_v2.0.yaml")
        nres = len(ch_mult)
        curr_res = resolution // 2 ** (nres - 1) if resolution else 0
        block_in = ch * ch_mult[-1]
        self.conv_in = _Param((block_in, z_channels, 3, 3), kind="conv")
        self.mid = _mid(block_in)
        ups = []
        for i in reversed(range(nres)):
            lvl = nn.Module()
            lvl.block = nn.ModuleList()
            lvl.attn = nn.ModuleList()
            block_out = ch * ch_mult[i]
            for _ in range(num_res_blocks + 1):
                lvl.block.append(_ResnetBlock(block_in, block_out))
                block_in = block_out
                if curr_res and curr_res in attn_resolutions:
                    raise NotImplementedError("attn_resolutions != [] is not used by inference_t2v_512_v2.0.yaml")
            if i != 0:
                lvl.upsample = _Resample(block_in)
                curr_res *= 2
            ups.insert(0, lvl)
        self.up = nn.ModuleList(ups)
        self.norm_out = _Param((block_in,), kind="norm")
        self.conv_out = _Param((out_ch, block_in, 3, 3), kind="conv")
        self.out_ch, self.z_channels = out_ch, z_channels


class _VaePlan(_PlanBase):
    """recorded decode of `n` latent frames [n, zc, h, w]"""

    def __init__(self, model, n, h, w, in_dtype, device):
        super().__init__(model, device)
        self.n, self.h, self.w = n, h, w
        self.z_in = torch.empty(n, model.decoder.z_channels, h, w, dtype=in_dtype, device=device)
        self.out = torch.empty(n, model.decoder.out_ch, 8 * h if len(model.decoder.up) == 4 else h * 2 ** (len(model.decoder.up) - 1),
                               8 * w if len(model.decoder.up) == 4 else w * 2 ** (len(model.decoder.up) - 1),
                               dtype=torch.float32, device=device)
        self._build()

    def resnet(self, mod, x):
        """ResnetBlock.forward, ae_modules.py:188-207 (temb is None)"""
        P = self.P
        g1 = self.gn(x, P[id(mod.norm1)], fps=1, eps=1e-6, silu=True)
        h1 = self.conv(_FMap(g1, x.F, x.H, x.W, x.C), P[id(mod.conv1)])
        self._release(g1)
        g2 = self.gn(h1, P[id(mod.norm2)], fps=1, eps=1e-6, silu=True)
        self._release(h1.buf)
        sk = x.buf if mod.cin == mod.cout else self.linear(x.buf, x.M, P[id(mod.nin_shortcut)])
        h2 = self.conv(_FMap(g2, x.F, x.H, x.W, mod.cout), P[id(mod.conv2)], residual=sk)
        self._release(g2)
        if sk is not x.buf:
            self._release(sk)
        return h2

    def attn(self, mod, x):
        """AttnBlock.forward, ae_modules.py:51-78"""
        P, c, tok = self.P, mod.c, x.H * x.W
        n = self.gn(x, P[id(mod.norm)], fps=1, eps=1e-6, silu=False)
        qk = self.linear(n, x.M, P[(id(mod), "qk")])                      # [M][2c]: q | k (+bias)
        wv, bv = P[(id(mod), "wv")]
        o = self.pool.get(x.M, c)
        for f in range(x.F):
            r0 = f * tok
            kf = ops.PackedWeight(qk[r0:r0 + tok, c:2 * c], None, tok, c, tok)
            s = self.pool.get(tok, tok, torch.float32)
            self._emit(ops.gemm, qk[r0:r0 + tok, :c], kf, s, M=tok, lda=2 * c, out_f32=True)
            pr = self.pool.get(tok, tok)
            self._emit(ops.softmax_rows, s, pr, R=tok, N=tok, scale=float(c) ** -0.5)
            self.pool.put(s)
            xf = ops.PackedWeight(n[r0:r0 + tok], None, tok, c, tok)       # Vᵀ[c][tok] = W_v · Xᵀ
            vt = self.pool.get(c, tok)
            self._emit(ops.gemm, wv, xf, vt, M=c, lda=c)
            vtw = ops.PackedWeight(vt, bv, c, tok, c)                       # O = P · V + b_v
            self._emit(ops.gemm, pr, vtw, o[r0:r0 + tok], M=tok, lda=tok)
            self.pool.put(pr, vt)
        self._release(n, qk)
        out = self.linear(o, x.M, P[id(mod.proj_out)], residual=x.buf)
        self._release(o)
        return _FMap(out, x.F, x.H, x.W, x.C)

    def _build(self):
        m, d, P = self.model, self.model.decoder, self.P
        n, h, w = self.n, self.h, self.w
        z8 = self.pool.get(n * h * w, 8)
        wq, bq = P["post_quant"]
        self._emit(ops.channel_mix, self.z_in, wq, bq, z8, B=n, Cin=d.z_channels, T=1, HW=h * w, Cout=d.z_channels, Cpad=8,
                   inv_scale=1.0)
        x = self.conv(_FMap(z8, n, h, w, 8), P[id(d.conv_in)])
        self._release(z8)

        def step(fn, *a):
            nonlocal x
            nx = fn(*a)
            self._release(x.buf)
            x = nx

        step(self.resnet, d.mid.block_1, x)
        step(self.attn, d.mid.attn_1, x)
        step(self.resnet, d.mid.block_2, x)
        for i in reversed(range(len(d.up))):
            for blk in d.up[i].block:
                step(self.resnet, blk, x)
            if i != 0:
                step(lambda fm: self.conv(fm, P[id(d.up[i].upsample.conv)], up=1), x)
        g = self.gn(x, P[id(d.norm_out)], fps=1, eps=1e-6, silu=True)
        self._release(x.buf)
        o = self.conv(_FMap(g, x.F, x.H, x.W, x.C), P[id(d.conv_out)])
        self._release(g)
        self._emit(ops.nhwc_to_ncthw, o.buf, o.C, self.out, B=n, Cout=d.out_ch, T=1, HW=o.H * o.W)
        self._release(o.buf)

    def run(self, z):
        cur = torch.cuda.current_stream(self.device)
        self.stream.wait_stream(cur)
        with torch.cuda.stream(self.stream):
            self.z_in.copy_(z, non_blocking=True)
            handle = self.stream.cuda_stream
            ops.set_stream(handle)
            try:
                self._launch(handle)
            finally:
                ops.set_stream(None)
            out = self.out.clone()
        self.n_runs += 1
        out.record_stream(cur)
        cur.wait_stream(self.stream)
        return out


class _EncPlan(_PlanBase):
    """recorded encode of `n` RGB frames [n, 3, H, W] -> moments [n, 2*embed_dim, H/8, W/8] (fp32)"""

    def __init__(self, model, n, H, W, in_dtype, device):
        super().__init__(model, device)
        self.n, self.H, self.W = n, H, W
        e = model.encoder
        down = 2 ** (len(e.down) - 1)
        self.x_in = torch.empty(n, e.in_channels, H, W, dtype=in_dtype, device=device)
        self.out = torch.empty(n, 2 * model.embed_dim, H // down, W // down, dtype=torch.float32, device=device)
        self._build()

    resnet = _VaePlan.resnet
    attn = _VaePlan.attn

    def _build(self):
        m, e, P = self.model, self.model.encoder, self.P
        n, H, W = self.n, self.H, self.W
        x8 = self.pool.get(n * H * W, 8)
        self._emit(ops.ncthw_to_nhwc, self.x_in, x8, B=n, Cin=e.in_channels, T=1, HW=H * W, Cpad=8)
        x = self.conv(_FMap(x8, n, H, W, 8), P[id(e.conv_in)])
        self._release(x8)

        def step(fn, *a):
            nonlocal x
            nx = fn(*a)
            self._release(x.buf)
            x = nx

        for i, lvl in enumerate(e.down):                                   # Encoder.forward, ae_modules.py:429-464
            for blk in lvl.block:
                step(self.resnet, blk, x)
            if i != len(e.down) - 1:
                step(self.down, lvl.downsample, x)
        step(self.resnet, e.mid.block_1, x)
        step(self.attn, e.mid.attn_1, x)
        step(self.resnet, e.mid.block_2, x)
        g = self.gn(x, P[id(e.norm_out)], fps=1, eps=1e-6, silu=True)
        self._release(x.buf)
        o = self.conv(_FMap(g, x.F, x.H, x.W, x.C), P[id(e.conv_out)])   # [M][64]: 2*z_channels real columns
        self._release(g)
        q = self.linear(o.buf, o.M, P["quant"])                           # quant_conv 1x1 (autoencoder.py:100)
        self._release(o.buf)
        self._emit(ops.nhwc_to_ncthw, q, q.shape[1], self.out, B=n, Cout=2 * m.embed_dim, T=1, HW=o.H * o.W)
        self._release(q)

    def down(self, mod, x):
        """Downsample.forward with_conv (ae_modules.py:98-103): F.pad(x, (0,1,0,1)) then a stride-2 3x3 conv without padding"""
        if x.H % 2 or x.W % 2:
            raise NotImplementedError("AutoencoderKL.encode needs H and W divisible by 8")
        M = x.F * (x.H // 2) * (x.W // 2)
        pw = self.P[id(mod.conv)]
        out = self._gemm(x.buf, pw, M, mode=_l.MOCA_A_CONV3X3, conv=(x.C, x.H, x.W, x.H // 2, x.W // 2, 2, 0, 1))
        return _FMap(out, x.F, x.H // 2, x.W // 2, pw.N)

    def run(self, x):
        cur = torch.cuda.current_stream(self.device)
        self.stream.wait_stream(cur)
        with torch.cuda.stream(self.stream):
            self.x_in.copy_(x, non_blocking=True)
            handle = self.stream.cuda_stream
            ops.set_stream(handle)
            try:
                self._launch(handle)
            finally:
                ops.set_stream(None)
            out = self.out.clone()
        self.n_runs += 1
        out.record_stream(cur)
        cur.wait_stream(self.stream)
        return out


class DiagonalGaussianDistribution:
    """lvdm/distributions.py:24-67 over device moments [n, 2z, h, w] (fp32): `.sample(noise=None)`, `.mode()`, `.mean`,
    `.logvar`, `.std`, `.var`; the arithmetic of sample/mode is one HIP kernel (`moca_gaussian_sample_f32`)."""

    def __init__(self, parameters, deterministic=False):
        self.parameters = parameters
        self.deterministic = deterministic
        self.mean, self._logvar_raw = torch.chunk(parameters, 2, dim=1)

    @property
    def logvar(self):
        return torch.clamp(self._logvar_raw, -30.0, 20.0)

    @property
    def std(self):
        return torch.zeros_like(self.mean) if self.deterministic else torch.exp(0.5 * self.logvar)

    @property
    def var(self):
        return torch.zeros_like(self.mean) if self.deterministic else torch.exp(self.logvar)

    def _draw(self, noise, scale=1.0):
        n, z2, h, w = self.parameters.shape
        out = torch.empty(n, z2 // 2, h, w, dtype=torch.float32, device=self.parameters.device)
        if noise is not None:
            noise = noise.to(device=out.device, dtype=torch.float32).contiguous()
        ops.set_stream(None)
        ops.gaussian_sample(self.parameters.contiguous(), noise, out, n=n, z=z2 // 2, hw=h * w, scale=scale)
        return out

    def sample(self, noise=None):
        if self.deterministic:
            return self._draw(None)
        if noise is None:
            noise = torch.randn(self.mean.shape, device=self.parameters.device)
        return self._draw(noise)

    def mode(self):
        return self._draw(None)


class AutoencoderKL(nn.Module):
    """drop-in for `lvdm.models.autoencoder.AutoencoderKL` (autoencoder.py:13-50) on the decode path"""

    max_frames_per_launch = 8

    def __init__(self, ddconfig, lossconfig=None, embed_dim=4, ckpt_path=None, ignore_keys=(), image_key="image",
                 colorize_nlabels=None, monitor=None, test=False, logdir=None, input_dim=4, test_args=None, use_graph=True):
        super().__init__()
        assert ddconfig["double_z"]
        if ckpt_path is not None:
            raise NotImplementedError("AutoencoderKL(ckpt_path=...): load the state dict through load_state_dict")
        self.image_key, self.embed_dim, self.input_dim = image_key, embed_dim, input_dim
        self.encoder = _Encoder(**ddconfig)
        self.decoder = _Decoder(**ddconfig)
        self.quant_conv = _Param((2 * embed_dim, 2 * ddconfig["z_channels"], 1, 1), kind="conv")
        self.post_quant_conv = _Param((ddconfig["z_channels"], embed_dim, 1, 1), kind="conv")
        self.use_graph = use_graph
        self._packed, self._plans = None, {}
        self.register_load_state_dict_post_hook(lambda module, incompatible: module._invalidate())

    def _invalidate(self):
        self._packed, self._plans = None, {}

    def _apply(self, fn, recurse=True):
        out = super()._apply(fn, recurse)
        self._invalidate()
        return out

    def _pack(self):
        dev = next(self.parameters()).device
        if dev.type != "cuda":
            raise RuntimeError("moca_video_amd.AutoencoderKL runs on an MI355X only; call .cuda() first (no CPU path)")
        P, d = {}, self.decoder
        if any(m.cout % 64 or m.cin % 64 for m in self.modules() if isinstance(m, _ResnetBlock)):
            raise NotImplementedError("decoder channel counts must be multiples of 64 (ch=128 in inference_t2v_512_v2.0.yaml)")
        f32 = lambda t: t.detach().to(device=dev, dtype=torch.float32).contiguous()
        conv = lambda m, **kw: ops.pack_conv3x3(m.weight.detach(), m.bias.detach(), device=dev, **kw)
        for mod in list(d.modules()) + list(self.encoder.modules()):
            if isinstance(mod, _ResnetBlock):
                P[id(mod.norm1)] = (f32(mod.norm1.weight), f32(mod.norm1.bias))
                P[id(mod.norm2)] = (f32(mod.norm2.weight), f32(mod.norm2.bias))
                P[id(mod.conv1)], P[id(mod.conv2)] = conv(mod.conv1), conv(mod.conv2)
                if mod.cin != mod.cout:
                    P[id(mod.nin_shortcut)] = ops.pack_conv1x1(mod.nin_shortcut.weight.detach(), mod.nin_shortcut.bias.detach(), device=dev)
            elif isinstance(mod, _AttnBlock):
                if mod.c % 64:
                    raise NotImplementedError("AttnBlock channels must be a multiple of 64")
                P[id(mod.norm)] = (f32(mod.norm.weight), f32(mod.norm.bias))
                P[(id(mod), "qk")] = ops.pack_linear_cat([mod.q.weight.detach(), mod.k.weight.detach()],
                                                         [mod.q.bias.detach(), mod.k.bias.detach()], device=dev)
                P[(id(mod), "wv")] = (mod.v.weight.detach().reshape(mod.c, mod.c).to(device=dev, dtype=torch.float16).contiguous(),
                                      f32(mod.v.bias))
                P[id(mod.proj_out)] = ops.pack_conv1x1(mod.proj_out.weight.detach(), mod.proj_out.bias.detach(), device=dev)
            elif isinstance(mod, _Resample):
                P[id(mod.conv)] = conv(mod.conv)
        P[id(d.conv_in)] = conv(d.conv_in, cpad=8)
        P[id(d.norm_out)] = (f32(d.norm_out.weight), f32(d.norm_out.bias))
        P[id(d.conv_out)] = conv(d.conv_out)
        pq = self.post_quant_conv
        P["post_quant"] = (f32(pq.weight).reshape(pq.weight.shape[0], -1).contiguous(), f32(pq.bias))
        e = self.encoder
        P[id(e.conv_in)] = conv(e.conv_in, cpad=8)
        P[id(e.norm_out)] = (f32(e.norm_out.weight), f32(e.norm_out.bias))
        P[id(e.conv_out)] = conv(e.conv_out)
        P["quant"] = ops.pack_conv1x1(self.quant_conv.weight.detach(), self.quant_conv.bias.detach(), device=dev)
        self._packed = P

    # ---- reference API -------------------------------------------------------------------
    @torch.no_grad()
    def encode(self, x, **kwargs):
        """autoencoder.py:98-102: x [n, 3, H, W] in [-1, 1] -> DiagonalGaussianDistribution over [n, 2*embed_dim, H/8, W/8]
        (used by the DAVIS-video mode, funcs.py:47-48; prompt mode never encodes)"""
        if x.dim() != 4 or x.shape[1] != self.encoder.in_channels:
            raise ValueError(f"encode expects [n,{self.encoder.in_channels},H,W], got {tuple(x.shape)}")
        if not x.is_cuda:
            raise ValueError("moca_video_amd.AutoencoderKL.encode needs a CUDA (HIP) tensor; there is no CPU path")
        down = 2 ** (len(self.encoder.down) - 1)
        if x.shape[2] % down or x.shape[3] % down or ((x.shape[2] // down) * (x.shape[3] // down)) % 64:
            raise NotImplementedError("H, W must be multiples of 8 and (H/8)*(W/8) a multiple of 64")
        if self._packed is None:
            self._pack()
        outs = []
        for i in range(0, x.shape[0], self.max_frames_per_launch):
            xi = x[i:i + self.max_frames_per_launch].contiguous()
            key = ("enc", xi.shape[0], xi.shape[2], xi.shape[3], xi.dtype)
            plan = self._plans.get(key)
            if plan is None:
                plan = self._plans[key] = _EncPlan(self, xi.shape[0], xi.shape[2], xi.shape[3], xi.dtype, xi.device)
            outs.append(plan.run(xi))
        return DiagonalGaussianDistribution(outs[0] if len(outs) == 1 else torch.cat(outs, 0))

    @torch.no_grad()
    def decode(self, z, **kwargs):
        """z [n, z_channels, h, w] (already divided by scale_factor, ddpm3d.py:559) -> [n, 3, 8h, 8w] fp32"""
        if z.dim() != 4 or z.shape[1] != self.decoder.z_channels:
            raise ValueError(f"decode expects [n,{self.decoder.z_channels},h,w], got {tuple(z.shape)}")
        if not z.is_cuda:
            raise ValueError("moca_video_amd.AutoencoderKL.decode needs a CUDA (HIP) tensor; there is no CPU path")
        if (z.shape[2] * z.shape[3]) % 64:
            raise NotImplementedError("h*w must be a multiple of 64 (mid-attention logits are one GEMM per frame)")
        if self._packed is None:
            self._pack()
        outs = []
        for i in range(0, z.shape[0], self.max_frames_per_launch):
            zi = z[i:i + self.max_frames_per_launch].contiguous()
            key = (zi.shape[0], zi.shape[2], zi.shape[3], zi.dtype)
            plan = self._plans.get(key)
            if plan is None:
                plan = self._plans[key] = _VaePlan(self, zi.shape[0], zi.shape[2], zi.shape[3], zi.dtype, zi.device)
            outs.append(plan.run(zi))
        return outs[0] if len(outs) == 1 else torch.cat(outs, 0)

    def forward(self, input, sample_posterior=True):
        raise NotImplementedError("AutoencoderKL.forward (encode + decode) is a training path")
