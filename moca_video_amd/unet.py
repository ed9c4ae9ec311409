"""MI355X-native drop-in for `lvdm.modules.networks.openaimodel3d.UNetModel`.

Same constructor kwargs as `configs/inference_t2v_512_v2.0.yaml: unet_config.params`
(reference ctor: openaimodel3d.py:307-337), same `state_dict()` keys/shapes as the
reference module tree (incl. the `temopral_conv` spelling, openaimodel3d.py:188) and the
same `forward(x, timesteps, context=None, features_adapter=None, fps=16, **kwargs)`
(openaimodel3d.py:534).  Swapping it in = changing the YAML `target:` string.

Internally nothing of the reference's execution model survives: parameters are
re-packed once into K-contiguous fp16 GEMM operands, activations live channels-last
([frame][y][x][c] fp16), every contraction is the hand-written implicit-GEMM kernel,
and the ~1000 launches of a forward are recorded into a hipGraph and replayed.
There is no PyTorch compute fallback: without libmoca_hip.so the import fails.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import lib as _l
from . import ops

__all__ = ["UNetModel"]


# --------------------------------------------------------------------------------------
# parameter containers: reproduce the reference module tree / state_dict names only
# --------------------------------------------------------------------------------------
class _Param(nn.Module):
    """weight (+bias) holder standing in for nn.Conv2d / nn.Conv3d / nn.Linear / norms."""

    def __init__(self, wshape, bias=True, kind="linear"):
        super().__init__()
        self.kind = kind
        self.weight = nn.Parameter(torch.empty(*wshape), requires_grad=False)
        if bias:
            self.bias = nn.Parameter(torch.empty(wshape[0]), requires_grad=False)
        else:
            self.register_parameter("bias", None)


def _seq(*mods):
    return nn.Sequential(*mods)


class _ResBlock(nn.Module):
    """openaimodel3d.py:124-193"""

    def __init__(self, cin, emb_ch, cout, temporal_conv):
        super().__init__()
        self.cin, self.cout = cin, cout
        self.in_layers = _seq(_Param((cin,), kind="norm"), nn.Identity(), _Param((cout, cin, 3, 3), kind="conv"))
        self.emb_layers = _seq(nn.Identity(), _Param((cout, emb_ch)))
        self.out_layers = _seq(_Param((cout,), kind="norm"), nn.Identity(), nn.Identity(), _Param((cout, cout, 3, 3), kind="conv"))
        if cin != cout:
            self.skip_connection = _Param((cout, cin, 1, 1), kind="conv")
        else:
            self.skip_connection = nn.Identity()
        self.use_temporal_conv = temporal_conv
        if temporal_conv:
            self.temopral_conv = _TemporalConvBlock(cout)


class _TemporalConvBlock(nn.Module):
    """openaimodel3d.py:242-267"""

    def __init__(self, ch):
        super().__init__()
        self.conv1 = _seq(_Param((ch,), kind="norm"), nn.Identity(), _Param((ch, ch, 3, 1, 1), kind="conv"))
        for name in ("conv2", "conv3", "conv4"):
            setattr(self, name, _seq(_Param((ch,), kind="norm"), nn.Identity(), nn.Identity(), _Param((ch, ch, 3, 1, 1), kind="conv")))


class _CrossAttention(nn.Module):
    """attention.py:45-57"""

    def __init__(self, query_dim, context_dim, heads, dim_head):
        super().__init__()
        inner = heads * dim_head
        context_dim = query_dim if context_dim is None else context_dim
        self.heads, self.dim_head, self.is_self = heads, dim_head, context_dim == query_dim
        self.to_q = _Param((inner, query_dim), bias=False)
        self.to_k = _Param((inner, context_dim), bias=False)
        self.to_v = _Param((inner, context_dim), bias=False)
        self.to_out = _seq(_Param((query_dim, inner)), nn.Identity())


class _GEGLU(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = _Param((dim_out * 2, dim_in))


class _FeedForward(nn.Module):
    """attention.py:386-403 (glu=True, mult=4)"""

    def __init__(self, dim):
        super().__init__()
        self.net = _seq(_GEGLU(dim, dim * 4), nn.Identity(), _Param((dim, dim * 4)))


class _BasicTransformerBlock(nn.Module):
    """attention.py:189-202"""

    def __init__(self, dim, n_heads, d_head, context_dim):
        super().__init__()
        self.attn1 = _CrossAttention(dim, None, n_heads, d_head)
        self.ff = _FeedForward(dim)
        self.attn2 = _CrossAttention(dim, context_dim, n_heads, d_head)
        self.attn2.is_self = context_dim is None
        self.norm1 = _Param((dim,), kind="norm")
        self.norm2 = _Param((dim,), kind="norm")
        self.norm3 = _Param((dim,), kind="norm")


class _SpatialTransformer(nn.Module):
    """attention.py:233-259"""

    def __init__(self, ch, n_heads, d_head, depth, context_dim, use_linear):
        super().__init__()
        inner = n_heads * d_head
        self.ch, self.inner, self.heads = ch, inner, n_heads
        self.norm = _Param((ch,), kind="norm")
        self.proj_in = _Param((inner, ch) if use_linear else (inner, ch, 1, 1))
        self.transformer_blocks = nn.ModuleList([_BasicTransformerBlock(inner, n_heads, d_head, context_dim) for _ in range(depth)])
        self.proj_out = _Param((ch, inner) if use_linear else (ch, inner, 1, 1))


class _TemporalTransformer(nn.Module):
    """attention.py:288-329 (only_self_att=True)"""

    def __init__(self, ch, n_heads, d_head, depth, use_linear):
        super().__init__()
        inner = n_heads * d_head
        self.ch, self.inner, self.heads = ch, inner, n_heads
        self.norm = _Param((ch,), kind="norm")
        self.proj_in = _Param((inner, ch) if use_linear else (inner, ch, 1))
        self.transformer_blocks = nn.ModuleList([_BasicTransformerBlock(inner, n_heads, d_head, None) for _ in range(depth)])
        self.proj_out = _Param((ch, inner) if use_linear else (ch, inner, 1))


class _Downsample(nn.Module):
    def __init__(self, ch):
        super().__init__()
        self.op = _Param((ch, ch, 3, 3), kind="conv")


class _Upsample(nn.Module):
    def __init__(self, ch):
        super().__init__()
        self.conv = _Param((ch, ch, 3, 3), kind="conv")


# --------------------------------------------------------------------------------------
# buffer pool for the recorded forward
# --------------------------------------------------------------------------------------
class _Pool:
    def __init__(self, device):
        self.device = device
        self.free = {}
        self.total_bytes = 0

    def get(self, rows, cols, dtype=torch.float16):
        key = (rows * cols, dtype)
        lst = self.free.get(key)
        if lst:
            return lst.pop().view(rows, cols)
        t = torch.empty(rows * cols, dtype=dtype, device=self.device)
        self.total_bytes += t.numel() * t.element_size()
        return t.view(rows, cols)

    def put(self, *ts):
        for t in ts:
            if t is None:
                continue
            flat = t.reshape(-1)
            self.free.setdefault((flat.numel(), flat.dtype), []).append(flat)


class _FMap:
    """channels-last feature map: buf [F*H*W][C] fp16; `colsum` = (f32 [row tiles][C][2] buffer, rows per tile) when the
    GEMM that produced it also left per-(row tile, channel) sums and sums of squares behind (GroupNorm statistics)"""
    __slots__ = ("buf", "F", "H", "W", "C", "colsum", "src", "gstat", "cs_used", "cs_rows", "gstat_own", "own_F", "slabs")

    def __init__(self, buf, F, H, W, C, colsum=None, src=None, gstat=None):
        self.buf, self.F, self.H, self.W, self.C, self.colsum = buf, F, H, W, C, colsum
        self.src = src          # index of the recorded GEMM launch that produced buf (with colsum): plan.gn() may re-target it
        self.gstat = gstat      # (f64 accumulators, frames_per_stat) when the producer already accumulated finished statistics
        self.cs_used = False    # a GroupNorm consumed the column sums (else the producer is re-recorded without them)
        self.cs_rows = colsum[1] if colsum is not None else 0   # rows per tile of the producer's statistics epilogue (kept when colsum is dropped)
        self.gstat_own = None   # (accumulators, frames_per_stat): the producer was re-targeted to FINISHED statistics of this map (32 groups of C / 32)
        self.own_F = 0          # > 0: gstat_own covers own_F frames and buf is F / own_F copies of them (the repeat that ends the shared prefix)
        self.slabs = None       # (split-K workspace, index of the recorded GEMM launch): buf's producer ran split-K and its reduce may still move into the consuming GroupNorm

    @property
    def M(self):
        return self.F * self.H * self.W


class _CatMap:
    """the VIRTUAL torch.cat([h, skip], dim=channels) of openaimodel3d.py:571: never materialised -- its two consumers (ResBlock.in_layers[0]
    and ResBlock.skip_connection, :149,190-195) read h and skip where their producers left them.  `gcat` = statistics accumulators in the
    concat's grouping (h's share, and skip's unless `gb` = skip's own finished statistics is merged by the GroupNorm)"""
    __slots__ = ("h", "skip", "gcat", "gb", "Fb")

    def __init__(self, h, skip, gcat, gb, Fb=0):
        self.h, self.skip, self.gcat, self.gb, self.Fb = h, skip, gcat, gb, Fb

    F = property(lambda self: self.h.F)
    H = property(lambda self: self.h.H)
    W = property(lambda self: self.h.W)
    C = property(lambda self: self.h.C + self.skip.C)
    M = property(lambda self: self.h.M)



def pack_tree(root, dev):
    """Pack every block under `root` (a whole UNetModel, or a holder of ONE block: blockplan.BlockRunner) into GEMM operands.
    Returns (P, emb_cols, kv_cols): P maps id(parameter module) -> packed operands; all ResBlock.emb_layers Linears share
    the input SiLU(emb) and all cross-attention to_k/to_v share the text context, so each family is fused into ONE wide GEMM
    per forward (P["emb_all"], P["ctx_kv_all"]; column offsets are multiples of 64) with the per-module column ranges in
    emb_cols / kv_cols."""
    P = {}
    f32 = lambda t: t.detach().to(device=dev, dtype=torch.float32).contiguous()

    def norm(m):
        P[id(m)] = (f32(m.weight), f32(m.bias))

    def lin(m):
        P[id(m)] = ops.pack_linear(m.weight.detach(), None if m.bias is None else m.bias.detach(), device=dev)

    for mod in root.modules():
        if isinstance(mod, _ResBlock):
            norm(mod.in_layers[0]); norm(mod.out_layers[0])
            c1 = mod.in_layers[2]
            P[id(c1)] = ops.pack_conv3x3(c1.weight.detach(), c1.bias.detach(), device=dev)
            c2 = mod.out_layers[3]
            P[id(c2)] = ops.pack_conv3x3(c2.weight.detach(), c2.bias.detach(), device=dev)
            if isinstance(mod.skip_connection, _Param):
                sk = mod.skip_connection
                P[id(sk)] = ops.pack_conv1x1(sk.weight.detach(), sk.bias.detach(), device=dev)
        elif isinstance(mod, _TemporalConvBlock):
            for name, idx in (("conv1", 2), ("conv2", 3), ("conv3", 3), ("conv4", 3)):
                sq = getattr(mod, name)
                norm(sq[0])
                P[id(sq[idx])] = ops.pack_tconv3(sq[idx].weight.detach(), sq[idx].bias.detach(), device=dev)
        elif isinstance(mod, (_SpatialTransformer, _TemporalTransformer)):
            norm(mod.norm); lin(mod.proj_in); lin(mod.proj_out)
        elif isinstance(mod, _BasicTransformerBlock):
            norm(mod.norm1); norm(mod.norm2); norm(mod.norm3)
            for att in (mod.attn1, mod.attn2):
                if att.is_self:
                    P[id(att)] = ops.pack_linear_cat([att.to_q.weight.detach(), att.to_k.weight.detach(), att.to_v.weight.detach()], device=dev)
                else:
                    P[id(att)] = ops.pack_linear(att.to_q.weight.detach(), device=dev)
                lin(att.to_out[0])
            g = mod.ff.net[0].proj
            P[id(g)] = ops.pack_geglu(g.weight.detach().to(dev), g.bias.detach().to(dev), device=dev)
            lin(mod.ff.net[2])
        elif isinstance(mod, _Downsample):
            P[id(mod.op)] = ops.pack_conv3x3(mod.op.weight.detach(), mod.op.bias.detach(), device=dev)
        elif isinstance(mod, _Upsample):
            P[id(mod.conv)] = ops.pack_conv3x3(mod.conv.weight.detach(), mod.conv.bias.detach(), device=dev)
            if mod.conv.weight.shape[1] % 64 == 0:          # the four 2 x 2 phase convs (ops.pack_upconv_phases), used where they pay
                P[("up_phases", id(mod.conv))] = ops.pack_upconv_phases(mod.conv.weight.detach(), mod.conv.bias.detach(), device=dev)
    res = [m for m in root.modules() if isinstance(m, _ResBlock)]
    emb_cols, off = {}, 0
    for m in res:
        emb_cols[id(m)] = (off, m.cout)
        off += m.cout
    if res:
        P["emb_all"] = ops.pack_linear_cat([m.emb_layers[1].weight.detach() for m in res],
                                           [m.emb_layers[1].bias.detach() for m in res], device=dev)
    cross = [b.attn2 for b in root.modules() if isinstance(b, _BasicTransformerBlock) and not b.attn2.is_self]
    kv_cols, off = {}, 0
    ws = []
    for a in cross:
        inner = a.to_k.weight.shape[0]
        kv_cols[id(a)] = (off, inner)
        off += 2 * inner
        ws += [a.to_k.weight.detach(), a.to_v.weight.detach()]
    if ws:
        P["ctx_kv_all"] = ops.pack_linear_cat(ws, device=dev)
    return P, emb_cols, kv_cols


# --------------------------------------------------------------------------------------
# the model
# --------------------------------------------------------------------------------------
_SAME_FPS_CACHE = {}          # (id, version) of two fps tensors -> verdict: one device read-back per pair of tensors, not per DDIM step


def same_fps(fps_list):
    """True when every entry of a per-segment fps list describes the same frame rates (ints, or tensors with equal values).
    Comparing two distinct tensors reads the device; the verdict is cached per pair of (tensor object, in-place version), so the
    host-driven paths (p_sample_ddim, unet_windows, forward_segments: one call per step / window batch) pay the read-back once."""
    first = fps_list[0]
    for f in fps_list[1:]:
        if f is first:
            continue
        if isinstance(first, int) and isinstance(f, int):
            if f != first:
                return False
            continue
        key = None
        if torch.is_tensor(first) and torch.is_tensor(f):
            key = (id(first), first._version, id(f), f._version)
            hit = _SAME_FPS_CACHE.get(key)
            if hit is not None and hit[0]() is first and hit[1]() is f:
                if not hit[2]:
                    return False
                continue
        a = torch.as_tensor(first).reshape(-1).to(torch.int64).cpu()
        b = torch.as_tensor(f).reshape(-1).to(torch.int64).cpu()
        ok = True
        if a.shape != b.shape:
            if a.numel() != 1 and b.numel() != 1:
                ok = False
            else:
                a, b = torch.broadcast_tensors(a, b)
        ok = ok and bool(torch.equal(a, b))
        if key is not None:
            import weakref
            if len(_SAME_FPS_CACHE) > 64:
                _SAME_FPS_CACHE.clear()
            _SAME_FPS_CACHE[key] = (weakref.ref(first), weakref.ref(f), ok)
        if not ok:
            return False
    return True


class UNetModel(nn.Module):
    def __init__(self, in_channels, model_channels, out_channels, num_res_blocks, attention_resolutions,
                 dropout=0.0, channel_mult=(1, 2, 4, 8), conv_resample=True, dims=2, context_dim=None,
                 use_scale_shift_norm=False, resblock_updown=False, num_heads=-1, num_head_channels=-1,
                 transformer_depth=1, use_linear=False, use_checkpoint=False, temporal_conv=False,
                 tempspatial_aware=False, temporal_attention=True, temporal_selfatt_only=True,
                 use_relative_position=True, use_causal_attention=False, temporal_length=None, use_fp16=False,
                 addition_attention=False, use_image_attention=False, temporal_transformer_depth=1, fps_cond=False):
        super().__init__()
        # same asserts as openaimodel3d.py:339-342
        if num_heads == -1:
            assert num_head_channels != -1, 'Either num_heads or num_head_channels has to be set'
        if num_head_channels == -1:
            assert num_heads != -1, 'Either num_heads or num_head_channels has to be set'
        unsupported = []
        if dims != 2: unsupported.append("dims != 2")
        if use_scale_shift_norm: unsupported.append("use_scale_shift_norm")
        if resblock_updown: unsupported.append("resblock_updown")
        if tempspatial_aware: unsupported.append("tempspatial_aware")
        if use_relative_position: unsupported.append("use_relative_position")
        if use_causal_attention: unsupported.append("use_causal_attention")
        if use_image_attention: unsupported.append("use_image_attention")
        if not temporal_selfatt_only: unsupported.append("temporal_selfatt_only=False")
        if not conv_resample: unsupported.append("conv_resample=False")
        if num_head_channels != 64: unsupported.append("num_head_channels != 64 (kernels are head-dim 64)")
        if model_channels % 32 or model_channels % 64: unsupported.append("model_channels % 64 != 0")
        if unsupported:
            raise NotImplementedError("moca_video_amd.UNetModel covers the inference_t2v_512_v2.0.yaml surface; "
                                      "unsupported: " + ", ".join(unsupported))
        self.in_channels, self.model_channels, self.out_channels = in_channels, model_channels, out_channels
        self.num_res_blocks = num_res_blocks
        self.attention_resolutions = list(attention_resolutions)
        self.dropout = dropout            # inference only: Dropout is the identity
        self.channel_mult = list(channel_mult)
        self.temporal_attention = temporal_attention
        self.use_checkpoint = use_checkpoint  # accepted and ignored (no autograd on this path)
        self.dtype = torch.float16            # compute dtype of the HIP path (reference: :355)
        self.addition_attention = addition_attention
        self.fps_cond = fps_cond
        self.context_dim = context_dim
        self.temporal_length = temporal_length
        time_embed_dim = model_channels * 4

        self.time_embed = _seq(_Param((time_embed_dim, model_channels)), nn.Identity(), _Param((time_embed_dim, time_embed_dim)))
        if fps_cond:
            self.fps_embedding = _seq(_Param((time_embed_dim, model_channels)), nn.Identity(), _Param((time_embed_dim, time_embed_dim)))

        self.input_blocks = nn.ModuleList([_seq(_Param((model_channels, in_channels, 3, 3), kind="conv"))])
        if addition_attention:
            self.init_attn = _seq(_TemporalTransformer(model_channels, 8, num_head_channels, transformer_depth, use_linear=False))

        def attn_layers(ch):
            heads = ch // num_head_channels
            layers = [_SpatialTransformer(ch, heads, num_head_channels, transformer_depth, context_dim, use_linear)]
            if temporal_attention:
                layers.append(_TemporalTransformer(ch, heads, num_head_channels, temporal_transformer_depth, use_linear))
            return layers

        input_block_chans = [model_channels]
        ch, ds = model_channels, 1
        for level, mult in enumerate(channel_mult):
            for _ in range(num_res_blocks):
                layers = [_ResBlock(ch, time_embed_dim, mult * model_channels, temporal_conv)]
                ch = mult * model_channels
                if ds in attention_resolutions:
                    layers += attn_layers(ch)
                self.input_blocks.append(_seq(*layers))
                input_block_chans.append(ch)
            if level != len(channel_mult) - 1:
                self.input_blocks.append(_seq(_Downsample(ch)))
                input_block_chans.append(ch)
                ds *= 2
        mid = [_ResBlock(ch, time_embed_dim, ch, temporal_conv),
               _SpatialTransformer(ch, ch // num_head_channels, num_head_channels, transformer_depth, context_dim, use_linear)]
        if temporal_attention:
            mid.append(_TemporalTransformer(ch, ch // num_head_channels, num_head_channels, temporal_transformer_depth, use_linear))
        mid.append(_ResBlock(ch, time_embed_dim, ch, temporal_conv))
        self.middle_block = _seq(*mid)

        self.output_blocks = nn.ModuleList([])
        for level, mult in list(enumerate(channel_mult))[::-1]:
            for i in range(num_res_blocks + 1):
                ich = input_block_chans.pop()
                layers = [_ResBlock(ch + ich, time_embed_dim, mult * model_channels, temporal_conv)]
                ch = model_channels * mult
                if ds in attention_resolutions:
                    layers += attn_layers(ch)
                if level and i == num_res_blocks:
                    layers.append(_Upsample(ch))
                    ds //= 2
                self.output_blocks.append(_seq(*layers))
        self.out = _seq(_Param((ch,), kind="norm"), nn.Identity(), _Param((out_channels, model_channels, 3, 3), kind="conv"))

        self._packed = None       # id(param module) -> packed GEMM operands
        self._plans = {}          # (B,T,H,W,L,in_dtype) -> _Plan
        self.use_graph = True
        self.register_load_state_dict_post_hook(lambda m, k: m._invalidate())
        _l.load()                 # fail loudly at construction time if the HIP library is missing

    # ---- weights -----------------------------------------------------------------------
    def _invalidate(self):
        self._packed = None
        self._packed_only = False
        for pl in getattr(self, "_plans", {}).values():      # the hipGraphExec of a dropped plan is a device-side object
            pl.close()
        self._plans = {}

    def _apply(self, fn, recurse=True):   # .cuda()/.to() moves parameters: re-pack lazily
        out = super()._apply(fn, recurse)
        self._invalidate()
        return out

    def _pack(self):
        dev = next(self.parameters()).device
        if dev.type != "cuda":
            raise RuntimeError("moca_video_amd.UNetModel runs on an MI355X only; call .cuda() first (no CPU path)")
        P, self._emb_cols, self._kv_cols = pack_tree(self, dev)
        f32 = lambda t: t.detach().to(device=dev, dtype=torch.float32).contiguous()
        for sq in (self.time_embed,) + ((self.fps_embedding,) if self.fps_cond else ()):
            for m in (sq[0], sq[2]):
                P[id(m)] = ops.pack_linear(m.weight.detach(), m.bias.detach(), device=dev)
        cin = self.input_blocks[0][0]
        P[id(cin)] = ops.pack_conv3x3(cin.weight.detach(), cin.bias.detach(), cpad=8, device=dev)
        P[id(self.out[0])] = (f32(self.out[0].weight), f32(self.out[0].bias))
        P[id(self.out[2])] = ops.pack_conv3x3(self.out[2].weight.detach(), self.out[2].bias.detach(), device=dev)
        self._packed = P

    # ---- forward -----------------------------------------------------------------------
    def _prepare(self, x, timesteps, context, features_adapter, fps, check_context=True):
        """argument checks + per-(b,t) timestep / fps rows shared by forward() and forward_concurrent()"""
        if features_adapter is not None:
            raise NotImplementedError("features_adapter is always None on the MoCA path")
        if context is None:
            raise ValueError("context is required (conditioning_key='crossattn')")
        if x.dim() != 5 or not x.is_cuda:
            raise ValueError("x must be a CUDA tensor [B, C, T, h, w]")
        if self._packed is None:
            self._pack()
        B, Cin, T, H, W = x.shape
        assert Cin == self.in_channels
        if T > 16:
            raise ValueError("temporal attention kernel supports T <= 16 (temporal_length of the YAML)")
        timesteps = torch.as_tensor(timesteps, device=x.device).reshape(-1).to(torch.int64)
        n_t = timesteps.shape[0]
        if n_t == B:                       # not is_fifo: emb.repeat_interleave(T) (:548-549)
            t_rows = timesteps.repeat_interleave(T)
        elif B == 1 and n_t == T:          # is_fifo (:535): one timestep per frame
            t_rows = timesteps
        elif n_t == B * T:                 # extension: per-(b,t) timesteps
            t_rows = timesteps
        else:
            raise ValueError(f"timesteps has {n_t} entries for x of batch {B}, frames {T}")
        if isinstance(fps, int):
            fps_rows = torch.full_like(t_rows, fps)       # :540-541
        else:
            fps = torch.as_tensor(fps, device=x.device).reshape(-1).to(torch.int64)
            if fps.shape[0] == B:
                fps_rows = fps.repeat_interleave(T)
            elif fps.shape[0] == B * T:
                fps_rows = fps
            elif fps.shape[0] == 1:
                fps_rows = fps.expand(B * T)
            else:
                raise ValueError("fps must be an int or a tensor of B entries")
        if check_context and context.shape[0] != B:
            raise ValueError("context batch must equal x batch")
        return t_rows, fps_rows

    def _plan_for(self, x, L, replica=0, shared_x=False):
        """L: context tokens, or a tuple of (videos, tokens) segments (see _Plan.segs); shared_x: x holds the distinct latents of
        a batch that repeats them once per segment"""
        B, _, T, H, W = x.shape
        if shared_x:
            B *= len(L)
        key = (B, T, H, W, L, x.dtype, x.device.index, replica, shared_x)
        plan = self._plans.get(key)
        if plan is None and getattr(self, "_packed_only", False):
            raise RuntimeError("this rank received the packed operand set of the plans built before dist.broadcast_packed and holds no "
                               "parameters to pack a new plan from")
        if plan is None:
            plan = _Plan(self, B, T, H, W, L, x.dtype, x.device, shared_x=shared_x)
            self._plans[key] = plan
        return plan

    @torch.no_grad()
    def forward(self, x, timesteps, context=None, features_adapter=None, fps=16, **kwargs):
        """openaimodel3d.py:534-578.  x [B,C,T,h,w]; timesteps int64 [B] (or [T] with B == 1: the
        FIFO per-frame-timestep path, :535; or [B*T] per-(b,t), an extension); context [B,L,ctx];
        fps int or [B]; unknown kwargs (clean_cond, gamma, ...) are ignored exactly as upstream."""
        t_rows, fps_rows = self._prepare(x, timesteps, context, features_adapter, fps)
        return self._plan_for(x, context.shape[1]).run(x, t_rows, fps_rows, context)

    @torch.no_grad()
    def forward_segments(self, x, timesteps, contexts, fps=16, shared_x=False):
        """One forward over a batch whose videos carry contexts of DIFFERENT lengths: `contexts` = list of [n_i, L_i, D]
        tensors in batch order (sum n_i = B), e.g. the 2n conditional FIFO windows with two prompts (154 tokens) followed
        by their unconditional copies (77 tokens).  Same values as one forward() per segment: every UNet op is per-sample
        and each cross-attention runs per segment on its own keys (no padding, no masking).

        shared_x=True: the segments are context variants of the SAME latents -- the two `apply_model` calls of classifier-free
        guidance (ddim.py:298-299,366-369).  x [n, ...], timesteps and fps describe the n distinct videos, every context is
        [n, L_i, D]; returns [len(contexts) * n, ...] (segment-major).  Everything before the first cross-attention is computed
        once (see _Plan) with the fps embedding of the first segment, so a per-segment fps list must hold EQUAL entries
        (ValueError otherwise: `same_fps`)."""
        n = x.shape[0]
        if shared_x:
            if any(c.shape[0] != n for c in contexts):
                raise ValueError("shared_x: every context must have one row block per latent video")
            segs = tuple((n, int(c.shape[1])) for c in contexts)
            fps_list = list(fps) if isinstance(fps, (list, tuple)) else [fps] * len(contexts)
            rows = [self._prepare(x, timesteps, contexts[0], None, f, check_context=False) for f in fps_list]
            if not same_fps(fps_list):
                # the shared prefix (conv_in .. the first ResBlock) adds ONE fps embedding to the rows both branches read
                raise ValueError("shared_x: the segments share everything before the first cross-attention, so their fps must be "
                                 "equal (use shared_x=False for branches with different fps)")
            t_rows = torch.cat([r[0] for r in rows])
            fps_rows = torch.cat([r[1] for r in rows])
            return self._plan_for(x, segs, shared_x=True).run(x, t_rows, fps_rows, list(contexts))
        segs = tuple((int(c.shape[0]), int(c.shape[1])) for c in contexts)
        if sum(nv for nv, _ in segs) != n:
            raise ValueError("contexts must cover the batch of x")
        t_rows, fps_rows = self._prepare(x, timesteps, contexts[0], None, fps, check_context=False)
        return self._plan_for(x, segs).run(x, t_rows, fps_rows, list(contexts))

    @torch.no_grad()
    def forward_concurrent(self, calls):
        """Several independent forwards (e.g. the conditional and unconditional CFG branch) launched as
        separate hipGraphs on separate streams so the GPU overlaps them: one chain's partial last round of
        tiles, kernel prologues/epilogues and launch gaps are filled by the other chain's kernels.
        calls: list of dicts(x, timesteps, context, fps).  Returns the list of outputs (same values as forward)."""
        prepared = []
        for i, c in enumerate(calls):
            t_rows, fps_rows = self._prepare(c["x"], c["timesteps"], c["context"], None, c.get("fps", 16))
            prepared.append((self._plan_for(c["x"], c["context"].shape[1], replica=i), c["x"], t_rows, fps_rows, c["context"]))
        cur = torch.cuda.current_stream(prepared[0][1].device)
        outs = [pl.launch_async(x, t, f, ctx, cur) for pl, x, t, f, ctx in prepared]
        for (pl, *_), o in zip(prepared, outs):
            cur.wait_stream(pl.stream)
            o.record_stream(cur)
        return outs


from .plan import _Plan  # noqa: E402  (split for readability; needs the classes above)
