"""MI355X-native text tower of `FrozenOpenCLIPEmbedder` (`lvdm/modules/encoders/condition.py:174-235`):
token + positional embedding, the pre-LN causal transformer (all resblocks for layer="last", all but the last for
"penultimate", as the YAML asks), `ln_final`.  Runs once per prompt on 77 tokens, so nothing here is tuned: LayerNorm,
the fused in_proj GEMM, causal head-dim-64 attention, out_proj + residual, c_fc with the exact-GELU epilogue, c_proj + residual
are the UNet's kernels (plus `moca_attention_causal_f16`, `MOCA_EP_GELU`, `moca_embed_tokens_f16`).

The transformer blocks themselves are `open_clip_torch==2.30.0` code (requirements.txt:196), which is not in this image and whose
weights (`laion2b_s32b_b79k`) and BPE vocabulary are not available offline: the module mirrors open_clip's parameter names
(`model.token_embedding.weight`, `model.positional_embedding`, `model.transformer.resblocks.N.{ln_1,attn.in_proj_weight,
attn.in_proj_bias,attn.out_proj,ln_2,mlp.c_fc,mlp.c_proj}`, `model.ln_final`, `model.text_projection`, `model.logit_scale`) so a
`cond_stage_model.*` checkpoint loads; it takes TOKEN IDS, or strings once `.tokenizer` holds a `SimpleTokenizer` (tokenizer.py: the
published byte-level BPE, parity unpinned for the same reason).
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import ops
from .unet import _Param

__all__ = ["FrozenOpenCLIPEmbedder"]


class _MHA(nn.Module):
    def __init__(self, width):
        super().__init__()
        self.in_proj_weight = nn.Parameter(torch.empty(3 * width, width), requires_grad=False)
        self.in_proj_bias = nn.Parameter(torch.empty(3 * width), requires_grad=False)
        self.out_proj = _Param((width, width))


class _ResBlock(nn.Module):
    def __init__(self, width, mlp_ratio=4):
        super().__init__()
        self.ln_1 = _Param((width,), kind="norm")
        self.attn = _MHA(width)
        self.ln_2 = _Param((width,), kind="norm")
        self.mlp = nn.Module()
        self.mlp.c_fc = _Param((mlp_ratio * width, width))
        self.mlp.c_proj = _Param((width, mlp_ratio * width))


class _TextModel(nn.Module):
    def __init__(self, vocab, width, layers, context):
        super().__init__()
        self.token_embedding = _Param((vocab, width), bias=False)
        self.positional_embedding = nn.Parameter(torch.empty(context, width), requires_grad=False)
        self.transformer = nn.Module()
        self.transformer.resblocks = nn.ModuleList([_ResBlock(width) for _ in range(layers)])
        self.ln_final = _Param((width,), kind="norm")
        self.text_projection = nn.Parameter(torch.empty(width, width), requires_grad=False)     # unused by the embedder
        self.logit_scale = nn.Parameter(torch.empty(()), requires_grad=False)


class FrozenOpenCLIPEmbedder(nn.Module):
    LAYERS = ["last", "penultimate"]

    def __init__(self, arch="ViT-H-14", version="laion2b_s32b_b79k", device="cuda", max_length=77, freeze=True, layer="last",
                 vocab_size=49408, width=1024, heads=16, layers=24):
        super().__init__()
        assert layer in self.LAYERS
        if arch != "ViT-H-14" and (width, heads, layers) == (1024, 16, 24):
            raise NotImplementedError(f"arch {arch!r}: pass width/heads/layers explicitly (only the ViT-H-14 text tower is built in)")
        if width % heads or width // heads != 64:
            raise NotImplementedError("the attention kernel is head-dim 64")
        self.model = _TextModel(vocab_size, width, layers, max_length)
        self.max_length, self.heads, self.width = max_length, heads, width
        self.layer = layer
        self.layer_idx = 0 if layer == "last" else 1
        self._packed = None
        self.tokenizer = None          # `open_clip.tokenize` stand-in (moca_video_amd.tokenizer.SimpleTokenizer) once a vocabulary exists
        self.register_load_state_dict_post_hook(lambda module, incompatible: setattr(module, "_packed", None))

    def _apply(self, fn, recurse=True):
        out = super()._apply(fn, recurse)
        self._packed = None
        return out

    def _pack(self):
        dev = self.model.positional_embedding.device
        if dev.type != "cuda":
            raise RuntimeError("moca_video_amd.FrozenOpenCLIPEmbedder runs on an MI355X only; call .cuda() first (no CPU path)")
        f32 = lambda t: t.detach().to(device=dev, dtype=torch.float32).contiguous()
        P = {"tok": f32(self.model.token_embedding.weight), "pos": f32(self.model.positional_embedding),
             "ln_final": (f32(self.model.ln_final.weight), f32(self.model.ln_final.bias)), "blocks": []}
        for r in self.model.transformer.resblocks:
            P["blocks"].append(dict(
                ln1=(f32(r.ln_1.weight), f32(r.ln_1.bias)), ln2=(f32(r.ln_2.weight), f32(r.ln_2.bias)),
                qkv=ops.pack_linear(r.attn.in_proj_weight.detach(), r.attn.in_proj_bias.detach(), device=dev),
                out=ops.pack_linear(r.attn.out_proj.weight.detach(), r.attn.out_proj.bias.detach(), device=dev),
                fc=ops.pack_linear(r.mlp.c_fc.weight.detach(), r.mlp.c_fc.bias.detach(), device=dev),
                proj=ops.pack_linear(r.mlp.c_proj.weight.detach(), r.mlp.c_proj.bias.detach(), device=dev)))
        self._packed = P

    def forward(self, text):
        """condition.py:205-209: `tokens = open_clip.tokenize(text)`.  Strings need `self.tokenizer` (a
        `moca_video_amd.tokenizer.SimpleTokenizer` built from the CLIP merges file, which is not available offline); token ids
        [B, 77] are taken as they are."""
        if isinstance(text, (str, list, tuple)) and not torch.is_tensor(text):
            if self.tokenizer is None:
                raise NotImplementedError("no BPE vocabulary: set `.tokenizer = SimpleTokenizer(<bpe_simple_vocab_16e6.txt.gz>)` or pass "
                                          "token ids [B, 77]")
            text = self.tokenizer(text, self.max_length).to(self.model.positional_embedding.device)
        return self.encode_with_transformer(text)

    encode = forward

    @torch.no_grad()
    def encode_with_transformer(self, tokens):
        """condition.py:205-225: tokens int64 [B, L] -> [B, L, width] fp32"""
        if tokens.dim() != 2 or tokens.shape[1] != self.max_length:
            raise ValueError(f"expected token ids [B, {self.max_length}], got {tuple(tokens.shape)}")
        if not tokens.is_cuda:
            raise ValueError("moca_video_amd.FrozenOpenCLIPEmbedder needs CUDA (HIP) token ids; there is no CPU path")
        if self._packed is None:
            self._pack()
        P, L, C, H = self._packed, self.max_length, self.width, self.heads
        dev = tokens.device
        outs = []
        ops.set_stream(None)
        new = lambda cols, dt=torch.float16: torch.empty(L, cols, dtype=dt, device=dev)
        for b in range(tokens.shape[0]):                  # one prompt at a time: M = 77 rows (the 128-row GEMM kernel)
            x = new(C)
            ops.embed_tokens(tokens[b].contiguous().long(), P["tok"], P["pos"], x, n_tokens=L, L=L, Cn=C, vocab=P["tok"].shape[0])
            nblk = len(P["blocks"]) - self.layer_idx      # text_transformer_forward: stop before the last `layer_idx` blocks
            for blk in P["blocks"][:nblk]:
                l = ops.layernorm(x, new(C), *blk["ln1"], M=L, Cn=C)
                qkv = ops.gemm(l, blk["qkv"], new(3 * C), M=L)
                a = ops.attention_causal(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], new(C), B=1, heads=H, N=L,
                                         ldq=3 * C, ldk=3 * C, ldv=3 * C, ldo=C, scale=64 ** -0.5)
                x = ops.gemm(a, blk["out"], new(C), M=L, residual=x)
                l = ops.layernorm(x, new(C), *blk["ln2"], M=L, Cn=C)
                h = ops.gemm(l, blk["fc"], new(4 * C), M=L, gelu=True)
                x = ops.gemm(h, blk["proj"], new(C), M=L, residual=x)
            outs.append(ops.layernorm(x, new(C), *P["ln_final"], M=L, Cn=C).float())
        return torch.stack(outs, 0)
