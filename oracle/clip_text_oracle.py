"""TEST INFRASTRUCTURE ONLY -- CPU fp32 restatement of the text tower behind `FrozenOpenCLIPEmbedder.encode_with_transformer`
(lvdm/modules/encoders/condition.py:205-225).  The resblocks are `open_clip_torch==2.30.0` (requirements.txt:196), a dependency
that is absent from /root/reference and from this image; restated here from its published architecture (CLIP text transformer:
pre-LayerNorm residual blocks, nn.MultiheadAttention with the causal additive mask, MLP c_fc -> GELU -> c_proj), anchored on the
reference's call site (layer "penultimate" = stop one block early, then ln_final).  PINNING: no reference golden can exist
(package, weights and tokenizer are unavailable offline); tests/test_clip_text.py checks this restatement against
`transformers.CLIPTextModel` -- an independent implementation of the same published architecture -- with shared random weights.
Only tests/ may import this file."""
import torch
import torch.nn.functional as F


def encode_with_transformer(sd, tokens, heads, layer_idx=1, pre="model"):
    x = sd[f"{pre}.token_embedding.weight"][tokens] + sd[f"{pre}.positional_embedding"]           # :206-207
    B, L, C = x.shape
    n = 0
    while f"{pre}.transformer.resblocks.{n}.ln_1.weight" in sd:
        n += 1
    mask = torch.full((L, L), float("-inf")).triu_(1)                                             # open_clip build_attention_mask
    for i in range(n - layer_idx):                                                                # :214-217
        p = f"{pre}.transformer.resblocks.{i}"
        h = F.layer_norm(x, (C,), sd[p + ".ln_1.weight"], sd[p + ".ln_1.bias"], 1e-5)
        qkv = F.linear(h, sd[p + ".attn.in_proj_weight"], sd[p + ".attn.in_proj_bias"])
        q, k, v = (t.reshape(B, L, heads, C // heads).transpose(1, 2) for t in qkv.chunk(3, dim=-1))
        att = (q @ k.transpose(-1, -2)) * (C // heads) ** -0.5 + mask
        o = (att.softmax(-1) @ v).transpose(1, 2).reshape(B, L, C)
        x = x + F.linear(o, sd[p + ".attn.out_proj.weight"], sd[p + ".attn.out_proj.bias"])
        h = F.layer_norm(x, (C,), sd[p + ".ln_2.weight"], sd[p + ".ln_2.bias"], 1e-5)
        h = F.gelu(F.linear(h, sd[p + ".mlp.c_fc.weight"], sd[p + ".mlp.c_fc.bias"]))
        x = x + F.linear(h, sd[p + ".mlp.c_proj.weight"], sd[p + ".mlp.c_proj.bias"])
    return F.layer_norm(x, (C,), sd[f"{pre}.ln_final.weight"], sd[f"{pre}.ln_final.bias"], 1e-5)  # :210
