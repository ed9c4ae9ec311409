"""Text tower of FrozenOpenCLIPEmbedder.  CPU: the oracle restatement against transformers.CLIPTextModel (an
independent implementation of the published CLIP text transformer) with shared weights -- "penultimate" = final LayerNorm of
hidden_states[-2], "last" = last_hidden_state.  GPU: the HIP path against the oracle.  (The reference's own blocks are
open_clip_torch, absent offline: parity for this row is pinned by that cross-check only.)"""
import pytest
import torch

from helpers import relerr, state_dict_for
from oracle import clip_text_oracle as CO

CFG = dict(vocab_size=1000, width=128, heads=2, layers=3)


def _model(layer="penultimate", **kw):
    from moca_video_amd.clip_text import FrozenOpenCLIPEmbedder
    return FrozenOpenCLIPEmbedder(layer=layer, **{**CFG, **kw})


def _tokens(B=2, vocab=1000, seed=0):
    g = torch.Generator().manual_seed(seed)
    return torch.randint(0, vocab, (B, 77), generator=g)


def test_state_dict_names_follow_open_clip():
    sd = _model().state_dict()
    for k in ("model.token_embedding.weight", "model.positional_embedding", "model.transformer.resblocks.0.attn.in_proj_weight",
              "model.transformer.resblocks.2.mlp.c_proj.bias", "model.ln_final.weight", "model.text_projection", "model.logit_scale"):
        assert k in sd, k
    assert sd["model.transformer.resblocks.0.attn.in_proj_weight"].shape == (384, 128)
    from moca_video_amd.clip_text import FrozenOpenCLIPEmbedder
    full = FrozenOpenCLIPEmbedder(layer="penultimate")
    assert sum(v.numel() for v in full.state_dict().values()) == 354_032_641      # ViT-H-14 text tower + text_projection + logit_scale


@pytest.mark.parametrize("layer", ["penultimate", "last"])
def test_oracle_vs_transformers_clip_text_model(layer):
    from transformers import CLIPTextConfig, CLIPTextModel
    m = _model(layer)
    sd = state_dict_for(m, 21)
    sd["model.logit_scale"] = torch.tensor(1.0)
    C, H, NL = CFG["width"], CFG["heads"], CFG["layers"]
    hf = CLIPTextModel(CLIPTextConfig(vocab_size=CFG["vocab_size"], hidden_size=C, intermediate_size=4 * C, num_hidden_layers=NL,
                                      num_attention_heads=H, max_position_embeddings=77, hidden_act="gelu", layer_norm_eps=1e-5)).eval()
    hs = {"embeddings.token_embedding.weight": sd["model.token_embedding.weight"],
          "embeddings.position_embedding.weight": sd["model.positional_embedding"],
          "final_layer_norm.weight": sd["model.ln_final.weight"], "final_layer_norm.bias": sd["model.ln_final.bias"]}
    for i in range(NL):
        p, q = f"model.transformer.resblocks.{i}", f"encoder.layers.{i}"
        w, b = sd[p + ".attn.in_proj_weight"], sd[p + ".attn.in_proj_bias"]
        for j, n in enumerate(("q_proj", "k_proj", "v_proj")):
            hs[f"{q}.self_attn.{n}.weight"], hs[f"{q}.self_attn.{n}.bias"] = w[j * C:(j + 1) * C], b[j * C:(j + 1) * C]
        hs[f"{q}.self_attn.out_proj.weight"], hs[f"{q}.self_attn.out_proj.bias"] = sd[p + ".attn.out_proj.weight"], sd[p + ".attn.out_proj.bias"]
        hs[f"{q}.layer_norm1.weight"], hs[f"{q}.layer_norm1.bias"] = sd[p + ".ln_1.weight"], sd[p + ".ln_1.bias"]
        hs[f"{q}.layer_norm2.weight"], hs[f"{q}.layer_norm2.bias"] = sd[p + ".ln_2.weight"], sd[p + ".ln_2.bias"]
        hs[f"{q}.mlp.fc1.weight"], hs[f"{q}.mlp.fc1.bias"] = sd[p + ".mlp.c_fc.weight"], sd[p + ".mlp.c_fc.bias"]
        hs[f"{q}.mlp.fc2.weight"], hs[f"{q}.mlp.fc2.bias"] = sd[p + ".mlp.c_proj.weight"], sd[p + ".mlp.c_proj.bias"]
    target = hf.text_model if hasattr(hf, "text_model") and not any(k.startswith("embeddings") for k in hf.state_dict()) else hf
    missing = target.load_state_dict(hs, strict=False)
    assert not [k for k in missing.missing_keys if "position_ids" not in k], missing.missing_keys
    tok = _tokens()
    with torch.no_grad():
        o = hf(input_ids=tok, output_hidden_states=True)
        lnf = target.final_layer_norm
        ref = lnf(o.hidden_states[-2]) if layer == "penultimate" else o.last_hidden_state
    got = CO.encode_with_transformer(sd, tok, H, layer_idx=1 if layer == "penultimate" else 0)
    assert relerr(got, ref) < 2e-5


@pytest.mark.gpu
@pytest.mark.parametrize("layer", ["penultimate", "last"])
def test_hip_text_tower_vs_oracle(layer):
    m = _model(layer)
    sd = state_dict_for(m, 21)
    m.load_state_dict(sd, strict=True)
    m = m.cuda()
    tok = _tokens(B=3)
    ref = CO.encode_with_transformer(sd, tok, CFG["heads"], layer_idx=m.layer_idx)
    out = m.encode_with_transformer(tok.cuda())
    assert out.shape == ref.shape == (3, 77, 128)
    assert relerr(out.cpu(), ref) < 2e-3          # 1.5 x the observed 1.07e-3 (24 blocks, fp16 storage)
    with pytest.raises(NotImplementedError):
        m(["a prompt"])
    with pytest.raises(ValueError):
        m.encode_with_transformer(tok[:, :50].cuda())
    m.tokenizer, _ = _toy_tokenizer()               # strings go through `open_clip.tokenize`'s stand-in (condition.py:207)
    ids = m.tokenizer(["hello world", ""], 77)
    assert torch.equal(m(["hello world", ""]), m.encode_with_transformer(ids.cuda()))


# ---------------------------------------------------------------- byte-level BPE tokenizer (CPU; synthetic merge table: the real vocabulary is absent)
def _toy_tokenizer():
    from moca_video_amd.tokenizer import SimpleTokenizer
    merges = [("h", "e"), ("l", "l"), ("he", "ll"), ("hell", "o</w>"), ("w", "o"), ("r", "l"), ("wo", "rl"), ("worl", "d</w>"),
              ("c", "a"), ("ca", "t</w>"), ("a", "b"), ("ab", "c</w>")]
    return SimpleTokenizer(merges=merges), merges


def test_tokenizer_bpe_on_a_synthetic_merge_table():
    """id layout (256 byte symbols, 256 word-final ones, one id per merge, start / end), merge PRIORITY (lowest rank first, not left
    to right), word-final marker, lower-casing / whitespace / html clean-up, punctuation and digits split off, utf-8 bytes,
    context padding and truncation that keeps `<end_of_text>` -- the published CLIP algorithm on a 12-merge table"""
    tok, merges = _toy_tokenizer()
    base = 512
    assert len(tok.encoder) == 512 + len(merges) + 2 and tok.sot == 512 + len(merges) and tok.eot == tok.sot + 1
    mid = lambda pair: base + merges.index(pair)
    assert tok.encode("hello") == [mid(("hell", "o</w>"))]
    assert tok.encode("  Hello   WORLD ") == [mid(("hell", "o</w>")), mid(("worl", "d</w>"))]
    assert tok.encode("hello&amp;cat") == [mid(("hell", "o</w>")), 256 + list(tok.byte_encoder.values()).index("&"), mid(("c", "a")) + 1]
    # "hell" is not word-final: h e l l -> he ll -> hell, then no rule for (hell</w>)...: symbols ['he', 'll</w>']?  the final l carries </w>
    assert tok.bpe("hell") == "he l l</w>"                     # ('l','l</w>') is not a rule: only ('l','l') is -> 'he' merges, 'll' cannot
    assert tok.encode("abc ab") == [mid(("ab", "c</w>")), tok.encoder["a"], tok.encoder["b</w>"]]     # ('a','b</w>') is not a rule
    assert tok.encode("7up!") == [tok.encoder["7</w>"], tok.encoder["u"], tok.encoder["p</w>"], tok.encoder["!</w>"]]
    ids = tok.encode("é")                                      # two utf-8 bytes -> two byte symbols, the last one word-final
    assert len(ids) == 2 and ids[0] < 256 <= ids[1] < 512
    assert tok.decode(tok.encode("hello cat é")).strip() == "hello cat é"
    t = tok(["hello world", "", "cat " * 100], context_length=9)
    assert t.shape == (3, 9) and t.dtype == torch.int64
    assert t[0].tolist() == [tok.sot, mid(("hell", "o</w>")), mid(("worl", "d</w>")), tok.eot, 0, 0, 0, 0, 0]
    assert t[1].tolist() == [tok.sot, tok.eot] + [0] * 7
    assert t[2, 0] == tok.sot and t[2, -1] == tok.eot and (t[2, 1:-1] == mid(("ca", "t</w>"))).all()
    with pytest.raises(ValueError):
        type(tok)()


def test_tokenizer_reads_a_merges_file(tmp_path):
    from moca_video_amd.tokenizer import SimpleTokenizer, bytes_to_unicode
    _, merges = _toy_tokenizer()
    import gzip
    body = "#version: 0.2\n" + "\n".join(f"{a} {b}" for a, b in merges) + "\n"
    (tmp_path / "m.txt").write_text(body, encoding="utf-8")
    with gzip.open(tmp_path / "m.txt.gz", "wt", encoding="utf-8") as f:
        f.write(body)
    for name in ("m.txt", "m.txt.gz"):
        tok = SimpleTokenizer(str(tmp_path / name))
        assert tok.encode("hello world") == [512 + 3, 512 + 7]
    b2u = bytes_to_unicode()
    assert len(b2u) == 256 and len(set(b2u.values())) == 256 and b2u[ord("a")] == "a" and b2u[0] == chr(256)
