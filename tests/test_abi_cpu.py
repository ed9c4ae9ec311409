"""CPU: the C-ABI library loads and exports every symbol include/moca_hip.h declares (no compute
calls without a GPU); ctypes mirrors the header; argument validation returns MOCA_E_BADARG before
any launch; the host-side mirrors keep the reference's interface."""
import ctypes as C
import inspect
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    src = open(os.path.join(ROOT, "include", "moca_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(moca_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from moca_video_amd import lib
    l = lib.load()
    syms = header_symbols()
    assert len(syms) >= 30
    for s in syms:
        assert hasattr(l, s), f"{s} declared in include/moca_hip.h but not exported"
    assert sorted(lib.SIGNATURES) == syms, "ctypes SIGNATURES and the header disagree"
    assert "gfx950" in lib.version()


def test_gemm_params_struct_matches_header():
    from moca_video_amd import lib
    src = open(os.path.join(ROOT, "include", "moca_hip.h")).read()
    body = src[src.index("typedef struct moca_gemm_params {"):src.index("} moca_gemm_params;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = []
    for line in body.splitlines()[1:]:
        line = line.strip().rstrip(";")
        if not line:
            continue
        decl = re.sub(r"^(const\s+)?(void|float|double|int32_t|uint32_t|int64_t)\s*\*?\s*", "", line)
        names += [n.strip().lstrip("*") for n in decl.split(",") if n.strip()]
    assert names == [f[0] for f in lib.GemmParams._fields_]
    assert C.sizeof(lib.GemmParams) == 288   # (256 + a2 + lda2 + k1 + gstat_cpg + gstat_coff + wgroup_rows + wgroup_stride) 7 pointers + 23 int32 padded to 8, + colsum, ln_gamma, ln_beta, ln_out + ld_ln + ln_eps + rowsum, lnf_part, lnf_wsum + lnf_nparts + pad + gstat + gstat_rows + tattn_scale + sk_sync + sk_big + pad


def _header_struct_fields(name):
    src = open(os.path.join(ROOT, "include", "moca_hip.h")).read()
    body = src[src.index("typedef struct %s {" % name):src.index("} %s;" % name)]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = []
    for line in body.splitlines()[1:]:
        line = line.strip().rstrip(";")
        if not line:
            continue
        decl = re.sub(r"^(const\s+)?(struct\s+)?(void|float|double|int32_t|uint32_t|int64_t|uint64_t|moca_fifo_state)\s*\*?\s*", "", line)
        names += [re.sub(r"\[\d+\]", "", n.strip().lstrip("*")) for n in decl.split(",") if n.strip()]
    return names


def test_fifo_structs_match_header():
    from moca_video_amd import lib
    assert _header_struct_fields("moca_fifo_state") == [f[0] for f in lib.FifoState._fields_]
    assert C.sizeof(lib.FifoState) == 32
    assert _header_struct_fields("moca_fifo_step_params") == [f[0] for f in lib.FifoStepParams._fields_]
    assert C.sizeof(lib.FifoStepParams) == 18 * 8 + 5 * 4 + 6 * 4 + 4          # 18 pointers, 5 floats, 6 ints, tail padding


def test_bad_arguments_are_rejected_without_a_gpu():
    from moca_video_amd import lib
    l = lib.load()
    p = lib.GemmParams()
    assert l.moca_gemm_f16(C.byref(p), None) == -1                 # null pointers
    assert l.moca_layernorm_f16(None, None, None, None, 4, 64, 1e-5, None) == -1
    assert l.moca_temporal_attention_f16(C.c_void_p(8), C.c_void_p(8), C.c_void_p(8), C.c_void_p(8), 1, 17, 4, 1, 192, 64, 0.125, None) == -1
    assert l.moca_fifo_step_windows_f32(C.byref(lib.FifoStepParams()), None) == -1
    assert l.moca_sam_select_masks_f32(None, None, None, None, None, None, 4, 8, 256, None) == -1
    assert l.moca_fifo_gather_windows_f32(C.c_void_p(8), C.c_void_p(8), None, None, None, 0, 1, 4, 20, 8, 256, None) == -1   # neither x nor anchor
    assert l.moca_fifo_randn_f32(None, None, 16, None) == -1 and l.moca_repeat_f16(C.c_void_p(16), C.c_void_p(32), 24, 2, None) == -1
    assert l.moca_set_tuning(99, 1) == -1 and l.moca_set_tuning(lib.MOCA_TUNE_GEMM_G4, 1) == 1
    assert l.moca_gemm_splitk_ws_bytes(640, 1280, 4) == 4 * 640 * 1280 * 4
    assert l.moca_groupnorm_ws_bytes(16, 2560, 320) > 0
    with pytest.raises(lib.MocaHipError):
        lib.check(-1, "x")


def test_unet_state_dict_surface_matches_reference_counts():
    """1484 tensors / 1 413 284 420 parameters, typo'd `temopral_conv` key included."""
    from moca_video_amd import UNetModel
    from helpers import FULL
    m = UNetModel(**FULL)
    sd = m.state_dict()
    assert len(sd) == 1484 and sum(v.numel() for v in sd.values()) == 1413284420
    expect = {"time_embed.0.weight": (1280, 320), "input_blocks.0.0.weight": (320, 4, 3, 3),
              "input_blocks.1.0.temopral_conv.conv1.2.weight": (320, 320, 3, 1, 1),
              "input_blocks.1.1.transformer_blocks.0.attn2.to_k.weight": (320, 1024),
              "input_blocks.1.1.transformer_blocks.0.ff.net.0.proj.weight": (2560, 320),
              "init_attn.0.proj_in.weight": (512, 320, 1), "out.2.weight": (4, 320, 3, 3),
              "middle_block.1.proj_out.bias": (1280,), "output_blocks.2.1.conv.weight": (1280, 1280, 3, 3),
              "input_blocks.3.0.op.weight": (320, 320, 3, 3), "output_blocks.11.2.norm.weight": (320,)}
    for k, shp in expect.items():
        assert tuple(sd[k].shape) == shp, k
    sig = inspect.signature(m.forward)
    assert list(sig.parameters)[:5] == ["x", "timesteps", "context", "features_adapter", "fps"]
    with pytest.raises(NotImplementedError):
        UNetModel(**{**FULL, "use_relative_position": True})
    with pytest.raises(AssertionError):
        UNetModel(in_channels=4, model_channels=64, out_channels=4, num_res_blocks=1, attention_resolutions=[1])


def test_yaml_target_resolves_to_hip_unet():
    import yaml
    from moca_video_amd import DiffusionWrapper, UNetModel, instantiate_from_config
    cfg = {"target": "lvdm.modules.networks.openaimodel3d.UNetModel",
           "params": yaml.safe_load("{in_channels: 4, out_channels: 4, model_channels: 64, attention_resolutions: [1], "
                                    "num_res_blocks: 1, channel_mult: [1], num_head_channels: 64, context_dim: 64, use_linear: true, "
                                    "temporal_conv: true, use_relative_position: false, temporal_length: 16}")}
    assert isinstance(instantiate_from_config(cfg), UNetModel)
    assert isinstance(DiffusionWrapper(cfg, "crossattn").diffusion_model, UNetModel)
    with pytest.raises(KeyError):
        instantiate_from_config({"params": {}})


def test_pack_layouts_on_cpu():
    """weight pre-packing is plain tensor reshuffling and can be checked without a GPU."""
    from moca_video_amd import ops
    w = torch.randn(16, 5, 3, 3)
    p = ops.pack_conv3x3(w, torch.randn(16), cpad=8, device="cpu")
    assert p.w.shape == (64, 128) and p.K == 72 and p.N == 64
    assert torch.equal(p.w[:16, :72].view(16, 3, 3, 8)[..., :5], w.permute(0, 2, 3, 1).half())
    assert (p.w[16:] == 0).all() and (p.w[:, 72:] == 0).all()
    wt = torch.randn(8, 4 * 2, 3, 1, 1)
    pt = ops.pack_tconv3(wt, None, device="cpu")
    assert torch.equal(pt.w[:8, :24].view(8, 3, 8), wt[..., 0, 0].permute(0, 2, 1).half())
    wg, bg = torch.randn(128, 16), torch.randn(128)
    pg = ops.pack_geglu(wg, bg, device="cpu")
    assert pg.geglu and pg.n_out == 64 and pg.N == 128
    assert torch.equal(pg.w[:32, :16], wg[:32].half()) and torch.equal(pg.w[32:64, :16], wg[64:96].half())
    assert torch.equal(pg.w[64:96, :16], wg[32:64].half()) and torch.equal(pg.bias[32:64], bg[64:96])
    pc = ops.pack_linear_cat([torch.randn(64, 32), torch.randn(64, 32), torch.randn(64, 32)], device="cpu")
    assert pc.w.shape == (192, 64)


def test_layernorm_fold_algebra_cpu():
    """ops.fold_layernorm (host side of MOCA_EP_LNFOLD): Linear(LayerNorm(x)) == rstd * (x W'^T - mean * rowsum(W')) + b' with
    W' = W diag(gamma), b' = b + W beta -- the epilogue formula of the consumer kernels, restated in torch on the CPU."""
    import torch
    from moca_video_amd import ops
    g = torch.Generator().manual_seed(3)
    x = torch.randn(50, 96, generator=g) * 2.0 + 0.7
    w, b = torch.randn(40, 96, generator=g) * 0.1, torch.randn(40, generator=g)
    gamma, beta = torch.rand(96, generator=g) + 0.5, torch.randn(96, generator=g) * 0.2
    wf, bf = ops.fold_layernorm(w, b, gamma, beta)
    mean = x.mean(1, keepdim=True)
    var = (x * x).mean(1, keepdim=True) - mean * mean            # what the row partial sums give the kernel
    rstd = (var + 1e-5).rsqrt()
    got = rstd * (x @ wf.t() - mean * wf.sum(1)[None, :]) + bf
    ref = torch.nn.functional.layer_norm(x, (96,), gamma, beta, 1e-5) @ w.t() + b
    assert (got - ref).abs().max() < 1e-4 * ref.abs().max()


def test_split_k_factors_cpu():
    """plan.gemm_splits: the measured rules of the 50-tile level (M = N = 1280) and the untouched small / full launches"""
    import types
    import torch
    from moca_video_amd.plan import gemm_splits
    pw = lambda n, k: types.SimpleNamespace(N=n, geglu=False, w=torch.empty(0, k))
    assert gemm_splits(1280, pw(1280, 1280)) == 1          # unsplit: slabs + reduce launch cost more than the idle CUs
    assert gemm_splits(1280, pw(1280, 3840)) == 4
    assert gemm_splits(1280, pw(1280, 5120)) == 4
    assert gemm_splits(1280, pw(1280, 11520)) == 5         # the 3x3 convs want every CU
    assert gemm_splits(81920, pw(320, 320)) == 1
    assert gemm_splits(5120, pw(1280, 1280)) == 1


def test_upsample_phase_algebra_cpu():
    """ops.pack_upconv_phases (host side of moca_gemm_params.up_phase): nearest x2 + conv3x3 (openaimodel3d.py:96-106) == four 2 x 2
    convs on the low-resolution grid, phase (a, b) producing output pixels (2i + a, 2j + b) from input pixels (i + a - 1 + r,
    j + b - 1 + s), r, s in {0, 1}, with zero padding -- restated with F.conv2d on the CPU from the PACKED rows [N][(r, s, c)]."""
    import torch.nn.functional as F
    from moca_video_amd import ops
    g = torch.Generator().manual_seed(11)
    x = torch.randn(2, 8, 5, 7, generator=g)
    w = (torch.randn(64, 8, 3, 3, generator=g) * 0.2).half().float()      # fp16-representable weights, fp32 sums stay exact enough
    b = torch.randn(64, generator=g)
    ref = F.conv2d(F.interpolate(x, scale_factor=2, mode="nearest"), w, b, padding=1)
    phases = ops.pack_upconv_phases(w, b, device="cpu")
    assert len(phases) == 4 and all(p.K == 32 and p.N == 64 for p in phases)
    out = torch.zeros_like(ref)
    for ph, pw in enumerate(phases):
        a, bb = ph >> 1, ph & 1
        k = pw.w[:64, :32].float().view(64, 2, 2, 8).permute(0, 3, 1, 2)   # [N][C][r][s]
        xp = F.pad(x, (1 - bb, bb, 1 - a, a))                              # input rows i + a - 1 .. i + a, columns j + b - 1 .. j + b
        out[:, :, a::2, bb::2] = F.conv2d(xp, k, pw.bias[:64])
    assert (out - ref).abs().max() < 2e-3 * ref.abs().max()                # (the summed weight pairs are rounded to fp16 once)


def test_product_library_is_not_a_diagnostic_build():
    """Diagnostic builds (stamps, timing-only variants with wrong results) export the whole API; they report "DIAG:<name>" in
    moca_version() and lib.load() refuses them unless MOCA_HIP_DIAG=1.  Every diagnostic target of the Makefile must set the name."""
    import os
    import re
    from moca_video_amd import lib
    assert "DIAG:" not in lib.version()
    mk = open(os.path.join(os.path.dirname(lib.LIB_PATH), "csrc", "Makefile")).read()
    for target in ("stamps", "gndiag", "diagx"):
        body = mk[mk.index(f"\n{target}:"):].split("\n\n")[0]
        compile_lines = [l for l in body.splitlines() if "-c gemm.hip" in l]
        assert compile_lines and all("-DMOCA_DIAG_NAME=" in l for l in compile_lines), target
    src = open(os.path.join(os.path.dirname(lib.LIB_PATH), "lib.py")).read()
    assert re.search(r'"DIAG:" in ver and os\.environ\.get\("MOCA_HIP_DIAG"\) != "1"', src)
