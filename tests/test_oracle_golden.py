"""CPU: the oracle (oracle/*.py, our restatement) against golden outputs of the REAL reference
(tests/golden/*.npz, produced by tools/make_golden.py from /root/reference).  fp32 vs fp32:
tolerance 2e-5 relative to the output's max magnitude (different but equivalent op orderings)."""
import numpy as np
import pytest
import torch

from helpers import REDUCED, golden, inp, relerr, state_dict_for
from oracle import unet_oracle as UO

TOL = 2e-5


def _sd(shapes, seed):
    from moca_video_amd.weightgen import gen_state_dict
    return gen_state_dict(shapes, seed)


def _unet_skeleton(params):
    from moca_video_amd.unet import UNetModel
    return UNetModel(**params)


def test_timestep_embedding():
    g = golden("timestep_embedding")
    y = UO.timestep_embedding(torch.from_numpy(g["t"]), 320)
    assert relerr(y, g["y"]) < 1e-6


def _block_sd(mod_factory, seed):
    m = mod_factory()
    return state_dict_for(m, seed)


def test_resblock():
    from moca_video_amd.unet import _ResBlock
    for name, cout, seed in (("block_resblock", 128, 1), ("block_resblock_same", 64, 2)):
        sd = {"rb." + k: v for k, v in state_dict_for(_ResBlock(64, 256, cout, True), seed).items()}
        x, emb = inp("rb.x", (8, 64, 6, 10)), inp("rb.emb", (8, 256))
        y = UO.res_block(sd, "rb", x, emb, 2)
        assert relerr(y, golden(name)["y"]) < TOL


def test_spatial_transformer():
    from moca_video_amd.unet import _SpatialTransformer
    sd = {"st." + k: v for k, v in state_dict_for(_SpatialTransformer(128, 2, 64, 1, 96, True), 3).items()}
    y = UO.spatial_transformer(sd, "st", inp("st.x", (4, 128, 6, 10)), inp("st.ctx", (4, 77, 96)), 2)
    assert relerr(y, golden("block_spatial_transformer")["y"]) < TOL


def test_temporal_transformers():
    from moca_video_amd.unet import _TemporalTransformer
    sd = {"tt." + k: v for k, v in state_dict_for(_TemporalTransformer(128, 2, 64, 1, True), 4).items()}
    y = UO.temporal_transformer(sd, "tt", inp("tt.x", (2, 128, 8, 3, 5)), 2)
    assert relerr(y, golden("block_temporal_transformer")["y"]) < TOL
    sd = {"ti." + k: v for k, v in state_dict_for(_TemporalTransformer(64, 8, 64, 1, False), 5).items()}
    y = UO.temporal_transformer(sd, "ti", inp("ti.x", (1, 64, 16, 3, 5)), 8)
    assert relerr(y, golden("block_init_attn")["y"]) < TOL


def test_down_up():
    from moca_video_amd.unet import _Downsample, _Upsample
    g = golden("block_down_up")
    x = inp("ud.x", (3, 64, 6, 10))
    sd = {"b.0." + k: v for k, v in state_dict_for(_Downsample(64), 6).items()}
    assert relerr(UO._run_sequential(sd, "b", x, None, None, 1, 64), g["down"]) < TOL
    sd = {"b.0." + k: v for k, v in state_dict_for(_Upsample(64), 7).items()}
    assert relerr(UO._run_sequential(sd, "b", x, None, None, 1, 64), g["up"]) < TOL


@pytest.mark.parametrize("case,B", [("uniform", 1), ("fifo", 1), ("batch2", 2)])
def test_unet_reduced(case, B):
    g = golden("unet_reduced")
    sd = state_dict_for(_unet_skeleton(REDUCED), 11)
    L = int(g[case + "__L"])
    x = inp(f"reduced.{case}.x", (B, 4, 8, 16, 16))
    ctx = inp(f"reduced.{case}.ctx", (B, L, 128))
    t = torch.from_numpy(g[case + "__t"])
    fps = g[case + "__fps"]
    fps = int(fps) if fps.ndim == 0 else torch.from_numpy(fps)
    y = UO.unet_forward(sd, x, t, ctx, fps=fps)
    assert y.shape == g[case].shape
    assert relerr(y, g[case]) < TOL


def test_unet_full_width_cfg0():
    """The oracle at the full 1.41 B-parameter width against the REAL reference UNet's output on the config[0] shape
    (`tools/make_golden.py --only full`); ~1.5 min: 46 s of weight generation + two forwards."""
    import os
    from helpers import FULL, GOLD
    if not os.path.exists(os.path.join(GOLD, "unet_full.npz")):
        pytest.skip("full-width goldens not generated")
    g = golden("unet_full")
    sd = state_dict_for(_unet_skeleton(FULL), 11)
    for case in ("cfg0_uniform", "cfg0_fifo"):
        L = int(g[case + "__L"])
        x = inp(f"full.{case}.x", (1, 4, 8, 32, 32))
        ctx = inp(f"full.{case}.ctx", (1, L, 1024))
        y = UO.unet_forward(sd, x, torch.from_numpy(g[case + "__t"]), ctx, fps=torch.from_numpy(np.atleast_1d(g[case + "__fps"])))
        assert relerr(y, g[case]) < TOL, case
