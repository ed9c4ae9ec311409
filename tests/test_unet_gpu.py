"""GPU parity of the HIP UNet (through the drop-in UNetModel / DiffusionWrapper boundary, i.e.
through the C-ABI) against (a) golden outputs of the real reference and (b) the CPU oracle on
the same seeded inputs.  fp16 storage / fp32 accumulate vs the fp32 reference:
tolerance = 2e-2 * max|ref| on the UNet output (stated fp16 tolerance), 6e-3 per block."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from helpers import REDUCED, golden, inp, relerr, state_dict_for  # noqa: E402

TOL_UNET = 2e-2
TOL_BLOCK = 6e-3


@pytest.fixture(scope="module")
def reduced_model():
    from moca_video_amd import UNetModel
    m = UNetModel(**REDUCED)
    m.load_state_dict(state_dict_for(m, 11), strict=True)
    return m.cuda()


@pytest.mark.parametrize("case,B", [("uniform", 1), ("fifo", 1), ("batch2", 2)])
def test_unet_reduced_vs_reference_golden(reduced_model, case, B):
    g = golden("unet_reduced")
    L = int(g[case + "__L"])
    x = inp(f"reduced.{case}.x", (B, 4, 8, 16, 16)).cuda()
    ctx = inp(f"reduced.{case}.ctx", (B, L, 128)).cuda()
    t = torch.from_numpy(g[case + "__t"]).cuda()
    fps = g[case + "__fps"]
    fps = int(fps) if fps.ndim == 0 else torch.from_numpy(fps).cuda()
    ref = torch.from_numpy(g[case])
    for it in range(3):     # eager pass, graph-capture pass, graph replay: all must agree
        y = reduced_model(x, t, context=ctx, fps=fps, clean_cond=True, gamma=0.5)
        assert y.shape == ref.shape and y.dtype == x.dtype
        e = relerr(y.cpu(), ref)
        assert e < TOL_UNET, f"{case} pass {it}: rel err {e:.3e}"
    plan = next(iter(reduced_model._plans.values()))
    assert any(p.graph is not None for p in reduced_model._plans.values()), "hipGraph replay path was not taken"


def test_unet_reduced_vs_oracle_fresh_inputs(reduced_model):
    """inputs that no golden holds: HIP vs the CPU oracle computed here."""
    from oracle import unet_oracle as UO
    sd = state_dict_for(reduced_model, 11)
    x = inp("fresh.x", (1, 4, 16, 8, 24))
    ctx = inp("fresh.ctx", (1, 154, 128))
    t = torch.arange(0, 960, 60, dtype=torch.long)
    ref = UO.unet_forward(sd, x, t, ctx, fps=torch.tensor([10]))
    y = reduced_model(x.cuda(), t.cuda(), context=ctx.cuda(), fps=torch.tensor([10]).cuda())
    assert relerr(y.cpu(), ref) < TOL_UNET


def test_diffusion_wrapper_boundary():
    """DiffusionWrapper.forward(x, t, c_crossattn=[a, b]) concatenates the contexts (ddpm3d.py:711)."""
    from moca_video_amd import DenoiseModel
    dm = DenoiseModel({"target": "lvdm.modules.networks.openaimodel3d.UNetModel", "params": REDUCED})
    unet = dm.model.diffusion_model
    unet.load_state_dict(state_dict_for(unet, 11), strict=True)
    dm = dm.cuda()
    g = golden("unet_reduced")
    x = inp("reduced.fifo.x", (1, 4, 8, 16, 16)).cuda()
    ctx = inp("reduced.fifo.ctx", (1, 154, 128)).cuda()
    t = torch.from_numpy(g["fifo__t"]).cuda()
    cond = {"c_crossattn": [ctx[:, :77], ctx[:, 77:]], "fps": torch.tensor([10]).cuda()}
    y = dm.apply_model(x, t, cond, clean_cond=True)
    assert relerr(y.cpu(), torch.from_numpy(g["fifo"])) < TOL_UNET


def test_unet_rejects_cpu_and_bad_shapes(reduced_model):
    x = inp("reduced.uniform.x", (1, 4, 8, 16, 16))
    with pytest.raises(ValueError):
        reduced_model(x, torch.tensor([1]), context=torch.zeros(1, 77, 128))          # CPU tensor: no CPU path
    with pytest.raises(ValueError):
        reduced_model(x.cuda(), torch.tensor([1, 2, 3]).cuda(), context=torch.zeros(1, 77, 128).cuda())
