"""CPU: sampler / FreeInit / FIFO-queue oracles against golden outputs of the real reference."""
import numpy as np
import pytest
import torch

from helpers import golden, inp, relerr
from oracle import freeinit_oracle as FO
from oracle import sampler_oracle as SO

BUF = SO.ddpm_buffers()


def test_make_schedule_exact():
    g = golden("sampler_schedule")
    for S in (10, 50, 64):
        sch = SO.make_schedule(BUF, S, 1.0)
        for k, v in sch.items():
            np.testing.assert_array_equal(np.asarray(v), g[f"S{S}_{k}"], err_msg=f"S{S} {k}")


def test_p_sample_ddim():
    g = golden("sampler_p_sample_ddim")
    sch = SO.make_schedule(BUF, 50, 1.0)
    shape = (1, 4, 8, 16, 24)
    for index in (49, 20, 0):
        xp, p0 = SO.p_sample_ddim(sch, inp(f"ps.x{index}", shape), inp(f"ps.ec{index}", shape), inp(f"ps.eu{index}", shape),
                                  12.0, index, inp(f"ps.nz{index}", shape))
        assert torch.equal(xp, torch.from_numpy(g[f"i{index}_x_prev"]))
        assert torch.equal(p0, torch.from_numpy(g[f"i{index}_pred_x0"]))


def ddim_step_inputs(tag, C, F, H, W, call):
    shape = (1, C, F, H, W)
    x, e = inp(f"ds.{tag}.x{call}", shape), inp(f"ds.{tag}.e{call}", shape)
    noises = [inp(f"ds.{tag}.nz{call}.{i}", (1, C, 1, H, W)) for i in range(F)]
    cond = (inp(f"ds.{tag}.cond", (1, C, 1, H, W)) * 0.25 + 0.5).clamp(0, 1)
    mask = (inp(f"ds.{tag}.mask", (1, 1, F, H, W)) > 0.5).float()
    mask[:, :, 1] = 0.0
    return x, e, noises, cond, mask


@pytest.mark.parametrize("tag,dims", [("small", (4, 6, 16, 16)), ("cfgN", (4, 16, 40, 64))])
def test_ddim_step(tag, dims):
    g = golden(f"sampler_ddim_step_{tag}")
    sch = SO.make_schedule(BUF, 64, 1.0)
    C, F, H, W = dims
    momentum = torch.zeros(1, C, F, H, W)
    for call in (0, 1):
        x, e, noises, cond, mask = ddim_step_inputs(tag, C, F, H, W, call)
        ts = torch.from_numpy(g[f"c{call}_ts"]).long()
        xp, p0 = SO.ddim_step(sch, x, e, g[f"c{call}_indices"], cond, ts, noises, momentum, davis_masks=mask)
        assert torch.equal(xp, torch.from_numpy(g[f"c{call}_x_prev"]))
        assert torch.equal(p0, torch.from_numpy(g[f"c{call}_pred_x0"]))
        assert torch.equal(momentum, torch.from_numpy(g[f"c{call}_momentum"]))


def sam_candidates(F, H, W, seq):
    """The scripted Grounded-SAM-2 outputs tools/make_golden.py::sam_candidates fed to the REAL `_apply_segmentation`
    (same lists, rebuilt here: a fixture holds outputs only).  One entry per frame, None = no box detected."""
    def rect(y0, y1, x0, x1):
        m = torch.zeros(H, W)
        m[y0:y1, x0:x1] = 1.0
        return m
    big = torch.ones(H, W)
    if seq == 0:
        cands = [None, rect(2, 10, 3, 12)[None], rect(2, 10, 4, 12)[None], rect(11, 15, 0, 5)[None], None,
                 torch.stack([big, rect(1, 5, 1, 6)]), torch.stack([rect(1, 5, 1, 6), big])]
    else:
        cands = [torch.stack([rect(2, 10, 3, 12), big]), None, rect(2, 10, 4, 12)[None]]
    return (cands + [None] * F)[:F]


def sam_golden(key, F):
    """(x_prev, pred_x0 [1,C,F,H,W], indices, ts) of tests/golden/sampler_ddim_step_sam.npz.  The reference's `torch.where`
    broadcasts a [1,C,H,W] mask against the 5-D pred_x0 (ddim.py:897-901), so a frame that went through an injection comes
    back REPLICATED C times along the frame axis (`frames_out` records 1 or C per frame that entered the branch); the copies
    are asserted identical and folded to one frame here."""
    g = golden("sampler_ddim_step_sam")
    p0 = torch.from_numpy(g[key + "_pred_x0"])
    counts = [int(c) for c in g[key + "_frames_out"]]
    if counts == [0]:                       # t > 300 everywhere: the branch was never entered
        counts = []
    counts = counts + [1] * (F - len(counts))
    assert sum(counts) == p0.shape[2]
    frames, k = [], 0
    for c in counts:
        blk = p0[:, :, k:k + c]
        for j in range(1, c):
            assert torch.equal(blk[:, :, j], blk[:, :, 0])
        frames.append(blk[:, :, :1])
        k += c
    return torch.from_numpy(g[key + "_x_prev"]), torch.cat(frames, 2), g[key + "_indices"], g[key + "_ts"]


@pytest.mark.parametrize("key,seq", [("seq0_low", 0), ("seq1_low", 1), ("seq0_high", 0)])
def test_ddim_step_sam_branch_vs_reference(key, seq):
    """segmentation branch (ddim.py:592-606,739-903): the oracle restatement against the REAL reference code driven by
    fake Grounded-SAM-2 objects (tools/make_golden.py::sampler_sam_cases) -- bit-exact"""
    C, F, H, W = 4, 8, 16, 16
    xp_g, p0_g, idx, tsn = sam_golden(key, F)
    sch = SO.make_schedule(BUF, 64, 1.0)
    x, e, noises, cond, _ = ddim_step_inputs("small", C, F, H, W, 0)
    noises = noises + [inp(f"ds.small.nz0.{i}", (1, C, 1, H, W)) for i in range(len(noises), F)]
    x, e = inp("ds.small.x0", (1, C, F, H, W)), inp("ds.small.e0", (1, C, F, H, W))
    mom = torch.zeros(1, C, F, H, W)
    xp, p0 = SO.ddim_step(sch, x, e, idx, cond, torch.from_numpy(tsn).long(), noises, mom, sam_masks=sam_candidates(F, H, W, seq))
    assert torch.equal(xp, xp_g)
    assert torch.equal(p0, p0_g)


FI_SHAPES = ((1, 4, 1, 40, 64), (1, 4, 16, 40, 64), (1, 2, 3, 5, 7), (1, 4, 8, 32, 32))
FI_FILTERS = (("gaussian", 0.25, 0.25), ("butterworth", 0.25, 0.25), ("ideal", 0.25, 0.25), ("box", 0.25, 0.25),
              ("gaussian", 0.3, 0.6), ("box", 0.5, 0.5))


def test_freeinit_filters_and_mix():
    g = golden("freeinit")
    for shp in FI_SHAPES:
        tag = "x".join(map(str, shp[2:]))
        for ft, ds, dt in FI_FILTERS:
            lpf = FO.get_freq_filter(shp, ft, 4, ds, dt)
            np.testing.assert_array_equal(lpf[0, 0].numpy(), g[f"{tag}_{ft}_{ds}_{dt}_lpf"], err_msg=f"{tag} {ft}")
            mix = FO.freq_mix_3d(inp(f"fi.x.{tag}", shp), inp(f"fi.n.{tag}", shp), lpf)
            assert relerr(mix, g[f"{tag}_{ft}_{ds}_{dt}_mix"]) < 1e-6


def test_fifo_queue():
    g = golden("fifo_queue")
    sch = SO.make_schedule(BUF, 64, 1.0)
    z = inp("fifo.z", (1, 4, 16, 8, 12))
    n_prep = int(g["n_noise_prepare"])
    assert n_prep == 72 and int(g["n_noise_total"]) == 73
    noises = [inp(f"fifo.nz{i}", (1, 4, 1, 8, 12)) for i in range(n_prep)]
    lat = FO.prepare_latents(z, sch["ddim_alphas"], 64, 16, True, noises)
    assert torch.equal(lat, torch.from_numpy(g["prepared"]))
    sh = FO.shift_latents(lat.clone(), inp("fifo.nz72", (1, 4, 8, 12)))
    assert relerr(sh, g["shifted"]) < 1e-6
    assert torch.equal(sh[:, :, :-1], lat[:, :, 1:])
