"""GPU parity of the fp32 sampler / FreeInit / FIFO-queue HIP path (through the DDIMSampler /
freq_mix_3d / shift_latents mirrors, i.e. through the C-ABI) against golden outputs of the real
reference.  fp32 arithmetic with FMA contraction off: tolerance 2e-6 relative (a couple of ulp;
the GPU's division/sqrt are correctly rounded, only summation order differs in the DFT: 2e-5)."""
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from helpers import golden, inp, relerr  # noqa: E402
from test_oracle_sampler import FI_FILTERS, FI_SHAPES, ddim_step_inputs, sam_candidates, sam_golden  # noqa: E402

TOL = 2e-6


class _Eps:
    """stands in for the UNet: returns queued eps tensors (the sampler arithmetic is what is under test)"""

    def __init__(self):
        self.q = []


def _sampler(S):
    from moca_video_amd.sampler import DDIMSampler
    from moca_video_amd.wrapper import DenoiseModel

    class M(torch.nn.Module):
        pass
    dm = DenoiseModel.__new__(DenoiseModel)
    torch.nn.Module.__init__(dm)
    # schedule buffers only (no UNet needed for these tests)
    from moca_video_amd.wrapper import make_beta_schedule
    betas = make_beta_schedule("linear", 1000, linear_start=0.00085, linear_end=0.012)
    ac = np.cumprod(1. - betas, axis=0)
    dm.num_timesteps = 1000
    dm.register_buffer("betas", torch.tensor(betas, dtype=torch.float32))
    dm.register_buffer("alphas_cumprod", torch.tensor(ac, dtype=torch.float32))
    dm.use_scale = True
    dm.register_buffer("scale_arr", torch.tensor(np.concatenate((np.linspace(1, 0.7, 400), np.full(1000, 0.7))), dtype=torch.float32))
    dm = dm.cuda()
    q = []
    dm.apply_model = lambda x, t, c, **kw: q.pop(0)
    s = DDIMSampler(dm)
    s.make_schedule(S, ddim_eta=1.0, verbose=False)
    return s, q


def test_schedule_matches_reference():
    g = golden("sampler_schedule")
    for S in (10, 50, 64):
        s, _ = _sampler(S)
        for k in ("ddim_timesteps", "ddim_sigmas", "ddim_alphas", "ddim_alphas_prev", "ddim_sqrt_one_minus_alphas",
                  "ddim_scale_arr", "ddim_scale_arr_prev"):
            np.testing.assert_array_equal(np.asarray(getattr(s, k)), g[f"S{S}_{k}"], err_msg=f"S{S} {k}")


def test_p_sample_ddim_cfg():
    g = golden("sampler_p_sample_ddim")
    s, q = _sampler(50)
    shape = (1, 4, 8, 16, 24)
    for index in (49, 20, 0):
        x = inp(f"ps.x{index}", shape).cuda()
        q[:] = [inp(f"ps.ec{index}", shape).cuda(), inp(f"ps.eu{index}", shape).cuda()]
        t = torch.full((1,), int(s.ddim_timesteps[index]), dtype=torch.long, device="cuda")
        xp, p0 = s.p_sample_ddim(x, [None], t, index, unconditional_guidance_scale=12.0, unconditional_conditioning=[None],
                                 noise=inp(f"ps.nz{index}", shape).cuda())
        assert relerr(xp.cpu(), g[f"i{index}_x_prev"]) < TOL
        assert relerr(p0.cpu(), g[f"i{index}_pred_x0"]) < TOL


@pytest.mark.parametrize("tag,dims", [("small", (4, 6, 16, 16)), ("cfgN", (4, 16, 40, 64))])
def test_ddim_step_moca(tag, dims):
    g = golden(f"sampler_ddim_step_{tag}")
    s, _ = _sampler(64)
    C, F, H, W = dims
    for call in (0, 1):
        x, e, noises, cond, mask = ddim_step_inputs(tag, C, F, H, W, call)
        ts = torch.from_numpy(g[f"c{call}_ts"]).long().cuda()
        xp, p0 = s.ddim_step(x.cuda(), e.cuda(), g[f"c{call}_indices"], cond.cuda(), None, ts, use_self_attention=True,
                             davis_masks=mask.cuda(), noise=torch.cat(noises, 2).cuda())
        assert relerr(xp.cpu(), g[f"c{call}_x_prev"]) < TOL
        assert relerr(p0.cpu(), g[f"c{call}_pred_x0"]) < TOL
        assert relerr(s.momentum.cpu(), g[f"c{call}_momentum"]) < TOL


def test_freeinit_filters_and_mix():
    from moca_video_amd.freeinit import freq_mix_3d, get_freq_filter
    g = golden("freeinit")
    for shp in FI_SHAPES:
        tag = "x".join(map(str, shp[2:]))
        for ft, ds, dt in FI_FILTERS:
            lpf = get_freq_filter(shp, "cuda", ft, 4, ds, dt)
            assert tuple(lpf.shape) == shp
            ref = torch.from_numpy(g[f"{tag}_{ft}_{ds}_{dt}_lpf"])
            assert (lpf[0, 0].cpu() - ref).abs().max().item() <= 1.2e-7 * max(ref.abs().max().item(), 1e-30), f"{tag} {ft}"
            mix = freq_mix_3d(inp(f"fi.x.{tag}", shp).cuda(), inp(f"fi.n.{tag}", shp).cuda(), lpf)
            assert tuple(mix.shape) == shp
            assert relerr(mix.cpu(), g[f"{tag}_{ft}_{ds}_{dt}_mix"]) < 2e-5, f"{tag} {ft}"


def test_fifo_queue():
    from moca_video_amd.fifo import prepare_latents, shift_latents
    g = golden("fifo_queue")
    s, _ = _sampler(64)
    args = types.SimpleNamespace(num_inference_steps=64, video_length=16, lookahead_denoising=True)
    z = inp("fifo.z", (1, 4, 16, 8, 12)).cuda()
    noises = [inp(f"fifo.nz{i}", (1, 4, 1, 8, 12)) for i in range(72)]
    lat = prepare_latents(args, None, s, initial_latents=z, noises=noises)
    assert relerr(lat.cpu(), g["prepared"]) < TOL
    sh = shift_latents(lat.clone(), noise=inp("fifo.nz72", (1, 4, 8, 12)))
    assert relerr(sh.cpu(), g["shifted"]) < 2e-5
    assert torch.equal(sh[:, :, :-1], lat[:, :, 1:])


def test_freeinit_linearity_and_identity():
    """size-independent properties at the headline shape: LPF=1 returns x, LPF=0 returns the noise,
    and the mix is linear in (x, noise)."""
    from moca_video_amd.freeinit import freq_mix_3d
    shp = (1, 4, 16, 40, 64)
    x, n = inp("prop.x", shp).cuda(), inp("prop.n", shp).cuda()
    one, zero = torch.ones(shp, device="cuda"), torch.zeros(shp, device="cuda")
    assert relerr(freq_mix_3d(x, n, one), x) < 2e-5
    assert relerr(freq_mix_3d(x, n, zero), n) < 2e-5
    lpf = torch.rand(16, 40, 64, device="cuda").expand(shp)
    a = freq_mix_3d(x, n, lpf); b = freq_mix_3d(2 * x, 2 * n, lpf)
    assert relerr(b, 2 * a) < 2e-5


def _sam_candidates(F, H, W, seq):
    """per-frame candidate masks exercising every rule of `_apply_segmentation` (with the expected "pred_x0 changed" flags).
    seq 0: no detection before any mask; a detection; a close one (taken); a deviating one (IoU < 0.5 -> previous reused);
           no detection (-> previous); an oversized mask first (skipped) then a small one; a re-ordered pair (IoU over the
           zipped pairs < 0.5 -> previous pair reused).
    seq 1: a small mask followed by an oversized one (wipes it: no injection, but it becomes the previous set); no detection
           (-> that set again: still nothing); a close single mask (IoU with the first of the previous pair high -> taken)."""
    def rect(y0, y1, x0, x1):
        m = torch.zeros(H, W)
        m[y0:y1, x0:x1] = 1.0
        return m
    big = torch.ones(H, W)
    if seq == 0:
        cands = [None, rect(2, 10, 3, 12)[None], rect(2, 10, 4, 12)[None], rect(11, 15, 0, 5)[None], None,
                 torch.stack([big, rect(1, 5, 1, 6)]), torch.stack([rect(1, 5, 1, 6), big])]
        changed = [False, True, True, True, True, True, True]
    else:
        cands = [torch.stack([rect(2, 10, 3, 12), big]), None, rect(2, 10, 4, 12)[None]]
        changed = [False, False, True]
    # frames past the list have no detection -> the previous masks are reused (:790-793): they keep being injected
    return (cands + [None] * F)[:F], (changed + [True] * F)[:F]


@pytest.mark.parametrize("key,seq", [("seq0_low", 0), ("seq1_low", 1), ("seq0_high", 0)])
def test_ddim_step_sam_branch_vs_reference_golden(key, seq):
    """HIP `ddim_step(sam_masks=...)` against the REAL reference's segmentation branch (ddim.py:592-606,739-903) run with fake
    Grounded-SAM-2 objects returning the same scripted candidates (tests/golden/sampler_ddim_step_sam.npz): IoU fallback,
    > 80 % reset, factor-2 injection, t <= 300 gate.  (The reference returns an injected frame replicated C times along the
    frame axis; `sam_golden` checks the copies are identical and folds them.)"""
    C, F, H, W = 4, 8, 16, 16
    xp_g, p0_g, idx, tsn = sam_golden(key, F)
    s, _ = _sampler(64)
    x, e = inp("ds.small.x0", (1, C, F, H, W)), inp("ds.small.e0", (1, C, F, H, W))
    noises = [inp(f"ds.small.nz0.{i}", (1, C, 1, H, W)) for i in range(F)]
    cond = (inp("ds.small.cond", (1, C, 1, H, W)) * 0.25 + 0.5).clamp(0, 1)
    ts = torch.from_numpy(tsn).long()
    cands = sam_candidates(F, H, W, seq)
    xp, p0 = s.ddim_step(x.cuda(), e.cuda(), idx, cond.cuda(), "object.", ts.cuda(), noise=torch.cat(noises, 2).cuda(),
                         sam_masks=[None if c is None else c.cuda() for c in cands])
    assert relerr(xp.cpu(), xp_g) < TOL
    assert relerr(p0.cpu(), p0_g) < TOL


@pytest.mark.parametrize("seq", [0, 1])
def test_ddim_step_sam_mask_branch_vs_oracle(seq):
    """The segmentation branch of ddim_step (ddim.py:592-606, 739-903) on precomputed candidate masks: HIP path vs the
    oracle's restatement (itself pinned bit-exactly to the real reference by test_oracle_sampler.py::
    test_ddim_step_sam_branch_vs_reference) plus the expected "which frames changed" pattern."""
    from oracle import sampler_oracle as SO
    from test_oracle_sampler import BUF
    s, _ = _sampler(64)
    sch = SO.make_schedule(BUF, 64, 1.0)
    C, F, H, W = 4, 8, 16, 16
    x, e, noises, cond, _ = ddim_step_inputs("small", C, F, H, W, 0)
    cands, expect_changed = _sam_candidates(F, H, W, seq)
    indices = np.arange(3, 3 + F)                                      # low timesteps: every frame has t <= 300
    ts = torch.as_tensor(np.asarray(s.ddim_timesteps)[indices]).long()
    assert int(ts.max()) <= 300
    mom = torch.zeros(1, C, F, H, W)
    xp_ref, p0_ref = SO.ddim_step(sch, x, e, indices, cond[:, :, 0], ts, noises, mom, sam_masks=cands)
    xp, p0 = s.ddim_step(x.cuda(), e.cuda(), indices, cond.cuda(), "object.", ts.cuda(), noise=torch.cat(noises, 2).cuda(),
                         sam_masks=[None if c is None else c.cuda() for c in cands])
    assert relerr(xp.cpu(), xp_ref) < TOL
    assert relerr(p0.cpu(), p0_ref) < TOL
    s2, _ = _sampler(64)
    _, p0_plain = s2.ddim_step(x.cuda(), e.cuda(), indices, cond.cuda(), "object.", ts.cuda(), noise=torch.cat(noises, 2).cuda())
    changed = [(p0[:, :, i] - p0_plain[:, :, i]).abs().max().item() > 1e-6 for i in range(F)]
    assert changed == expect_changed, changed
    # frames above t = 300 are never touched by this branch (:592)
    hi = np.arange(40, 40 + F)
    ts_hi = torch.as_tensor(np.asarray(s.ddim_timesteps)[hi]).long()
    assert int(ts_hi.min()) > 300
    s3, s4 = _sampler(64)[0], _sampler(64)[0]
    _, a = s3.ddim_step(x.cuda(), e.cuda(), hi, cond.cuda(), "object.", ts_hi.cuda(), noise=torch.cat(noises, 2).cuda(),
                        sam_masks=[None if c is None else c.cuda() for c in cands])
    _, b = s4.ddim_step(x.cuda(), e.cuda(), hi, cond.cuda(), "object.", ts_hi.cuda(), noise=torch.cat(noises, 2).cuda())
    assert torch.equal(a, b)
