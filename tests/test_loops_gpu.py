"""GPU: the HIP sampling loops (moca_video_amd.fifo.base_ddim_sampling / fifo_ddim_sampling, host-driven and as one hipGraph per
iteration) against goldens of the REAL reference loops (tests/golden/loop_*.npz, tools/make_golden.py::loop_cases), and the
device-side FIFO kernels of csrc/fifo.hip against the per-window kernel / a numpy Philox.

Tolerances: the loops feed the fp16-storage UNet's output back in under CFG 12 (the guided eps is e_u + 12 (e_c - e_u): the
UNet's ~2.7e-3 relative error enters 12-fold).  Observed (gpurun_out/r3_errlog.txt): 10 base steps 1.15e-2 of max|ref| -> bound
1.7e-2; 3 FIFO iterations 2.2e-2 (DAVIS mode; 2.0e-2 prompt mode) -> bound 3e-2; bounds <= 1.5 x observed.
A wrong window order, write-back slice, emission index, coefficient or mask index gives O(1)."""
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from helpers import REDUCED, golden, inp, loop_sam_candidates, relerr, state_dict_for  # noqa: E402

TOL_BASE = 1.7e-2
TOL_FIFO = 3e-2
VAE_DD = dict(double_z=True, z_channels=4, resolution=512, in_channels=3, out_ch=3, ch=64, ch_mult=[1, 2, 4, 4],
              num_res_blocks=2, attn_resolutions=[], dropout=0.0)
FIFO_ARGS = types.SimpleNamespace(num_inference_steps=16, video_length=8, lookahead_denoising=True, num_partitions=2,
                                  new_video_length=10)


@pytest.fixture(scope="module")
def dm():
    from moca_video_amd import DenoiseModel
    m = DenoiseModel({"target": "lvdm.modules.networks.openaimodel3d.UNetModel", "params": REDUCED},
                     first_stage_config={"target": "lvdm.models.autoencoder.AutoencoderKL",
                                         "params": {"embed_dim": 4, "ddconfig": VAE_DD, "lossconfig": {"target": "torch.nn.Identity"}}},
                     scale_factor=0.18215)
    m.model.diffusion_model.load_state_dict(state_dict_for(m.model.diffusion_model, 11), strict=True)
    m.first_stage_model.load_state_dict(state_dict_for(m.first_stage_model, 5), strict=True)
    return m.cuda()


def _text():
    return {k: inp(n, (1, 77, 128)).cuda() for k, n in (("c1", "loop.ctx1"), ("c2", "loop.ctx2"), ("uc", "loop.uctx"))}


@pytest.mark.parametrize("use_graph", [False, True])
def test_hip_base_loop_vs_reference_golden(dm, tmp_path, use_graph):
    """base_ddim_sampling -> DDIMSampler.sample (funcs.py:177-241, ddim.py:109-252): 10 steps, eta 1, CFG 12, use_scale, latent
    cache files, decode -- against the REAL loop's outputs; host-issued steps and one hipGraph per step (fifo_graph.BaseEngine)"""
    from moca_video_amd.fifo import base_ddim_sampling
    g = golden("loop_base")
    t = _text()
    shape = [1, 4, 8, 16, 16]
    x_T = inp("loop.base.randn0", shape).cuda()
    noises = [inp(f"loop.base.noise_like{i}", shape).cuda() for i in range(10)]
    cond = {"c_crossattn": [t["c1"]], "fps": torch.tensor([10]).cuda()}
    images, sampler, samples = base_ddim_sampling(dm, cond, shape, 10, 1.0, 12.0, uc_emb=t["uc"], latents_dir=str(tmp_path), x_T=x_T,
                                                  noises=noises, use_graph=use_graph)
    assert torch.equal(torch.load(str(tmp_path / "0.pt")).cpu(), torch.from_numpy(g["pt0"]))
    e = relerr(samples.cpu(), g["samples"])
    assert e < TOL_BASE, f"samples rel err {e:.3e}"
    assert relerr(torch.load(str(tmp_path / "10.pt")).cpu(), g["ptN"]) < TOL_BASE
    e = relerr(images.cpu(), g["images"])
    assert e < TOL_BASE, f"decoded images rel err {e:.3e}"


def test_base_step_graph_equals_p_sample_ddim(dm):
    """fifo_graph.BaseEngine (timestep rows + shared-prefix UNet + guidance + DDIM update in one hipGraph, schedule index taken from
    the device iteration counter) against DDIMSampler.p_sample_ddim step by step on the same latents and draws, B = 2 prompts; then
    engine reuse (`reset`) and the device noise stream (same seed -> same latents, next seed -> different)"""
    from moca_video_amd.fifo_graph import BaseEngine
    from moca_video_amd.sampler import DDIMSampler
    t = _text()
    s = DDIMSampler(dm)
    s.make_schedule(6, ddim_eta=1.0, verbose=False)
    g = torch.Generator(device="cuda").manual_seed(3)
    x0 = torch.randn(2, 4, 8, 16, 16, device="cuda", generator=g)
    fps = torch.tensor([10, 12]).cuda()
    cond = {"c_crossattn": [torch.cat([t["c1"], t["c2"]])], "fps": fps}
    uc = {"c_crossattn": [t["uc"].expand(2, -1, -1)], "fps": fps}
    assert BaseEngine.supported(dm, x0, cond, uc, 12.0)
    eng = BaseEngine(dm, s, x0, cond, uc, 12.0, seed=5, keep_pred_x0=True)
    x = x0.clone()
    worst = 0.0
    for i in range(6):
        index = 5 - i
        n = torch.randn(x.shape, device="cuda", generator=g)
        ts = torch.full((2,), int(s.ddim_timesteps[index]), device="cuda", dtype=torch.long)
        x_ref, p_ref = s.p_sample_ddim(x, cond, ts, index=index, unconditional_guidance_scale=12.0, unconditional_conditioning=uc, noise=n)
        eng.step(noise=n)
        got = eng.latents()
        worst = max(worst, relerr(got.cpu(), x_ref.cpu()), relerr(eng.last_pred_x0().cpu(), p_ref.cpu()))
        x = got                                    # same trajectory for both: per-step comparison, no compounding
    assert worst < 1e-5, f"graph step vs p_sample_ddim {worst:.3e}"   # same UNet plan kernels; the update differs by fp32 rounding order

    def run(seed):
        eng.reset(x0, cond, uc, seed)
        for _ in range(6):
            eng.step()
        return eng.latents()
    a, b, c = run(9), run(9), run(10)
    assert torch.equal(a, b) and torch.isfinite(a).all()
    assert not torch.equal(a, c)
    assert eng.plan.graph is not None
    eng.close()


def _fifo_noises(mode):
    """the recorded draws of the golden run, regrouped the way fifo_ddim_sampling takes them"""
    k = {"randn_like": 0, "noise_like": 0, "randn": 0}

    def nxt(kind, shape):
        t = inp(f"loop.fifo.{mode}.{kind}{k[kind]}", shape)
        k[kind] += 1
        return t
    enc = [nxt("randn", (1, 4, 16, 16)) for _ in range(3)] if mode == "davis" else None
    prep = [nxt("randn_like", (1, 4, 1, 16, 16)) for _ in range(20)]
    noises, shifts, anchors = [], [], []
    for _ in range(3):
        noises.append([torch.cat([nxt("noise_like", (1, 4, 1, 16, 16)) for _ in range(8)], 2).cuda() for _ in range(4)])
        if mode == "davis":
            anchors.append(nxt("randn", (1, 4, 16, 16)).unsqueeze(2).cuda())
        shifts.append(nxt("randn_like", (1, 4, 16, 16)).cuda())
    return enc, prep, noises, shifts, anchors


def _spy_steps(sampler):
    """records (x_prev, pred_x0) of every OUTER ddim_step call (the mask-producer path runs an inner first pass with gamma = 0)"""
    calls, orig, state = [], sampler.ddim_step, {"call": -1}

    def spy(*a, **kw):
        outer = kw.get("gamma", 0.5) != 0.0
        if outer:
            state["call"] += 1
        xp, p0 = orig(*a, **kw)
        if outer:
            calls.append((xp.clone(), p0.clone()))
        return xp, p0
    sampler.ddim_step = spy
    return calls, state


@pytest.mark.parametrize("producer", ["lists", "callback", "lists_graph"])
def test_hip_fifo_loop_prompt_mode_vs_reference_golden(dm, producer):
    """fifo_ddim_sampling (funcs.py:243-373) without DAVIS data: queue from the cached latents (prepare_latents), the
    segmentation branch of ddim_step with the scripted Grounded-SAM-2 masks (as per-call lists, and through the mask-producer
    callback that is shown pred_x0 frame by frame), decode of every emitted frame, FreeInit shift -- against the REAL loop.
    "lists_graph": the same candidate lists with the whole iteration, segmentation bookkeeping included, as one hipGraph (no
    ddim_step call reaches the host: the per-call comparison of that path is test_fifo_graph_prompt_mode_calls_vs_reference_golden)"""
    from moca_video_amd.fifo import fifo_ddim_sampling, prepare_latents
    from moca_video_amd.sampler import DDIMSampler
    g = golden("loop_fifo")
    t = _text()
    _, prep, noises, shifts, _ = _fifo_noises("prompt")
    s = DDIMSampler(dm)
    s.make_schedule(16, ddim_eta=1.0, verbose=False)
    lat = prepare_latents(FIFO_ARGS, None, s, initial_latents=inp("loop.z16", (1, 4, 8, 16, 16)).cuda(), noises=prep)
    cond = {"c_crossattn": [t["c1"], t["c2"]], "fps": torch.tensor([10]).cuda()}
    cimg = (inp("loop.cimg", (1, 4, 1, 16, 16)) * 0.25 + 0.5).clamp(0, 1).cuda()
    calls, state = _spy_steps(s)
    kw = {}
    graph = producer == "lists_graph"
    if producer.startswith("lists"):
        kw["sam_masks"] = lambda i, wi: loop_sam_candidates(4 * i + wi, 8, 16, 16)
        kw["use_graph"] = graph
    else:
        def fn(pred_x0_frame, target, frame):
            assert pred_x0_frame.shape == (1, 4, 1, 16, 16) and target == "object."
            return loop_sam_candidates(state["call"], 8, 16, 16)[frame]
        kw["sam_masks_fn"] = fn
    frames = fifo_ddim_sampling(FIFO_ARGS, dm, cond, (1, 4, 8, 16, 16), s, cfg_scale=12.0, uc_emb=t["uc"], latents=lat,
                                conditioned_image=cimg, n_iterations=3, noises=noises, shift_noises=shifts, decode=True,
                                targets="object.", **kw)
    assert len(frames) == 3 and len(calls) == (0 if graph else 12)
    for c in range(len(calls)):
        assert relerr(calls[c][0].cpu(), g["prompt_x_prev"][c]) < TOL_FIFO, f"call {c} x_prev"
        assert relerr(calls[c][1].cpu(), g["prompt_pred_x0"][c]) < TOL_FIFO, f"call {c} pred_x0"
    for i in range(3):
        e = relerr(frames[i].cpu(), g["prompt_frames"][i])
        assert e < TOL_FIFO, f"decoded frame {i}: {e:.3e}"
    e = relerr(lat.cpu(), g["prompt_queue"][2])
    assert e < TOL_FIFO, f"queue after 3 iterations: {e:.3e}"


def test_fifo_graph_prompt_mode_calls_vs_reference_golden(dm):
    """The one-hipGraph iteration in PROMPT mode (ddim.py:592-606 -> `_apply_segmentation`: only t <= 300, previous-mask and IoU
    fallbacks, > 80 % reset, factor 2 -- `moca_sam_select_masks_f32` + the sam branch of the step kernel): x_prev / pred_x0 of every
    window of 3 iterations (eager, capture, replay) against the REAL loop's 12 `ddim_step` calls with the scripted producer; the
    candidate lists are uploaded per iteration, nothing is read back inside the loop.  pred_x0 is where the branch shows (the
    reference's caller discards it: the injection never reaches x_prev / the queue)."""
    from moca_video_amd.fifo import prepare_latents
    from moca_video_amd.fifo_graph import FifoEngine
    from moca_video_amd.sampler import DDIMSampler
    g = golden("loop_fifo")
    t = _text()
    _, prep, noises, shifts, _ = _fifo_noises("prompt")
    s = DDIMSampler(dm)
    s.make_schedule(16, ddim_eta=1.0, verbose=False)
    lat = prepare_latents(FIFO_ARGS, None, s, initial_latents=inp("loop.z16", (1, 4, 8, 16, 16)).cuda(), noises=prep)
    cond = {"c_crossattn": [t["c1"], t["c2"]], "fps": torch.tensor([10]).cuda()}
    uc = {"c_crossattn": [t["uc"]], "fps": cond["fps"]}
    cimg = (inp("loop.cimg", (1, 4, 1, 16, 16)) * 0.25 + 0.5).clamp(0, 1).cuda()
    eng = FifoEngine(FIFO_ARGS, dm, s, cond, uc, 12.0, lat, conditioned_image=cimg, n_slots=3, sam_capacity=64)
    injected = 0
    for i in range(3):
        cands = [loop_sam_candidates(4 * i + w, 8, 16, 16) for w in range(4)]
        if i == 1:                                   # candidates that already live on the device take the device-side copy path
            cands = [[None if c is None else c.cuda() for c in cw] for cw in cands]
        eng.step(noise=noises[i], shift_noise=shifts[i], sam_masks=cands)
        xp, p0 = eng.window_outputs()
        idx = eng.sam_idx.cpu().reshape(4, 8)
        for w in range(4):
            c = 4 * i + w
            assert relerr(xp[w].cpu(), g["prompt_x_prev"][c]) < TOL_FIFO, f"call {c} x_prev"
            assert relerr(p0[w].cpu(), g["prompt_pred_x0"][c]) < TOL_FIFO, f"call {c} pred_x0"
            # the device bookkeeping against the host one (DDIMSampler.select_sam_masks), bit for bit
            eff_h, idx_h = s.select_sam_masks(loop_sam_candidates(c, 8, 16, 16), eng._t_host[w], 16, 16, "cuda")
            assert np.array_equal(np.where(idx[w].numpy() >= 0, np.arange(8), -1), idx_h)
            sel = idx_h >= 0
            assert torch.equal(eng.sam_eff[w][torch.from_numpy(sel)].cpu(), eff_h[torch.from_numpy(sel)].cpu())
            injected += int(sel.sum())
    assert injected > 0 and eng.plan.graph is not None
    assert relerr(eng.latents().cpu(), g["prompt_queue"][2]) < TOL_FIFO
    eng.close()


def test_engine_argument_contracts(dm):
    """the engines refuse what they cannot run faithfully: more candidate masks than the pool holds, candidates handed to an engine
    built without a pool, guidance branches with different fps on the shared-prefix step graph"""
    from moca_video_amd.fifo_graph import BaseEngine, FifoEngine
    from moca_video_amd.sampler import DDIMSampler
    t = _text()
    s = DDIMSampler(dm)
    s.make_schedule(16, ddim_eta=1.0, verbose=False)
    cond = {"c_crossattn": [t["c1"], t["c2"]], "fps": torch.tensor([10]).cuda()}
    uc = {"c_crossattn": [t["uc"]], "fps": cond["fps"]}
    lat = inp("loop.q0", (1, 4, 20, 16, 16)).cuda()
    many = [[torch.ones(3, 16, 16) * 0.0 for _ in range(8)] for _ in range(4)]          # 3 masks per window frame
    eng = FifoEngine(FIFO_ARGS, dm, s, cond, uc, 12.0, lat.clone(), sam_capacity=4)
    with pytest.raises(ValueError, match="sam_capacity"):
        eng.step(sam_masks=many)
    eng.close()
    eng = FifoEngine(FIFO_ARGS, dm, s, cond, uc, 12.0, lat.clone())
    with pytest.raises(ValueError, match="sam_capacity"):
        eng.step(sam_masks=many)
    eng.close()
    x = inp("loop.base.randn0", (1, 4, 8, 16, 16)).cuda()
    c1 = {"c_crossattn": [t["c1"]], "fps": torch.tensor([10]).cuda()}
    u1 = {"c_crossattn": [t["uc"]], "fps": torch.tensor([24]).cuda()}
    assert not BaseEngine.supported(dm, x, c1, u1, 12.0) and BaseEngine.supported(dm, x, c1, dict(u1, fps=c1["fps"]), 12.0)
    be = BaseEngine(dm, s, x, c1, dict(u1, fps=c1["fps"]), 12.0)
    with pytest.raises(ValueError, match="fps"):
        be.reset(x, c1, u1)
    be.close()


def test_fifo_graph_with_different_fps_per_branch(dm):
    """guidance branches with DIFFERENT fps cannot share the UNet prefix (it adds one fps embedding): the engine then runs the plain
    batch of 2 x nW windows (gather repeats the windows) -- same video as the host-driven loop, which calls the two branches separately"""
    from moca_video_amd.fifo import fifo_ddim_sampling
    from moca_video_amd.sampler import DDIMSampler
    t = _text()
    s = DDIMSampler(dm)
    s.make_schedule(16, ddim_eta=1.0, verbose=False)
    cond = {"c_crossattn": [t["c1"], t["c2"]], "fps": torch.tensor([10]).cuda()}
    gen = torch.Generator(device="cuda").manual_seed(4)
    rnd = lambda *shape: torch.randn(*shape, device="cuda", generator=gen)
    lat0 = rnd(1, 4, 20, 16, 16)
    noises = [[rnd(1, 4, 8, 16, 16) for _ in range(4)] for _ in range(3)]
    shifts = [rnd(1, 4, 16, 16) for _ in range(3)]
    outs = {}
    import moca_video_amd.fifo as F
    for use_graph in (True, False):
        lat = lat0.clone()
        # fifo_ddim_sampling builds the unconditional dict as a copy of `cond` (funcs.py:268-270): give that copy its own fps where
        # the two paths receive it (the engine's constructor, the host loop's batched UNet call)
        real = F.FifoEngine.__init__

        def init(self_e, args, model, sampler, cond_, uc, *a, **k):
            uc = dict(uc, fps=torch.tensor([24]).cuda())
            return real(self_e, args, model, sampler, cond_, uc, *a, **k)
        F.FifoEngine.__init__ = init
        real_win = s.unet_windows

        def win(windows, c_, ts_list, unconditional_guidance_scale=1., unconditional_conditioning=None, **kw):
            return real_win(windows, c_, ts_list, unconditional_guidance_scale=unconditional_guidance_scale,
                            unconditional_conditioning=dict(unconditional_conditioning, fps=torch.tensor([24]).cuda()), **kw)
        s.unet_windows = win
        try:
            frames = fifo_ddim_sampling(FIFO_ARGS, dm, cond, (1, 4, 8, 16, 16), s, cfg_scale=12.0, uc_emb=t["uc"], latents=lat, n_iterations=3,
                                        noises=noises, shift_noises=shifts, use_graph=use_graph)
        finally:
            F.FifoEngine.__init__ = real
            s.unet_windows = real_win
        outs[use_graph] = (lat.clone(), [f.clone() for f in frames])
    assert relerr(outs[True][0], outs[False][0]) < 2e-2
    for a, b in zip(outs[True][1], outs[False][1]):
        assert relerr(a, b) < 2e-2
    # and it differs from the equal-fps video (the fps embedding matters): the shared-prefix engine on the same inputs
    lat = lat0.clone()
    fifo_ddim_sampling(FIFO_ARGS, dm, cond, (1, 4, 8, 16, 16), s, cfg_scale=12.0, uc_emb=t["uc"], latents=lat, n_iterations=3,
                       noises=noises, shift_noises=shifts)
    assert relerr(lat, outs[True][0]) > 1e-3


def test_sam_select_kernel_equals_host_bookkeeping():
    """moca_sam_select_masks_f32 against DDIMSampler.select_sam_masks (the statement-by-statement restatement of ddim.py:739-903 the
    host loop uses, itself pinned by tests/golden/sampler_sam*.npz) on random candidate sets: overlapping blobs around a drifting
    centre (IoU on both sides of 0.5), missing detections, several masks per frame of unequal count, > 80 % masks, empty masks
    (union 0 -> IoU 1), timesteps on both sides of 300, HW not a multiple of the block -- effective masks and frame flags bit for bit"""
    import ctypes as C
    from moca_video_amd import lib as L
    from moca_video_amd.sampler import DDIMSampler
    lib = L.load()
    H, W, f, nW = 13, 21, 8, 6
    HW = H * W
    rng = np.random.default_rng(5)
    s = DDIMSampler(types.SimpleNamespace(num_timesteps=1000))
    yy, xx = np.mgrid[0:H, 0:W]
    ts = rng.choice([100, 250, 300, 301, 700], size=(nW, f), p=[0.3, 0.3, 0.2, 0.1, 0.1]).astype(np.int64)
    cands = []
    for w in range(nW):
        cy, cx, row = rng.uniform(3, H - 3), rng.uniform(3, W - 3), []
        for i in range(f):
            k = rng.integers(0, 8)
            if k == 0:
                row.append(None)
                continue
            n = int(rng.integers(1, 4))
            ms = []
            for j in range(n):
                kind = rng.integers(0, 10)
                if kind == 0:
                    ms.append(np.ones((H, W), np.float32))                       # > 80 %: resets the frame
                elif kind == 1:
                    ms.append(np.zeros((H, W), np.float32))                      # empty: union 0 against an empty previous mask
                else:
                    cy += rng.uniform(-1.5, 1.5); cx += rng.uniform(-2.5, 2.5)
                    r = rng.uniform(2.0, 4.5)
                    ms.append((((yy - cy) ** 2 + (xx - cx) ** 2) < r * r).astype(np.float32) * rng.choice([1.0, 0.75]))
            row.append(torch.from_numpy(np.stack(ms).astype(np.float32)))      # (float32 * np.float64 scalar promotes)
        cands.append(row)
    pool, off, cnt = [], np.zeros(nW * f, np.int32), np.zeros(nW * f, np.int32)
    for w in range(nW):
        for i in range(f):
            if cands[w][i] is not None:
                off[w * f + i], cnt[w * f + i] = len(pool), cands[w][i].shape[0]
                pool.extend(cands[w][i].reshape(-1, HW))
    pool_d = torch.stack(pool).cuda()
    eff = torch.full((nW, f, HW), 7.0, device="cuda")
    idx = torch.full((nW * f,), 99, dtype=torch.int32, device="cuda")
    off_d, cnt_d, ts_d = (torch.from_numpy(a).cuda() for a in (off, cnt, ts.reshape(-1).copy()))   # named: a temporary's block is reused by the next
    assert int((off + cnt).max()) <= pool_d.shape[0] and pool_d.dtype == torch.float32 and pool_d.is_contiguous()
    L.check(lib.moca_sam_select_masks_f32(L.ptr(pool_d), L.ptr(off_d), L.ptr(cnt_d), L.ptr(ts_d), L.ptr(eff), L.ptr(idx), nW, f, HW,
                                          C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    idx = idx.cpu().numpy().reshape(nW, f)
    n_fallback = n_inject = 0
    for w in range(nW):
        eff_h, idx_h = s.select_sam_masks(cands[w], ts[w], H, W, "cuda")
        assert np.array_equal(idx[w], idx_h), f"window {w}: {idx[w]} vs {idx_h}"
        for i in range(f):
            if idx_h[i] >= 0:
                assert torch.equal(eff[w, i], eff_h[i]), f"window {w} frame {i}"
                n_inject += 1
                if cands[w][i] is None or not torch.equal(((cands[w][i].reshape(-1, HW) > 0.5).any(0)).float().cuda(), eff_h[i]):
                    n_fallback += 1
    assert n_inject >= 10 and n_fallback >= 3          # both the plain and the fallback / reset paths were taken


def test_fifo_graph_without_lookahead_reads_anchor_after_write_back(dm):
    """lookahead_denoising=False: the rank-0 window rewrites queue frame 0 (funcs.py:353-354) before `shift_latents` reads it as the
    FreeInit anchor (funcs.py:88) -- the engine must fetch the anchor BEHIND the step kernel (with lookahead frame 0 is never
    rewritten and the pre-UNet gather may fetch it).  Anchor bit-exact, graph loop == host-driven loop."""
    from moca_video_amd.fifo import fifo_ddim_sampling
    from moca_video_amd.fifo_graph import FifoEngine
    from moca_video_amd.sampler import DDIMSampler
    t = _text()
    s = DDIMSampler(dm)
    s.make_schedule(16, ddim_eta=1.0, verbose=False)
    cond = {"c_crossattn": [t["c1"], t["c2"]], "fps": torch.tensor([10]).cuda()}
    uc = {"c_crossattn": [t["uc"]], "fps": cond["fps"]}
    gen = torch.Generator(device="cuda").manual_seed(1)
    rnd = lambda *shape: torch.randn(*shape, device="cuda", generator=gen)
    for look, Q in ((False, 16), (True, 20)):
        args = types.SimpleNamespace(num_inference_steps=16, video_length=8, lookahead_denoising=look, num_partitions=2, new_video_length=10)
        lat0 = rnd(1, 4, Q, 16, 16)
        nW = 4 if look else 2
        noises = [[rnd(1, 4, 8, 16, 16) for _ in range(nW)] for _ in range(3)]
        shifts = [rnd(1, 4, 16, 16) for _ in range(3)]
        eng = FifoEngine(args, dm, s, cond, uc, 12.0, lat0.clone(), n_slots=3)
        for i in range(3):
            before = eng.latents().clone()
            eng.step(noise=noises[i], shift_noise=shifts[i])
            xp, _ = eng.window_outputs()
            want = before[0, :, 0] if look else xp[-1][0, :, 0]               # rank 0 is the last window in call order
            assert torch.equal(eng.anchor.view(4, 16, 16), want), f"lookahead={look} iteration {i}: FreeInit anchor"
        q_graph, em = eng.latents().clone(), eng.emitted_frames(0, 3).clone()
        eng.close()
        lat_h = lat0.clone()
        fr_h = fifo_ddim_sampling(args, dm, cond, (1, 4, 8, 16, 16), s, cfg_scale=12.0, uc_emb=t["uc"], latents=lat_h, n_iterations=3,
                                  noises=noises, shift_noises=shifts, use_graph=False)
        assert relerr(q_graph, lat_h) < 2e-2
        for i in range(3):
            assert relerr(em[:, :, [i]], fr_h[i]) < 2e-2


@pytest.mark.parametrize("use_graph", [False, True])
def test_hip_fifo_loop_davis_mode_vs_reference_golden(dm, use_graph):
    """fifo_ddim_sampling with davis_data: queue from the VAE encoding of the frames, DAVIS masks in ddim_step, DAVIS branch of
    shift_latents (anchor = posterior sample of the last frame's encoding) -- against the REAL loop; host-driven (every ddim_step
    call compared) and as one hipGraph per iteration (the anchor drawn from the once-encoded posterior moments by a kernel)"""
    from moca_video_amd.fifo import fifo_ddim_sampling, prepare_latents
    from moca_video_amd.sampler import DDIMSampler
    g = golden("loop_fifo")
    t = _text()
    enc, prep, noises, shifts, anchors = _fifo_noises("davis")
    s = DDIMSampler(dm)
    s.make_schedule(16, ddim_eta=1.0, verbose=False)
    dframes = (inp("loop.davis.frames", (1, 4, 3, 128, 128)) * 0.5).clamp(-1, 1).cuda()
    dmasks = (inp("loop.davis.masks", (1, 1, 20, 16, 16)) > 0.3).float()
    dmasks[:, :, 7] = 0.0
    dmasks = dmasks.cuda()
    z = dm.encode_first_stage_2DAE(dframes[:, :3], noise=torch.stack(enc, 2).cuda())
    lat = prepare_latents(FIFO_ARGS, None, s, initial_latents=z, noises=prep)
    cond = {"c_crossattn": [t["c1"], t["c2"]], "fps": torch.tensor([10]).cuda()}
    cimg = (inp("loop.cimg", (1, 4, 1, 16, 16)) * 0.25 + 0.5).clamp(0, 1).cuda()
    calls, _ = _spy_steps(s)
    frames = fifo_ddim_sampling(FIFO_ARGS, dm, cond, (1, 4, 8, 16, 16), s, cfg_scale=12.0, uc_emb=t["uc"], latents=lat,
                                conditioned_image=cimg, n_iterations=3, noises=noises, shift_noises=shifts, decode=True,
                                davis_data=(dframes, dmasks), anchor_noises=anchors, use_graph=use_graph)
    assert len(frames) == 3 and len(calls) == (0 if use_graph else 12)
    for c in range(len(calls)):
        assert relerr(calls[c][0].cpu(), g["davis_x_prev"][c]) < TOL_FIFO, f"call {c} x_prev"
        assert relerr(calls[c][1].cpu(), g["davis_pred_x0"][c]) < TOL_FIFO, f"call {c} pred_x0"
    for i in range(3):
        assert relerr(frames[i].cpu(), g["davis_frames"][i]) < TOL_FIFO, f"decoded frame {i}"
    assert relerr(lat.cpu(), g["davis_queue"][2]) < TOL_FIFO
    assert torch.equal(dmasks.cpu(), torch.from_numpy(g["davis_masks_after"][2]))


def test_fifo_graph_iteration_vs_reference_golden_masks(dm):
    """The ONE-hipGraph iteration (fifo_graph.FifoEngine: ring queue, device tables, batched guidance + ddim_step + write-back,
    FreeInit mix, shift) on the DAVIS golden's queue and masks handed in directly (`masks=`: same ddim_step branch, prompt-mode
    shift).  The first three iterations (eager, capture, replay) are compared call by call with the REAL loop's x_prev / pred_x0
    where the two modes coincide (iteration 0: the queues only differ after the first shift) and with the host-driven loop after."""
    from moca_video_amd.fifo import fifo_ddim_sampling, prepare_latents
    from moca_video_amd.fifo_graph import FifoEngine
    from moca_video_amd.sampler import DDIMSampler
    g = golden("loop_fifo")
    t = _text()
    enc, prep, noises, shifts, _ = _fifo_noises("davis")
    s = DDIMSampler(dm)
    s.make_schedule(16, ddim_eta=1.0, verbose=False)
    dframes = (inp("loop.davis.frames", (1, 4, 3, 128, 128)) * 0.5).clamp(-1, 1).cuda()
    dmasks = (inp("loop.davis.masks", (1, 1, 20, 16, 16)) > 0.3).float()
    dmasks[:, :, 7] = 0.0
    z = dm.encode_first_stage_2DAE(dframes[:, :3], noise=torch.stack(enc, 2).cuda())
    lat0 = prepare_latents(FIFO_ARGS, None, s, initial_latents=z, noises=prep)
    cond = {"c_crossattn": [t["c1"], t["c2"]], "fps": torch.tensor([10]).cuda()}
    uc = {"c_crossattn": [t["uc"]], "fps": torch.tensor([10]).cuda()}
    cimg = (inp("loop.cimg", (1, 4, 1, 16, 16)) * 0.25 + 0.5).clamp(0, 1).cuda()
    eng = FifoEngine(FIFO_ARGS, dm, s, cond, uc, 12.0, lat0.clone(), conditioned_image=cimg, masks=dmasks.cuda(), n_slots=3)
    eng.step(noise=noises[0], shift_noise=shifts[0])
    xp, p0 = eng.window_outputs()
    for c in range(4):
        assert relerr(xp[c].cpu(), g["davis_x_prev"][c]) < TOL_FIFO, f"window {c} x_prev"
        assert relerr(p0[c].cpu(), g["davis_pred_x0"][c]) < TOL_FIFO, f"window {c} pred_x0"
    eng.step(noise=noises[1], shift_noise=shifts[1])
    eng.step(noise=noises[2], shift_noise=shifts[2])
    assert eng.plan.graph is not None, "the iteration was not captured into a hipGraph"
    q_graph, m_graph, em = eng.latents().clone(), eng.mask_queue().clone(), eng.emitted_frames(0, 3).clone()
    eng.close()
    # host-driven loop, same inputs
    lat_h, m_h = lat0.clone(), dmasks.cuda().clone()
    fr_h = fifo_ddim_sampling(FIFO_ARGS, dm, cond, (1, 4, 8, 16, 16), s, cfg_scale=12.0, uc_emb=t["uc"], latents=lat_h,
                              conditioned_image=cimg, masks=m_h, n_iterations=3, noises=noises, shift_noises=shifts, use_graph=False)
    assert torch.equal(m_graph, m_h)
    assert relerr(q_graph, lat_h) < 2e-2          # (one B = 8 forward with two context segments vs two B = 4 forwards: other tilings)
    for i in range(3):
        assert relerr(em[:, :, [i]], fr_h[i]) < 2e-2
    # and through the public entry point (graph path), in-place queue / mask update included
    lat_g, m_g = lat0.clone(), dmasks.cuda().clone()
    fr_g = fifo_ddim_sampling(FIFO_ARGS, dm, cond, (1, 4, 8, 16, 16), s, cfg_scale=12.0, uc_emb=t["uc"], latents=lat_g,
                              conditioned_image=cimg, masks=m_g, n_iterations=3, noises=noises, shift_noises=shifts)
    assert torch.equal(lat_g, q_graph) and torch.equal(m_g, m_graph)
    for i in range(3):
        assert torch.equal(fr_g[i], em[:, :, [i]])


def test_fifo_graph_device_noise_is_reproducible(dm):
    """without explicit noise the draws come from the device Philox stream keyed by (seed, iteration): same seed -> same video,
    other seed -> another one; 5 iterations = eager + capture + 3 replays; nothing read back inside the loop"""
    from moca_video_amd.fifo import fifo_ddim_sampling
    from moca_video_amd.sampler import DDIMSampler
    t = _text()
    s = DDIMSampler(dm)
    s.make_schedule(16, ddim_eta=1.0, verbose=False)
    cond = {"c_crossattn": [t["c1"], t["c2"]], "fps": torch.tensor([10]).cuda()}
    lat0 = inp("loop.q0", (1, 4, 20, 16, 16)).cuda()
    run = lambda seed: fifo_ddim_sampling(FIFO_ARGS, dm, cond, (1, 4, 8, 16, 16), s, cfg_scale=12.0, uc_emb=t["uc"],
                                          latents=lat0.clone(), n_iterations=5, seed=seed)
    a, b, c = run(7), run(7), run(8)
    assert len(a) == 5 and all(torch.isfinite(x).all() for x in a)
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    assert not torch.equal(a[-1], c[-1])


# ---------------------------------------------------------------------------------------------------------------- kernels
def _philox_numpy(seed, it, n):
    """Philox4x32-10, counter (i, i >> 32, iteration, 0x4d6f4341), key = seed words; Box-Muller like csrc/fifo.hip"""
    n4 = (n + 3) // 4
    c = np.zeros((n4, 4), np.uint64)
    c[:, 0] = np.arange(n4) & 0xffffffff
    c[:, 1] = np.arange(n4) >> 32
    c[:, 2] = it
    c[:, 3] = 0x4d6f4341
    k0, k1 = np.uint64(seed & 0xffffffff), np.uint64((seed >> 32) & 0xffffffff)
    M = np.uint64(0xffffffff)
    for _ in range(10):
        p0 = np.uint64(0xD2511F53) * c[:, 0]
        p1 = np.uint64(0xCD9E8D57) * c[:, 2]
        n0 = ((p1 >> np.uint64(32)) ^ c[:, 1] ^ k0) & M
        n2 = ((p0 >> np.uint64(32)) ^ c[:, 3] ^ k1) & M
        c = np.stack([n0, p1 & M, n2, p0 & M], 1)
        k0 = (k0 + np.uint64(0x9E3779B9)) & M
        k1 = (k1 + np.uint64(0xBB67AE85)) & M
    u = (c.astype(np.float32) + np.float32(0.5)) * np.float32(2.3283064365386963e-10)
    out = np.zeros((n4, 4), np.float64)
    for h in range(2):
        r = np.sqrt(-2.0 * np.log(u[:, 2 * h].astype(np.float64)))
        th = 6.283185307179586 * u[:, 2 * h + 1].astype(np.float64)
        out[:, 2 * h], out[:, 2 * h + 1] = r * np.cos(th), r * np.sin(th)
    return out.reshape(-1)[:n]


def test_fifo_randn_kernel_is_philox_box_muller():
    import ctypes as C
    from moca_video_amd import lib as L
    lib = L.load()
    n = 100003
    seed = 0x1234567_89abcdef
    for it in (0, 5):
        st = L.FifoState(3, it, seed & 0xffffffff, (seed >> 32) & 0xffffffff, 0)
        state = torch.frombuffer(bytearray(bytes(st)), dtype=torch.int32).cuda()
        out = torch.full((n,), 7.0, device="cuda")
        L.check(lib.moca_fifo_randn_f32(L.ptr(state), L.ptr(out), n, C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        got = out.cpu().numpy().astype(np.float64)
        ref = _philox_numpy(seed, it, n)
        assert np.abs(got - ref).max() < 2e-4          # fp32 log / sincos against float64
        assert abs(got.mean()) < 0.02 and abs(got.std() - 1.0) < 0.02
    state[4] = 1                                        # ext_noise: the kernel must leave the buffer alone
    out.fill_(7.0)
    L.check(lib.moca_fifo_randn_f32(L.ptr(state), L.ptr(out), n, C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    assert bool((out == 7.0).all())


def test_fifo_step_windows_equals_per_window_kernel():
    """moca_fifo_step_windows_f32 (guidance + step of all windows + ring write-back, head != 0) == moca_cfg_combine_f32 +
    moca_fifo_ddim_step_f32 per window + the reference's slice assignment, bit for bit; then the advance"""
    import ctypes as C
    from moca_video_amd import lib as L
    from moca_video_amd.sampler import DDIMSampler
    lib = L.load()
    stream = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
    nW, Cc, Q, f, H, W, head = 4, 4, 20, 8, 12, 10, 13
    HW = H * W
    from oracle import sampler_oracle as SO
    buf = SO.ddpm_buffers()
    fake = types.SimpleNamespace(num_timesteps=1000, alphas_cumprod=buf["alphas_cumprod"], use_scale=True, scale_arr=buf["scale_arr"],
                                 betas=buf["betas"])
    s = DDIMSampler(fake)
    s.make_schedule(16, ddim_eta=1.0, verbose=False)
    ts_all = np.concatenate([np.full((4,), s.ddim_timesteps[0]), s.ddim_timesteps])
    idx_all = np.concatenate([np.full((4,), 0), np.arange(16)])
    starts = [12, 8, 4, 0]
    g = torch.Generator(device="cuda").manual_seed(0)
    rnd = lambda *shape: torch.randn(*shape, device="cuda", generator=g)
    queue_lin = rnd(Cc, Q, HW)                                   # frame order
    mask_lin = (rnd(Q, HW) > 0.3).float()
    mask_lin[7] = 0.0
    ring = lambda t, dim: torch.roll(t, head, dims=dim).contiguous()
    queue, mask = ring(queue_lin, 1), ring(mask_lin, 0)
    x = torch.stack([queue_lin[:, s0:s0 + f] for s0 in starts]).contiguous()          # [nW][C][f][HW]
    e_c, e_u, nz = rnd(nW, Cc, f, HW), rnd(nW, Cc, f, HW), rnd(nW, Cc, f, HW)
    cond = torch.rand(Cc, HW, device="cuda", generator=g)
    coef = np.zeros((nW, f, 6), np.float32); enh = np.ones((nW, f), np.float32); mfr = np.full((nW, f), -1, np.int32)
    for w, s0 in enumerate(starts):
        coef[w], enh[w], mi = s.step_tables(idx_all[s0:s0 + f], ts_all[s0:s0 + f], H, f)
        mfr[w] = np.where(mi >= 0, s0 + mi, -1)
    dev = lambda a: torch.from_numpy(a).cuda()
    coef_d, enh_d, mfr_d, ws_d = dev(coef), dev(enh), dev(mfr), dev(np.asarray(starts, np.int32))
    st = L.FifoState(head, 2, 1, 2, 1)
    state = torch.frombuffer(bytearray(bytes(st)), dtype=torch.int32).cuda()
    msum = torch.empty(Q, device="cuda")
    L.check(lib.moca_mask_frame_sums_f32(L.ptr(mask), L.ptr(msum), Q, HW, stream()))
    assert torch.equal(msum.cpu(), mask.sum(1).cpu())
    mom = torch.zeros(nW, Cc, f, HW, device="cuda")
    xp, p0 = torch.empty_like(x), torch.empty_like(x)
    p = L.FifoStepParams()
    p.state, p.x, p.eps_c, p.eps_u, p.noise = (t.data_ptr() for t in (state, x, e_c, e_u, nz))
    p.momentum, p.queue, p.x_prev, p.pred_x0 = mom.data_ptr(), queue.data_ptr(), xp.data_ptr(), p0.data_ptr()
    p.coef, p.win_start, p.mask, p.mask_sums = coef_d.data_ptr(), ws_d.data_ptr(), mask.data_ptr(), msum.data_ptr()
    p.mask_frame, p.enh, p.cond = mfr_d.data_ptr(), enh_d.data_ptr(), cond.data_ptr()
    p.cfg_scale, p.beta, p.one_minus_beta, p.gamma, p.one_minus_gamma = 12.0, 0.9, float(np.float32(1 - 0.9)), 0.5, 0.5
    p.nW, p.C, p.Q, p.f, p.HW, p.wb_from = nW, Cc, Q, f, HW, f // 2
    L.check(lib.moca_fifo_step_windows_f32(C.byref(p), stream()))
    # reference composition: per-window kernels on the linear queue
    lin = queue_lin.clone()
    for w, s0 in enumerate(starts):
        eps = torch.empty_like(e_c[w])
        L.check(lib.moca_cfg_combine_f32(L.ptr(e_c[w]), L.ptr(e_u[w]), L.ptr(eps), 12.0, eps.numel(), stream()))
        m1 = torch.zeros(1, Cc, f, HW, device="cuda")
        a, b = torch.empty(1, Cc, f, HW, device="cuda"), torch.empty(1, Cc, f, HW, device="cuda")
        wmask = mask_lin[s0:s0 + f].reshape(1, 1, f, HW).contiguous()
        _, _, mi = s.step_tables(idx_all[s0:s0 + f], ts_all[s0:s0 + f], H, f)
        scratch = torch.empty(f, device="cuda")
        L.check(lib.moca_fifo_ddim_step_f32(L.ptr(x[w]), L.ptr(eps), L.ptr(nz[w]), L.ptr(m1), L.ptr(a), L.ptr(b), L.ptr(coef_d[w]),
                                            L.ptr(wmask), L.ptr(cond.reshape(1, Cc, HW)), L.ptr(dev(mi)), L.ptr(enh_d[w]), L.ptr(scratch),
                                            1, Cc, f, f, HW, 0.9, float(np.float32(1 - 0.9)), 0.5, 0.5, stream()))
        assert torch.equal(a[0], xp[w]), f"window {w} x_prev"
        assert torch.equal(b[0], p0[w]), f"window {w} pred_x0"
        assert torch.equal(m1[0], mom[w])
        lin[:, s0 + f // 2:s0 + f] = a[0][:, f // 2:]
    assert torch.equal(torch.roll(queue, -head, dims=1), lin)
    # advance: emission of frame f//2, new frame into the dequeued slot, mask tail kept, head / iter bumped, ext flag cleared
    newf, emitted = rnd(Cc, HW), torch.zeros(3, Cc, HW, device="cuda")
    L.check(lib.moca_fifo_advance_f32(L.ptr(state), L.ptr(queue), L.ptr(newf), L.ptr(emitted), 3, f // 2, L.ptr(mask), L.ptr(msum),
                                      Cc, Q, HW, stream()))
    sv = state.cpu().tolist()
    assert sv[0] == (head + 1) % Q and sv[1] == 3 and sv[4] == 0
    assert torch.equal(emitted[2], lin[:, f // 2]) and not emitted[:2].any()
    after = torch.roll(queue, -sv[0], dims=1)
    assert torch.equal(after[:, :-1], lin[:, 1:]) and torch.equal(after[:, -1], newf)
    m_after = torch.roll(mask, -sv[0], dims=0)
    assert torch.equal(m_after[:-1], mask_lin[1:]) and torch.equal(m_after[-1], mask_lin[-1])
    assert torch.equal(torch.roll(msum, -sv[0]).cpu(), m_after.sum(1).cpu())
