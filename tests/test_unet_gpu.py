"""GPU parity of the HIP UNet (through the drop-in UNetModel / DiffusionWrapper boundary, i.e.
through the C-ABI) against (a) golden outputs of the real reference and (b) the CPU oracle on
the same seeded inputs.  fp16 storage / fp32 accumulate vs the fp32 reference:
the stated fp16 tolerance and where it comes from: every stored activation is rounded to fp16 once (relative rms error
2^-11 / sqrt(3) = 2.8e-4); the longest residual path of the UNet crosses ~150 such roundings (25 blocks x ~6 stores), GroupNorm /
LayerNorm keep the relative scale, so the errors add as a random walk: sqrt(150) x 2.8e-4 = 3.4e-3 predicted relative rms at the
output.  Observed (gpurun_out/r3_errlog.txt, MOCA_ERRLOG): rms 2.74e-3 / max-norm 3.04e-3 on the reduced-width UNet (short K, small
tensors: the worst case), 2.27e-3 / 2.56e-3 at the full 1.41 B-parameter width, <= 8.1e-4 on single blocks.  Bounds = at most
1.5 x the largest observed value: whole UNet rms <= 4e-3 x rms(ref) AND max|err| <= 4.5e-3 x max|ref|; single blocks 1.2e-3."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from helpers import REDUCED, golden, inp, relerr, state_dict_for  # noqa: E402

TOL_UNET = 4.5e-3      # max-norm, whole UNet: 1.5 x the observed 3.04e-3
TOL_BLOCK = 1.2e-3     # max-norm, one block: 1.5 x the observed 8.1e-4
TOL_RMS = 4e-3         # relative RMS (1.46 x the observed 2.74e-3 on the reduced-width UNet; the random-walk estimate is 3.4e-3): a
                       # wrong epilogue in a low-magnitude region cannot hide under the max-norm bound


def rmserr(got, ref):
    got, ref = torch.as_tensor(got).float(), torch.as_tensor(ref).float()
    return ((got - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt().clamp_min(1e-20)).item()


def check(got, ref, tol_max, what):
    e, r = relerr(got, ref), rmserr(got, ref)
    print(f"[parity] {what}: max-norm rel err {e:.2e}, rel rms {r:.2e}")
    if os.environ.get("MOCA_ERRLOG"):
        with open(os.environ["MOCA_ERRLOG"], "a") as f:
            f.write(f"{r:.3e} rms {os.environ.get('PYTEST_CURRENT_TEST', '?')} {what}\n")
    assert e < tol_max and r < TOL_RMS, f"{what}: max-norm rel err {e:.3e} (tol {tol_max:.0e}), rel rms {r:.3e} (tol {TOL_RMS:.1e})"
    return e, r


def _filled(block, seed):
    block.load_state_dict(state_dict_for(block, seed), strict=True)
    return block.cuda()


# ---- single blocks against goldens of the REAL reference blocks (tools/make_golden.py::blocks) -------------------------
@pytest.mark.parametrize("name,cout,seed", [("block_resblock", 128, 1), ("block_resblock_same", 64, 2)])
def test_block_resblock_vs_reference_golden(name, cout, seed):
    """ResBlock._forward + TemporalConvBlock (openaimodel3d.py:208-234,269-276): GroupNorm+SiLU, conv3x3 + emb row add,
    skip 1x1 conv, residual epilogue, 4 x (5-D GroupNorm + temporal conv), b=2, t=4"""
    from moca_video_amd.blockplan import BlockRunner
    from moca_video_amd.unet import _ResBlock
    run = BlockRunner(_filled(_ResBlock(64, 256, cout, True), seed), B=2, T=4, H=6, W=10)
    x, emb = inp("rb.x", (8, 64, 6, 10)).cuda(), inp("rb.emb", (8, 256)).cuda()
    ref = golden(name)["y"]
    for it in range(3):          # eager, capture, replay
        check(run(x, emb=emb).cpu(), ref, TOL_BLOCK, f"{name} pass {it}")


def test_block_spatial_transformer_vs_reference_golden():
    """SpatialTransformer (attention.py:262-278): GroupNorm, proj_in, self-attn, cross-attn on a 77-token context per frame,
    GEGLU feed-forward, proj_out + residual"""
    from moca_video_amd.blockplan import BlockRunner
    from moca_video_amd.unet import _SpatialTransformer
    run = BlockRunner(_filled(_SpatialTransformer(128, 2, 64, 1, 96, True), 3), B=4, T=1, H=6, W=10, L=77, context_dim=96)
    x, ctx = inp("st.x", (4, 128, 6, 10)).cuda(), inp("st.ctx", (4, 77, 96)).cuda()
    for it in range(3):
        check(run(x, context=ctx).cpu(), golden("block_spatial_transformer")["y"], TOL_BLOCK, f"spatial transformer pass {it}")


def test_block_temporal_transformers_vs_reference_golden():
    """TemporalTransformer (attention.py:331-373): the linear-projection flavour (T=8) and the Conv1d-projected 8-head
    `init_attn` flavour (T=16)"""
    from moca_video_amd.blockplan import BlockRunner
    from moca_video_amd.unet import _TemporalTransformer
    run = BlockRunner(_filled(_TemporalTransformer(128, 2, 64, 1, True), 4), B=2, T=8, H=3, W=5)
    x5 = inp("tt.x", (2, 128, 8, 3, 5))
    ref = torch.from_numpy(golden("block_temporal_transformer")["y"])          # [b, c, t, h, w]
    y = run(x5.permute(0, 2, 1, 3, 4).reshape(16, 128, 3, 5).cuda()).cpu().reshape(2, 8, 128, 3, 5).permute(0, 2, 1, 3, 4)
    check(y, ref, TOL_BLOCK, "temporal transformer")
    run = BlockRunner(_filled(_TemporalTransformer(64, 8, 64, 1, False), 5), B=1, T=16, H=3, W=5)
    x5 = inp("ti.x", (1, 64, 16, 3, 5))
    ref = torch.from_numpy(golden("block_init_attn")["y"])
    y = run(x5.permute(0, 2, 1, 3, 4).reshape(16, 64, 3, 5).cuda()).cpu().reshape(1, 16, 64, 3, 5).permute(0, 2, 1, 3, 4)
    check(y, ref, TOL_BLOCK, "init_attn")


def test_block_down_up_vs_reference_golden():
    """Downsample (stride-2 conv) / Upsample (nearest x2 fused into the conv gather), openaimodel3d.py:56-121"""
    from moca_video_amd.blockplan import BlockRunner
    from moca_video_amd.unet import _Downsample, _Upsample
    g = golden("block_down_up")
    x = inp("ud.x", (3, 64, 6, 10)).cuda()
    check(BlockRunner(_filled(_Downsample(64), 6), B=3, T=1, H=6, W=10)(x).cpu(), g["down"], TOL_BLOCK, "downsample")
    check(BlockRunner(_filled(_Upsample(64), 7), B=3, T=1, H=6, W=10)(x).cpu(), g["up"], TOL_BLOCK, "upsample")


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
        check(y.cpu(), ref, TOL_UNET, f"{case} pass {it}")
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
    check(y.cpu(), ref, TOL_UNET, "fresh inputs vs oracle")


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
    check(y.cpu(), torch.from_numpy(g["fifo"]), TOL_UNET, "wrapper boundary")


def test_unet_rejects_cpu_and_bad_shapes(reduced_model):
    x = inp("reduced.uniform.x", (1, 4, 8, 16, 16))
    with pytest.raises(ValueError):
        reduced_model(x, torch.tensor([1]), context=torch.zeros(1, 77, 128))          # CPU tensor: no CPU path
    with pytest.raises(ValueError):
        reduced_model(x.cuda(), torch.tensor([1, 2, 3]).cuda(), context=torch.zeros(1, 77, 128).cuda())


def test_fifo_batched_windows_equal_sequential():
    """N2: one outer FIFO iteration with the 8 windows batched into one UNet launch equals the reference's
    sequential window loop (funcs.py:305-355) -- same queue afterwards, same emitted frame."""
    import types
    from moca_video_amd import DenoiseModel
    from moca_video_amd.fifo import fifo_ddim_sampling
    from moca_video_amd.sampler import DDIMSampler
    dm = DenoiseModel({"target": "lvdm.modules.networks.openaimodel3d.UNetModel", "params": REDUCED})
    unet = dm.model.diffusion_model
    unet.load_state_dict(state_dict_for(unet, 11), strict=True)
    dm = dm.cuda()
    args = types.SimpleNamespace(num_inference_steps=64, video_length=16, lookahead_denoising=True, num_partitions=4,
                                 new_video_length=100)
    h, w = 8, 8
    cond = {"c_crossattn": [inp("fb.ctx", (1, 77, 128)).cuda()], "fps": torch.tensor([10]).cuda()}
    uc_emb = inp("fb.uc", (1, 77, 128)).cuda()
    q0 = inp("fb.queue", (1, 4, 72, h, w)).cuda()
    noises = [[inp(f"fb.nz{i}.{wi}", (1, 4, 16, h, w)).cuda() for wi in range(8)] for i in range(2)]
    shift = [inp(f"fb.sh{i}", (1, 4, h, w)).cuda() for i in range(2)]
    mask = (inp("fb.mask", (1, 1, 72, h, w)) > 0.3).float().cuda()
    cimg = inp("fb.cimg", (1, 4, 1, h, w)).cuda()
    outs = []
    for batched in (False, True):
        s = DDIMSampler(dm)
        s.make_schedule(64, ddim_eta=1.0, verbose=False)
        lat = q0.clone()
        frames = fifo_ddim_sampling(args, dm, cond, (1, 4, 16, h, w), s, cfg_scale=12.0, uc_emb=uc_emb, latents=lat,
                                    conditioned_image=cimg, masks=mask.clone(), n_iterations=2, batch_windows=batched,
                                    noises=noises, shift_noises=shift)
        outs.append((lat, frames))
    # B = 16 (the one-graph iteration) and B = 1 launches take different tilings / split-k factors (different fp32 summation orders)
    # and the differences are fed back through two CFG-12 iterations: observed 5.3e-3 -> bound 8e-3 (1.5 x)
    assert relerr(outs[1][0], outs[0][0]) < 8e-3
    for a, b in zip(outs[0][1], outs[1][1]):
        assert relerr(b, a) < 8e-3
    assert not torch.equal(outs[0][0], q0)


def test_shared_cfg_prefix_equals_separate_forwards(reduced_model):
    """UNetModel.forward_segments(shared_x=True): the two classifier-free-guidance branches (same latents and timesteps, contexts of
    DIFFERENT lengths: 154 and 77 tokens) as one forward that computes everything before the first cross-attention once -- against
    one plain forward per branch (uniform and per-frame FIFO timesteps, eager / capture / replay)."""
    x = inp("sp.x", (2, 4, 8, 16, 16)).cuda()
    c154, c77 = inp("sp.c154", (2, 154, 128)).cuda(), inp("sp.c77", (2, 77, 128)).cuda()
    fps = torch.tensor([10, 24]).cuda()
    for t in (torch.tensor([981, 20]).cuda(), torch.arange(16).cuda() * 60):          # [B] and [B*T]
        ref = torch.cat([reduced_model(x, t, context=c154, fps=fps), reduced_model(x, t, context=c77, fps=fps)], 0)
        for it in range(3):
            out = reduced_model.forward_segments(x, t, [c154, c77], fps=[fps, fps], shared_x=True)
            assert out.shape == ref.shape
            # the prefix runs at half the batch (other tilings / summation orders): fp16 noise level, not bit equality
            assert relerr(out, ref) < TOL_UNET, f"shared prefix, pass {it}: {relerr(out, ref):.2e}"
    # equal context lengths: the cross-attentions after the split run as ONE launch over both segments
    out = reduced_model.forward_segments(x, t, [c77, c77.flip(0)], fps=[fps, fps], shared_x=True)
    ref = torch.cat([reduced_model(x, t, context=c77, fps=fps), reduced_model(x, t, context=c77.flip(0), fps=fps)], 0)
    assert relerr(out, ref) < TOL_UNET


def test_shared_cfg_prefix_refuses_different_fps(reduced_model):
    """the shared prefix adds ONE fps embedding to the rows both branches read (conv_in .. the first ResBlock): a per-segment fps
    list with different entries is refused, and the sampler's guidance falls back to the plain batch, which gives each branch its
    own fps (== two separate apply_model calls)"""
    import types
    from moca_video_amd.sampler import DDIMSampler
    from moca_video_amd.unet import same_fps
    x = inp("sp.x", (2, 4, 8, 16, 16)).cuda()
    c1, c2 = inp("sp.c77", (2, 77, 128)).cuda(), inp("sp.c77", (2, 77, 128)).cuda().flip(0)
    f1, f2 = torch.tensor([10, 24]).cuda(), torch.tensor([10, 12]).cuda()
    t = torch.tensor([981, 20]).cuda()
    assert same_fps([f1, f1.clone()]) and same_fps([16, 16]) and same_fps([8, torch.tensor([8])]) and not same_fps([f1, f2]) and not same_fps([8, 16])
    with pytest.raises(ValueError):
        reduced_model.forward_segments(x, t, [c1, c2], fps=[f1, f2], shared_x=True)
    wrap = types.SimpleNamespace(diffusion_model=reduced_model, conditioning_key="crossattn")
    model = types.SimpleNamespace(model=wrap, num_timesteps=1000,
                                  apply_model=lambda x_, t_, c_, **kw: reduced_model(x_, t_, context=torch.cat(c_["c_crossattn"], 1), fps=c_["fps"]))
    s = DDIMSampler(model)
    got = s._cfg_eps(x, t, {"c_crossattn": [c1], "fps": f1}, {"c_crossattn": [c2], "fps": f2}, 3.0)
    e_c, e_u = reduced_model(x, t, context=c1, fps=f1).float(), reduced_model(x, t, context=c2, fps=f2).float()
    right = e_u + 3.0 * (e_c - e_u)
    assert relerr(got, right) < 4 * TOL_UNET
    wrong = reduced_model(x, t, context=c2, fps=f1).float()                         # the unconditional branch with the OTHER fps
    assert relerr(got, wrong + 3.0 * (e_c - wrong)) > relerr(got, right)


def test_weight_prefetch_changes_nothing():
    """moca_gemm_params.prefetch: spare blocks of a GEMM launch's grid read the next weight-heavy launch's weights (they only move
    lines into the memory-side cache).  A UNet forward with the prefetches (eager, captured, replayed) is bit-identical to one
    without; a misaligned / null range is refused."""
    import ctypes as C
    from moca_video_amd import UNetModel, lib as L, ops
    from moca_video_amd import plan as P
    a, w = torch.randn(512, 256, device="cuda").half(), torch.randn(256, 256, device="cuda").half()
    pw = ops.pack_linear(w)
    big = torch.randn(3 << 20, device="cuda").half()
    out0, out1 = torch.empty(512, 256, device="cuda", dtype=torch.float16), torch.empty(512, 256, device="cuda", dtype=torch.float16)
    ops.gemm(a, pw, out0, M=512)
    ops.gemm(a, pw, out1, M=512, prefetch=big)
    assert torch.equal(out0, out1)
    prm = ops._gemm_params(a, pw, out1, M=512, prefetch=big)
    prm.prefetch = big.data_ptr() + 2
    assert L.load().moca_gemm_f16(C.byref(prm), None) == -1
    # whole forward: a width at which some launches carry >= PREFETCH_MIN_BYTES of weights
    cfg = dict(REDUCED, model_channels=128, context_dim=128)
    x = inp("pf.x", (2, 4, 8, 16, 16)).cuda()
    ctx = inp("pf.ctx", (2, 77, 128)).cuda()
    t = torch.tensor([700, 30]).cuda()
    outs = []
    old_min, old_host = P.PREFETCH_MIN_BYTES, P.PREFETCH_HOST_FLOP
    try:
        P.PREFETCH_MIN_BYTES, P.PREFETCH_HOST_FLOP = 1 << 20, 0.0
        for on in (True, False):
            m = UNetModel(**cfg)
            m.load_state_dict(state_dict_for(m, 3), strict=True)
            m = m.cuda()
            m.weight_prefetch = on
            ys = [m(x, t, context=ctx, fps=torch.tensor([8, 8]).cuda()) for _ in range(3)]
            plan = next(iter(m._plans.values()))
            n_pf = sum(1 for s in plan.steps if getattr(s, "keywords", {}).get("prefetch") is not None)
            assert (n_pf > 10) == on and plan.graph is not None
            assert torch.equal(ys[0], ys[1]) and torch.equal(ys[1], ys[2])
            outs.append(ys[2])
    finally:
        P.PREFETCH_MIN_BYTES, P.PREFETCH_HOST_FLOP = old_min, old_host
    assert torch.equal(outs[0], outs[1]) and torch.isfinite(outs[0]).all()


def test_forward_concurrent_equals_forward(reduced_model):
    """two forwards launched as separate hipGraphs on separate streams return what forward() returns"""
    g = golden("unet_reduced")
    x = inp("reduced.uniform.x", (1, 4, 8, 16, 16)).cuda()
    c1 = inp("reduced.uniform.ctx", (1, 77, 128)).cuda()
    c2 = inp("fresh.ctx2", (1, 77, 128)).cuda()
    t = torch.from_numpy(g["uniform__t"]).cuda()
    ref1 = reduced_model(x, t, context=c1, fps=16)
    ref2 = reduced_model(x, t, context=c2, fps=16)
    for _ in range(3):   # eager, capture, replay of the two replica plans
        o1, o2 = reduced_model.forward_concurrent([dict(x=x, timesteps=t, context=c1, fps=16),
                                                   dict(x=x, timesteps=t, context=c2, fps=16)])
        assert torch.equal(o1, ref1) and torch.equal(o2, ref2)
    check(o1.cpu(), torch.from_numpy(g["uniform"]), TOL_UNET, "concurrent")


@pytest.fixture(scope="module")
def full_dm():
    """The full 1.41 B-parameter UNet of inference_t2v_512_v2.0.yaml inside the drop-in DenoiseModel, weights = weightgen seed 11
    (what tools/make_golden.py loaded into the REAL reference; regenerated here, ~1-2 min of host time, shared by the
    full-width tests of this module)."""
    from helpers import FULL
    from moca_video_amd import DenoiseModel
    dm = DenoiseModel({"target": "lvdm.modules.networks.openaimodel3d.UNetModel", "params": FULL})
    m = dm.model.diffusion_model
    m.load_state_dict(state_dict_for(m, 11), strict=True)
    return dm.cuda()


def test_unet_full_width_vs_reference_golden(full_dm):
    """The full 1.41 B-parameter UNet (inference_t2v_512_v2.0.yaml) against outputs of the REAL reference UNet captured by
    `tools/make_golden.py --only full`: config[0] shape (uniform t, 77 tokens; FIFO per-frame t, 154 tokens) and the headline
    16x40x64 shape as a FIFO window call."""
    m = full_dm.model.diffusion_model
    for tag, shape in (("full", (4, 8, 32, 32)), ("full_cfgN", (4, 16, 40, 64))):
        g = golden("unet_" + tag)
        for name in sorted(k for k in g.files if "__" not in k):
            L = int(g[name + "__L"])
            x = inp(f"{tag}.{name}.x", (1,) + shape).cuda()
            ctx = inp(f"{tag}.{name}.ctx", (1, L, 1024)).cuda()
            t = torch.from_numpy(g[name + "__t"]).cuda()
            fps = torch.from_numpy(np.atleast_1d(g[name + "__fps"])).cuda()
            y = m(x, t, context=ctx, fps=fps, clean_cond=True, gamma=0.5)
            e, r = check(y.cpu(), torch.from_numpy(g[name]), TOL_UNET, f"{tag}.{name}")
            print(f"{tag}.{name}: max-norm rel err {e:.2e}, rel rms {r:.2e}")
    # size-independent properties at the headline shape [.,4,16,40,64] (no oracle needed):
    xs = [inp(f"prop.x{i}", (1, 4, 16, 40, 64)).cuda() for i in range(2)]
    cs = [inp(f"prop.c{i}", (1, 77, 1024)).cuda() for i in range(2)]
    ts = torch.tensor([981, 20]).cuda()
    fps = torch.tensor([10, 10]).cuda()
    single = [m(xs[i], ts[i:i + 1], context=cs[i], fps=fps[i:i + 1]) for i in range(2)]
    both = m(torch.cat(xs), ts, context=torch.cat(cs), fps=fps)
    for i in range(2):      # every op is per sample: a batch of two equals two single launches (other tile counts / split-k: tolerance)
        assert relerr(both[i:i + 1].cpu(), single[i].cpu()) < TOL_UNET, f"batch consistency sample {i}"      # fp16 noise level: 2-3e-3
    runs = [m(torch.cat(xs), ts, context=torch.cat(cs), fps=fps) for _ in range(3)]      # eager / capture / replay
    assert torch.equal(runs[0], runs[1]) and torch.equal(runs[1], runs[2]) and torch.equal(runs[0], both), "replays must be bit-identical"
    # FIFO semantics: a per-frame timestep vector that happens to be constant equals the uniform-timestep call
    t16 = torch.full((16,), 500).cuda()
    a = m(xs[0], t16, context=cs[0], fps=fps[:1])
    b = m(xs[0], torch.tensor([500]).cuda(), context=cs[0], fps=fps[:1])
    assert relerr(a.cpu(), b.cpu()) < 1e-6


def _b16_inputs():
    """the 8 window rows of one FIFO iteration, tiled from the two windows of tests/golden/unet_full_b16.npz (own latents, 16
    consecutive timesteps of the 50-step schedule each), + the 154-token and the 77-token context"""
    g = golden("unet_full_b16")
    wins = ["w0", "w1", "w1", "w0", "w0", "w0", "w1", "w1"]                 # not periodic in 2 or 4: a row permutation shows
    xw = {w: inp(f"full_b16.{w}.x", (1, 4, 16, 40, 64)) for w in ("w0", "w1")}
    x = torch.cat([xw[w] for w in wins]).cuda()
    t = torch.cat([torch.from_numpy(g[f"{w}__t"]) for w in wins]).cuda()
    c154 = inp("full_b16.ctx154", (1, 154, 1024)).cuda()
    c77 = inp("full_b16.ctx77", (1, 77, 1024)).cuda()
    return g, wins, x, t, c154, c77


def test_unet_full_width_b16_vs_reference_golden(full_dm):
    """The operating point of configs[2-4]: ONE B = 16 forward laid out as `FifoEngine` lays it out -- 8 window rows
    with the two-prompt 154-token context followed by the SAME 8 windows with the 77-token unconditional context, per-(b, t)
    timesteps (funcs.py:305-355 -> ddim.py:362-374 -> openaimodel3d.py:535-549).  M = 655 360 rows at the 320-channel level selects
    other kernels than B <= 2 (GEGLU / `+res` kernels switch at M >= 2^17, NT stores on output size), so EVERY row is held to the
    output the REAL reference UNet gave for that (window, context) at B = 1 (`tools/make_golden.py --only fullB16`), through
    (a) the shared-prefix plan the engine uses and (b) the plain two-segment batch it falls back to for unequal fps."""
    m = full_dm.model.diffusion_model
    g, wins, x, t, c154, c77 = _b16_inputs()
    fps = torch.tensor([10]).cuda()
    ctxs = [c154.expand(8, -1, -1).contiguous(), c77.expand(8, -1, -1).contiguous()]
    for it in range(3):                                                      # eager, hipGraph capture, replay
        y = m.forward_segments(x, t, ctxs, fps=fps, shared_x=True)
        assert y.shape == (16, 4, 16, 40, 64)
        if it == 0:
            y0 = y.clone()
    assert torch.equal(y, y0), "replays must be bit-identical"
    worst = 0.0
    for r in range(16):
        name = f"{wins[r % 8]}_{154 if r < 8 else 77}"
        e, _ = check(y[r:r + 1].cpu(), torch.from_numpy(g[name]), TOL_UNET, f"b16 shared-prefix row {r} ({name})")
        worst = max(worst, e)
    print(f"b16 shared prefix: worst row {worst:.2e}")
    # (b) the plain batch: 16 latent rows, two context segments, nothing shared
    y2 = m.forward_segments(torch.cat([x, x]), torch.cat([t, t]), ctxs, fps=fps, shared_x=False)
    for r in range(16):
        name = f"{wins[r % 8]}_{154 if r < 8 else 77}"
        check(y2[r:r + 1].cpu(), torch.from_numpy(g[name]), TOL_UNET, f"b16 plain row {r} ({name})")
    # batch consistency: row i of the B = 16 launch == the B = 1 launch of the same window (other tilings / split-K: tolerance)
    for r, L, ctx in ((0, 154, c154), (1, 154, c154), (9, 77, c77)):
        s = m(x[r % 8:r % 8 + 1], t[(r % 8) * 16:(r % 8) * 16 + 16], context=ctx, fps=fps)
        assert relerr(y[r:r + 1].cpu(), s.cpu()) < TOL_UNET, f"batch consistency row {r}"
    # the same window in other rows: equal to the fp16 noise level, not to the bit -- the weight-stationary kernel sums a statistics
    # group's column sums in a per-group (rotated) strip order, so two rows' GroupNorm statistics differ in the last fp32 bit
    assert relerr(y[1].cpu(), y[2].cpu()) < TOL_UNET and relerr(y[8].cpu(), y[11].cpu()) < TOL_UNET


def test_fifo_engine_full_size_iteration_vs_oracle_step(full_dm):
    """One FULL-SIZE `FifoEngine` iteration (72-frame queue of 40x64 latents, 8 windows x {154, 77} tokens = the B = 16 graph the
    `fifo` / `video` bench legs time) checked beyond `isfinite`:
      * its UNet output rows against the B = 1 launches of the same windows (which the goldens above pin to the reference);
      * guidance + MoCA `ddim_step` (ddim.py:366-372,405-430,556-609) of all 8 windows, in the reference's call order with the momentum
        state carried from call to call, against `oracle.sampler_oracle.ddim_step` fed the HIP eps -- fp32, <= 2e-5;
      * write-back, emitted frame, FreeInit mix and shift (funcs.py:336-371,86-99) against the oracle's queue ops;
      * iterations 2 and 3 (capture, replay) against the host-driven HIP loop on the same draws."""
    import types
    from moca_video_amd.fifo import fifo_ddim_sampling, prepare_latents
    from moca_video_amd.fifo_graph import FifoEngine, fifo_windows
    from moca_video_amd.sampler import DDIMSampler
    from oracle import freeinit_oracle as FO
    from oracle import sampler_oracle as SO
    dm = full_dm
    m = dm.model.diffusion_model
    T, H, W, S = 16, 40, 64, 64
    args = types.SimpleNamespace(num_inference_steps=S, video_length=T, lookahead_denoising=True, num_partitions=4, new_video_length=100)
    s = DDIMSampler(dm)
    s.make_schedule(S, ddim_eta=1.0, verbose=False)
    Q = S + T // 2
    fps = torch.tensor([10]).cuda()
    c1, c2, ucx = (inp(f"full_it.{n}", (1, 77, 1024)).cuda() for n in ("c1", "c2", "uc"))
    cond = {"c_crossattn": [c1, c2], "fps": fps}
    uc = {"c_crossattn": [ucx], "fps": fps}
    prep = [inp(f"full_it.prep{j}", (1, 4, 1, H, W)).cuda() for j in range(Q)]
    lat0 = prepare_latents(args, None, s, initial_latents=inp("full_it.z", (1, 4, T, H, W)).cuda(), noises=prep)
    mask = torch.zeros(1, 1, Q, H, W)
    mask[..., H // 4: 3 * H // 4, W // 4: 3 * W // 4] = 1.0
    mask[:, :, 5] = 0.0
    cimg = (inp("full_it.cimg", (1, 4, 1, H, W)) * 0.25 + 0.5).clamp(0, 1)
    n_it = 3
    noises = [[inp(f"full_it.n{i}.{w}", (1, 4, T, H, W)) for w in range(8)] for i in range(n_it)]
    shifts = [inp(f"full_it.sh{i}", (1, 4, H, W)) for i in range(n_it)]
    eng = FifoEngine(args, dm, s, cond, uc, 12.0, lat0.clone(), conditioned_image=cimg.cuda(), masks=mask.cuda(), n_slots=n_it)
    assert eng.plan.B == 16 and eng.nW == 8
    eng.step(noise=[n.cuda() for n in noises[0]], shift_noise=shifts[0].cuda())
    eng.sync_to()
    torch.cuda.synchronize()
    eps = eng.plan.out.reshape(16, 4, T, H, W).float().clone()              # [cond windows | uncond windows], reference call order
    xp, p0 = [[v.clone() for v in vs] for vs in eng.window_outputs()]
    q1, em0 = eng.latents().clone(), eng.emitted_frames(0, 1).clone()
    # ---- (1) the engine's eps rows == B = 1 launches of the same window
    wins = list(fifo_windows(args))
    ts_all = np.concatenate([np.full((T // 2,), s.ddim_timesteps[0]), s.ddim_timesteps])
    idx_all = np.concatenate([np.full((T // 2,), 0), np.arange(S)])
    cc = torch.cat([c1, c2], 1)
    for w in (0, 3, 7):
        s0, _, e0 = wins[w]
        xw = lat0[:, :, s0:e0].contiguous()
        tw = torch.as_tensor(ts_all[s0:e0].copy()).long().cuda()
        e_c = m(xw, tw, context=cc, fps=fps)
        e_u = m(xw, tw, context=ucx, fps=fps)
        assert relerr(eps[w:w + 1].cpu(), e_c.cpu()) < TOL_UNET, f"window {w} cond eps"
        assert relerr(eps[8 + w:9 + w].cpu(), e_u.cpu()) < TOL_UNET, f"window {w} uncond eps"
    # ---- (2) guidance + ddim_step of the 8 windows + write-back + shift on the oracle, from the HIP eps
    sch = SO.make_schedule(SO.ddpm_buffers(), S, 1.0)
    lat, msk = lat0.cpu().clone(), mask.clone()
    mom = torch.zeros(1, 4, T, H, W)
    eps_c = eps.cpu()
    for w, (s0, mid, e0) in enumerate(wins):
        x = lat[:, :, s0:e0].clone()
        e = eps_c[8 + w:9 + w] + 12.0 * (eps_c[w:w + 1] - eps_c[8 + w:9 + w])                      # ddim.py:372
        t = torch.as_tensor(ts_all[s0:e0].copy()).long()
        out, px0 = SO.ddim_step(sch, x, e, idx_all[s0:e0], cimg, t, [noises[0][w][:, :, [k]] for k in range(T)], mom,
                                davis_masks=msk[:, :, s0:e0].clone())
        assert relerr(xp[w].cpu(), out) < 2e-5, f"window {w} x_prev"
        assert relerr(p0[w].cpu(), px0) < 2e-5, f"window {w} pred_x0"
        lat[:, :, mid:e0] = out[:, :, -(T // 2):]
    assert relerr(em0.cpu(), lat[:, :, [T // 2]]) < 2e-5, "emitted frame"
    lat = FO.shift_latents(lat, shifts[0])
    assert relerr(q1.cpu(), lat) < 2e-5, "queue after write-back, FreeInit mix and shift"
    # ---- (3) capture + replay vs the host-driven HIP loop
    for i in (1, 2):
        eng.step(noise=[n.cuda() for n in noises[i]], shift_noise=shifts[i].cuda())
    assert eng.plan.graph is not None, "the iteration was not captured into a hipGraph"
    q3, em = eng.latents().clone(), eng.emitted_frames(0, n_it).clone()
    eng.close()
    lat_h, m_h = lat0.clone(), mask.cuda().clone()
    fr_h = fifo_ddim_sampling(args, dm, cond, (1, 4, T, H, W), s, cfg_scale=12.0, uc_emb=ucx, latents=lat_h, conditioned_image=cimg.cuda(),
                              masks=m_h, n_iterations=n_it, noises=[[n.cuda() for n in row] for row in noises],
                              shift_noises=[x.cuda() for x in shifts], use_graph=False)
    assert relerr(q3, lat_h) < 3e-2          # CFG 12 x the fp16 UNet noise, 3 iterations fed back (tests/test_loops_gpu.py: TOL_FIFO)
    for i in range(n_it):
        assert relerr(em[:, :, [i]], fr_h[i]) < 3e-2, f"emitted frame {i}"


def test_fifo_engine_full_size_prompt_mode_vs_oracle_step(full_dm):
    """The full-size iteration in PROMPT mode (what the `fifo_prompt_mode` / `video` bench legs time): no masks handed in, `ddim_step`
    takes its segmentation branch (ddim.py:592-606 -> `_apply_segmentation` :739-903: t <= 300 only, previous-mask / IoU < 0.5 fallbacks,
    > 80 % reset, factor 2) on scripted per-frame candidate masks, bookkeeping on the device inside the iteration graph: x_prev / pred_x0 of
    the 8 windows against `oracle.sampler_oracle.ddim_step(sam_masks=...)` on the HIP eps, two iterations (the second one replays nothing
    yet but moves the queue: other windows reach t <= 300)."""
    import types
    from helpers import loop_sam_candidates
    from moca_video_amd.fifo import prepare_latents
    from moca_video_amd.fifo_graph import FifoEngine, fifo_windows
    from moca_video_amd.sampler import DDIMSampler
    from oracle import sampler_oracle as SO
    dm = full_dm
    T, H, W, S = 16, 40, 64, 64
    args = types.SimpleNamespace(num_inference_steps=S, video_length=T, lookahead_denoising=True, num_partitions=4, new_video_length=100)
    s = DDIMSampler(dm)
    s.make_schedule(S, ddim_eta=1.0, verbose=False)
    Q = S + T // 2
    fps = torch.tensor([10]).cuda()
    c1, c2, ucx = (inp(f"full_it.{n}", (1, 77, 1024)).cuda() for n in ("c1", "c2", "uc"))
    cond, uc = {"c_crossattn": [c1, c2], "fps": fps}, {"c_crossattn": [ucx], "fps": fps}
    prep = [inp(f"full_it.prep{j}", (1, 4, 1, H, W)).cuda() for j in range(Q)]
    lat0 = prepare_latents(args, None, s, initial_latents=inp("full_it.z", (1, 4, T, H, W)).cuda(), noises=prep)
    cimg = (inp("full_it.cimg", (1, 4, 1, H, W)) * 0.25 + 0.5).clamp(0, 1)
    eng = FifoEngine(args, dm, s, cond, uc, 12.0, lat0.clone(), conditioned_image=cimg.cuda(), n_slots=2, sam_capacity=2 * 8 * T)
    wins = list(fifo_windows(args))
    ts_all = np.concatenate([np.full((T // 2,), s.ddim_timesteps[0]), s.ddim_timesteps])
    idx_all = np.concatenate([np.full((T // 2,), 0), np.arange(S)])
    sch = SO.make_schedule(SO.ddpm_buffers(), S, 1.0)
    mom = torch.zeros(1, 4, T, H, W)
    injected = 0
    for it in range(2):
        noises = [inp(f"full_pm.n{it}.{w}", (1, 4, T, H, W)) for w in range(8)]
        shift = inp(f"full_pm.sh{it}", (1, 4, H, W))
        cands = [loop_sam_candidates(8 * it + w, T, H, W) for w in range(8)]
        lat = eng.latents().cpu().clone()                       # the queue this iteration starts from
        eng.step(noise=[n.cuda() for n in noises], shift_noise=shift.cuda(), sam_masks=cands)
        eng.sync_to()
        torch.cuda.synchronize()
        eps = eng.plan.out.reshape(16, 4, T, H, W).float().cpu()
        xp, p0 = eng.window_outputs()
        for w, (s0, mid, e0) in enumerate(wins):
            x = lat[:, :, s0:e0].clone()
            e = eps[8 + w:9 + w] + 12.0 * (eps[w:w + 1] - eps[8 + w:9 + w])
            t = torch.as_tensor(ts_all[s0:e0].copy()).long()
            out, px0 = SO.ddim_step(sch, x, e, idx_all[s0:e0], cimg, t, [noises[w][:, :, [k]] for k in range(T)], mom, sam_masks=cands[w])
            assert relerr(xp[w].cpu(), out) < 2e-5, f"iteration {it} window {w} x_prev"
            assert relerr(p0[w].cpu(), px0) < 2e-5, f"iteration {it} window {w} pred_x0"
            lat[:, :, mid:e0] = out[:, :, -(T // 2):]
        injected += int((eng.sam_idx >= 0).sum())
    assert injected > 0, "no window frame took the segmentation branch: the test would not see it"
    eng.close()


def test_reloading_weights_releases_the_old_graphs(reduced_model):
    """`load_state_dict` / `.to()` drop the recorded plans: their instantiated hipGraphs must be destroyed and the
    next forward must re-record and give the same result."""
    g = golden("unet_reduced")
    x = inp("reduced.uniform.x", (1, 4, 8, 16, 16)).cuda()
    ctx = inp("reduced.uniform.ctx", (1, 77, 128)).cuda()
    t = torch.from_numpy(g["uniform__t"]).cuda()
    for _ in range(3):
        y0 = reduced_model(x, t, context=ctx, fps=16)
    old = list(reduced_model._plans.values())      # (the fixture is shared: plans other tests ran only once have no graph yet)
    assert any(p.graph is not None for p in old) and reduced_model._plan_for(x, ctx.shape[1]).graph is not None
    reduced_model.load_state_dict(state_dict_for(reduced_model, 11), strict=True)
    assert not reduced_model._plans and all(p.graph is None for p in old), "old plans must have released their graphs"
    for _ in range(3):
        y1 = reduced_model(x, t, context=ctx, fps=16)
    assert torch.equal(y0, y1)
