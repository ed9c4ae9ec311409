#!/usr/bin/env python3
"""bench.py -- denoising UNet-steps/s of the MI355X-native MoCA-Video hot path.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Workload (BASELINE.json configs[1]): VideoCrafter2 3D-UNet, 16 frames x 320x512
(latents [1,4,16,40,64]), fp16 storage / fp32 accumulate, one prompt (77 context tokens),
classifier-free guidance 12.0, DDIM eta 1.0, S=50 schedule, random-init weights of the real
architecture (1.41 B parameters), synthetic latents/context (seed 321 + rank).

One bench "step" = ONE DDIM step of that sampler = `DDIMSampler.p_sample_ddim`:
  2 UNet-steps (cond + uncond, evaluated as one batched [2,4,16,40,64] forward)
  + CFG combine + DDIM update with use_scale + fresh Gaussian noise.
`value` = UNet-steps/s over all ranks = 2 * K * N / t, 1 UNet-step = one
DiffusionWrapper.forward on [1,4,16,40,64] = 12.581 TFLOP.
Multi-GPU: independent prompt/seed per rank (weak scaling), weights broadcast from rank 0
over RCCL before the timed region, result latents gathered after it; no collective inside.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_PROMPTS = 64                      # BASELINE.json configs[4]: 64 rows of prompts/prompts.csv, sharded over the ranks
PROMPTS_PER_FORWARD = 8             # x 2 CFG branches = one B = 16 forward
FLOP_PER_UNET_STEP = 12.581e12      # FlopCounterMode on the reference UNet, [1,4,16,40,64], L=77
PEAK_F16_MFMA_TFLOPS = 2500.0       # MI355X dense fp16 MFMA (MI355X_MICROARCH.md)
# L2<->fabric bytes of one batched (B=2) UNet forward launch come from separate rocprofv3 --pmc FETCH_SIZE and
# --pmc WRITE_SIZE passes over this same command (tools/profile_round.sh -> tools/pmc_traffic_summary.py; FETCH_SIZE
# doubled per the gfx950 correction in MI355X_MICROARCH.md; Infinity-Cache hits are included in these counters).  The
# figure is READ from the committed summary of the current kernels -- never a constant in this file -- and the JSON
# names the file; it is null when no summary for this round exists.
TRAFFIC_PROFILES = ("profiles/r06_pmc_traffic_per_forward.txt", "profiles/r05_pmc_traffic_per_forward.txt")


def traffic_from_profile():
    import re
    for rel in TRAFFIC_PROFILES:
        path = os.path.join(ROOT, rel)
        if os.path.exists(path):
            m = re.search(r"=\s*([0-9.]+)\s*GB per UNet forward launch", open(path).read())
            if m:
                return float(m.group(1)) * 1e9, rel
    return None, None


FULL = dict(in_channels=4, out_channels=4, model_channels=320, attention_resolutions=[4, 2, 1], num_res_blocks=2,
            channel_mult=[1, 2, 4, 4], num_head_channels=64, transformer_depth=1, context_dim=1024, use_linear=True,
            use_checkpoint=True, temporal_conv=True, temporal_attention=True, temporal_selfatt_only=True,
            use_relative_position=False, use_causal_attention=False, temporal_length=16, addition_attention=True,
            fps_cond=True)


def build_model(device, seed=321, materialise=True):
    """materialise=False (ranks != 0 of a multi-GPU job): the parameters are allocated and filled with NaN -- they exist only
    once rank 0's copy has arrived through the C1 broadcast; a broadcast that did not happen cannot go unnoticed."""
    from moca_video_amd import DenoiseModel
    from moca_video_amd.weightgen import init_random_
    with torch.device(device):
        dm = DenoiseModel({"target": "lvdm.modules.networks.openaimodel3d.UNetModel", "params": FULL})
    dm = dm.to(device)
    if materialise:
        init_random_(dm.model.diffusion_model, seed)
    else:
        with torch.no_grad():
            for p in dm.model.diffusion_model.parameters():
                p.fill_(float("nan"))
        dm.model.diffusion_model._invalidate()
    return dm


def tensor_checksum(tensors, device):
    """(sum, sum of squares) over the tensors in float64, each tensor weighted by its POSITION in the list: identical bytes in the
    identical order give identical values (same reduction on every rank); a NaN-filled (never received) tensor makes it NaN, which
    equals nothing; two same-shape operands that a receiver enumerated in a different order (the data then sits in the wrong buffers)
    change both values -- an unweighted sum would not see that."""
    s = torch.zeros(2, dtype=torch.float64, device=device)
    with torch.no_grad():
        for i, p in enumerate(tensors):
            d = p.detach().double()
            w = 1.0 + (i % 1021) / 1021.0
            s[0] += w * d.sum()
            s[1] += w * (d * d).sum()
    return s


def broadcast_and_verify(module, device, world, rank, plans):
    """C1 with proof: barrier, timed flat-bucket broadcast from rank 0 (RCCL over xGMI; gloo in the CPU self-test) of the PACKED operand
    set the recorded launches of `plans` read (dist.broadcast_packed: fp16 weights + fp32 biases / norm parameters, 2.83 GB for the
    UNet; the receivers built the same plans from placeholder parameters, get the data in place and drop their fp32 masters), then
    the per-rank checksums of those packed buffers are all-gathered and must all equal rank 0's.  Returns the dict that goes into
    the JSON line; raises (non-zero exit on every rank) when a rank's operands differ."""
    import torch.distributed as tdist
    from moca_video_amd import dist as mdist
    if not tdist.is_initialized():                                # (one process, no launcher: nothing to prove)
        ts = mdist.packed_operands(module._packed, plans)
        return {"rccl_ranks": 1, "backend": None, "broadcast_bytes": 0, "broadcast_s": 0.0, "broadcast_GBps_per_receiver": None,
                "param_checksums_equal": True, "param_checksum_rank0": [float(v) for v in tensor_checksum(ts, device)]}
    sync = (lambda: torch.cuda.synchronize(device)) if device.type == "cuda" else (lambda: None)
    mdist.barrier(); sync()
    t0 = time.perf_counter()
    if os.environ.get("MOCA_BENCH_STUB_BROADCAST") == "1":       # negative test hook (tests/test_dist_cpu.py): C1 skipped
        nbytes, tensors = 0, mdist.packed_operands(module._packed, plans)
    else:
        nbytes, tensors = mdist.broadcast_packed(module, plans, src=0)
    sync(); mdist.barrier()
    dt = mdist.max_over_ranks(time.perf_counter() - t0, device)
    cs = tensor_checksum(tensors, device)
    allcs = [torch.empty_like(cs) for _ in range(world)]
    tdist.all_gather(allcs, cs)
    sums = [[float(c[0]), float(c[1])] for c in allcs]
    equal = all(torch.equal(c, allcs[0]) for c in allcs) and bool(torch.isfinite(allcs[0]).all())
    info = {"rccl_ranks": tdist.get_world_size(), "backend": tdist.get_backend(), "broadcast_bytes": int(nbytes),
            "broadcast_what": "packed operand set (fp16 weights + fp32 biases / norm parameters), %d tensors" % len(tensors),
            "fp32_masters_on_this_rank_bytes": int(sum(p.numel() * p.element_size() for p in module.parameters())),
            "broadcast_s": round(dt, 4), "broadcast_GBps_per_receiver": round(nbytes / dt / 1e9, 2) if dt > 0 else None,
            "param_checksums_equal": equal, "param_checksum_rank0": sums[0]}
    if not equal:
        raise RuntimeError(f"rank {rank}: packed-operand checksums differ after the C1 broadcast: {sums}")
    return info


def cpu_model_name():
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or platform.machine()


def cpu_baseline(dm, x, ctx, ts, threads, runs=3):
    """The CPU oracle (fp32 restatement of the reference UNet, oracle/unet_oracle.py) on the host cores, SURVEY 8(d) protocol:
    ONE UNet-step at the full [1,4,16,40,64] shape per run (bounded sample, 15-20 s each), 1 untimed warm-up run (it also pages the
    weights in) + `runs` timed runs, the MEDIAN is reported."""
    from oracle import unet_oracle as UO
    torch.set_num_threads(threads)
    sd = {k: v.detach().float().cpu() for k, v in dm.model.diffusion_model.state_dict().items()}
    xc, cc, tc = x.float().cpu(), ctx.float().cpu(), ts.cpu()
    times = []
    with torch.no_grad():
        y = UO.unet_forward(sd, xc, tc, cc, fps=torch.tensor([10]))          # warm-up
        for _ in range(runs):
            t0 = time.perf_counter()
            y = UO.unet_forward(sd, xc, tc, cc, fps=torch.tensor([10]))
            times.append(time.perf_counter() - t0)
    return sorted(times)[len(times) // 2], y, times


def plan_flops(plan):
    """FLOPs the recorded launch sequence of a plan actually EXECUTES (2 M N K of every GEMM / implicit-GEMM launch with its packed,
    i.e. zero-padded, N and the K it really walks -- 4 C per `Upsample` phase, the shared guidance prefix at half batch -- plus
    4 Nq Nk d per attention head): what the matrix pipe is busy with, as opposed to the reference's algorithmic count."""
    from moca_video_amd import ops
    total = 0.0
    for st in plan.steps:
        fn, kw = getattr(st, "func", None), getattr(st, "keywords", {})
        if fn is ops.gemm:
            pw = st.args[1]
            total += 2.0 * kw["M"] * pw.N * pw.K
            if kw.get("tattn") is not None:                       # + the temporal attention finished in the epilogue
                T, HW, _ = kw["tattn"]
                total += 4.0 * kw["M"] * T * 64 * (pw.N // 192)
        elif fn is ops.attention:
            total += 4.0 * kw["Bq"] * kw["heads"] * kw["Nq"] * kw["Nk"] * 64
        elif fn is ops.temporal_attention:
            total += 4.0 * kw["B"] * kw["HW"] * kw["heads"] * kw["T"] * kw["T"] * 64
    return total


class ZeroDataDenoiser:
    """Weights for the FIFO / whole-video legs.  With random-init weights the noise prediction has nothing to do with the noise in
    x; pred_x0 of neighbouring queue frames is then incoherent, the MoCA momentum term (ddim.py:422-430: pred_x0 += 2 (1 - t/1000) m,
    m an EMA of frame-to-frame pred_x0 differences) amplifies every frame by ~1.2 per iteration, the latents reach ~1e5 when the
    first enqueued noise frames arrive at the clean end of the queue (iteration ~50, `tools/diag_video_finite.py`) and the fp16
    UNet overflows: round 2's video leg ran two thirds of its iterations on NaN -- operand data that clocks higher than real data.
    Scaling the output down does not help (eps ~ 0 is just as incoherent).  The only bounded fixed point that needs no trained
    weights is the exact denoiser of an all-zero dataset, eps = x / sqrt(1 - abar_t) (pred_x0 = 0 for every frame: coherent):
      conv_in copies latent channel c mod 4 to channel c (centre tap); `init_attn.proj_out` and, in the LAST output block only, the
      ResBlock's second conv / temporal conv4 / the two transformers' proj_out (all zero-initialised in the reference,
      openaimodel3d.py:177,266-267, attention.py:256-258,326-328) are scaled by 2^-10; that block's 1x1 skip conv passes the
      conv_in half of the concat through (2^-10 x random on the other half); out = GroupNorm(gamma 1, beta 10) -> SiLU (~identity
      at 10 +- 6) -> a centre-tap conv with bias -10.  The final GroupNorm divides by the per-frame standard deviation of x,
      which IS sqrt(1 - abar_t) for pure-noise latents.
    9 of the 1484 tensors are touched; the other ~1.4 B parameters stay random at full scale, see ordinary GroupNorm-ed
    activations and execute every FLOP.  `restore()` puts the headline weights back (the headline `value` never sees this)."""

    def __init__(self, dm):
        self.u = u = dm.model.diffusion_model
        rb, st, tt = u.output_blocks[-1][0], u.output_blocks[-1][1], u.output_blocks[-1][2]
        self.small = [u.init_attn[0].proj_out, rb.out_layers[3], rb.temopral_conv.conv4[3], st.proj_out, tt.proj_out]
        self.crafted = [u.input_blocks[0][0], rb.skip_connection, u.out[0], u.out[2]]
        self.saved = [(m, m.weight.detach().clone(), m.bias.detach().clone()) for m in self.small + self.crafted]
        with torch.no_grad():
            for m in self.small:
                m.weight.mul_(2.0 ** -10)
                m.bias.mul_(2.0 ** -10)
            cin, skip, gn, cout = self.crafted
            ch = cin.weight.shape[0]
            cin.weight.zero_(); cin.bias.zero_()
            for c in range(ch):
                cin.weight[c, c % 4, 1, 1] = 1.0
            skip.weight.mul_(2.0 ** -10); skip.bias.zero_()
            k_in = skip.weight.shape[1]
            skip.weight[:, k_in - ch:].zero_()
            for c in range(ch):
                skip.weight[c, k_in - ch + c, 0, 0] = 1.0
            gn.weight.fill_(1.0); gn.bias.fill_(10.0)
            cout.weight.zero_(); cout.bias.fill_(-10.0)
            for k in range(cout.weight.shape[0]):
                cout.weight[k, k, 1, 1] = 1.0
        u._invalidate()

    def restore(self):
        with torch.no_grad():
            for m, w, b in self.saved:
                m.weight.copy_(w)
                m.bias.copy_(b)
        self.u._invalidate()


def synthetic_sam_candidates(T, H, W):
    """What a Grounded-SAM-2 producer could return for the frames of window `w` in iteration `i` (the producer is outside the path,
    SURVEY 8c: masks are synthetic inputs): a box drifting with the queue position, every 7th frame no detection (-> previous
    masks), every 5th a jump (IoU < 0.5 -> previous masks), every 11th a second small mask.  Built on the host once per shape."""
    import functools

    @functools.lru_cache(maxsize=None)
    def box(y0, x0, h, w_):
        m = torch.zeros(H, W)
        m[y0 % (H - h):y0 % (H - h) + h, x0 % (W - w_):x0 % (W - w_) + w_] = 1.0
        return m

    def cands(i, w):
        out = []
        for j in range(T):
            k = i + 8 * w + j                       # ~ the frame's age in the queue
            if k % 7 == 3:
                out.append(None)
            elif k % 5 == 4:
                out.append(box(k, 3 * k, H // 2, W // 2)[None])
            elif k % 11 == 0:
                out.append(torch.stack([box(H // 4 + k // 16, W // 4 + k // 8, H // 2, W // 2), box(2, 2 + k, H // 8, W // 8)]))
            else:
                out.append(box(H // 4 + k // 16, W // 4 + k // 8, H // 2, W // 2)[None])
        return out
    return cands


def fifo_leg(dm, device, T, H, W, iters=10, mode="masks", weights="zero-data"):
    """Extra (not the headline value): one outer iteration of the MoCA FIFO loop (configs[2-3]) at full size --
    8 diagonal windows x {cond (2 prompts = 154 tokens), uncond (77)} = 16 UNet-steps as ONE batched forward with two context
    segments, + noise, window gather, guidance, the MoCA ddim_step of the 8 windows with mask injection, write-back, emission,
    FreeInit mix and queue shift: the whole iteration is one hipGraph (fifo_graph.FifoEngine).  Timed with HIP events on the
    engine's stream around `iters` replays."""
    import types
    from moca_video_amd.fifo_graph import FifoEngine
    from moca_video_amd.sampler import DDIMSampler
    lib = __import__("moca_video_amd.lib", fromlist=["load"]).load()
    args = types.SimpleNamespace(num_inference_steps=64, video_length=T, lookahead_denoising=True, num_partitions=4,
                                 new_video_length=100)
    s = DDIMSampler(dm)
    s.make_schedule(64, ddim_eta=1.0, verbose=False)
    g = torch.Generator(device=device).manual_seed(7)
    Q = 64 + T // 2
    fps = torch.tensor([10], device=device)
    cond = {"c_crossattn": [torch.randn(1, 77, 1024, device=device, generator=g), torch.randn(1, 77, 1024, device=device, generator=g)],
            "fps": fps}
    uc = {"c_crossattn": [torch.randn(1, 77, 1024, device=device, generator=g)], "fps": fps}
    from moca_video_amd.fifo import prepare_latents
    lat = prepare_latents(args, None, s, initial_latents=torch.randn(1, 4, T, H, W, device=device, generator=g))
    mask = torch.zeros(1, 1, Q, H, W, device=device)
    mask[..., H // 4: 3 * H // 4, W // 4: 3 * W // 4] = 1.0
    cimg = torch.rand(1, 4, 1, H, W, device=device, generator=g)
    # mode "masks": injection masks handed in (`davis_masks` branch of ddim_step, ddim.py:565-590: factor 1.5 / 1.0, every timestep);
    # mode "prompt": no masks -> the segmentation branch (ddim.py:592-606 -> :739-903: t <= 300 only, previous-mask / IoU fallbacks,
    # > 80 % reset, factor 2) on per-iteration candidate masks, bookkeeping on the device inside the same graph
    sam = synthetic_sam_candidates(T, H, W) if mode == "prompt" else None
    eng = FifoEngine(args, dm, s, cond, uc, 12.0, lat, conditioned_image=cimg, masks=None if sam else mask, n_slots=8, seed=7,
                     sam_capacity=2 * 8 * T if sam else 0)
    nW = eng.nW
    it = [0]

    def step():
        eng.step(sam_masks=[sam(it[0], w) for w in range(nW)] if sam else None)
        it[0] += 1
    for _ in range(3):          # eager, hipGraph capture, first replay
        step()
    torch.cuda.synchronize()
    a, b = C.c_void_p(), C.c_void_p()
    lib.moca_event_create(C.byref(a)); lib.moca_event_create(C.byref(b))
    h = C.c_void_p(eng.plan.stream.cuda_stream)
    t0 = time.perf_counter()
    lib.moca_event_record(a, h)
    for _ in range(iters):
        step()
    lib.moca_event_record(b, h)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / iters
    ms = C.c_float()
    lib.moca_event_elapsed_ms(a, b, C.byref(ms))
    lib.moca_event_destroy(a); lib.moca_event_destroy(b)
    dt = ms.value * 1e-3 / iters
    finite = bool(torch.isfinite(eng.latents()).all())
    graph_on = eng.plan.graph is not None
    n_launch = len(eng.plan.steps) - 2 + 12 + (1 if sam else 0)
    injected = int((eng.sam_idx >= 0).sum()) if sam else None
    eng.close()
    if sam:
        return {"iteration_ms": round(dt * 1e3, 2), "iteration_wall_ms": round(wall * 1e3, 2), "unet_steps_per_s": round(16 / dt, 2),
                "one_hipgraph_per_iteration": graph_on, "launches_per_iteration": n_launch, "queue_finite": finite,
                "window_frames_injected_last_iteration": injected,
                "note": "PROMPT mode (the reference's default driver mode): no masks handed in, ddim_step takes its segmentation branch "
                        "(t <= 300 only, previous-mask / IoU < 0.5 fallbacks, > 80 % reset, factor 2) on synthetic per-iteration candidate "
                        "masks uploaded by the host (pinned double buffer, no sync); bookkeeping + injection inside the same hipGraph"}
    return {"iteration_ms": round(dt * 1e3, 2), "iteration_wall_ms": round(wall * 1e3, 2), "unet_steps_per_iteration": 16,
            "unet_steps_per_s": round(16 / dt, 2), "projected_s_per_video_148_iterations": round(148 * dt, 1),
            "one_hipgraph_per_iteration": graph_on, "launches_per_iteration": n_launch, "queue_finite": finite,
            "note": ("synthetic zero-data weights (ZeroDataDenoiser)" if weights == "zero-data" else
                     "RANDOM-INIT weights (the headline's): the same iteration graph on the operand statistics the headline runs on -- only a "
                     "few iterations are bounded on them (see ZeroDataDenoiser), hence 5 timed ones behind 3") +
                    "; injection masks handed in (`masks=`: the davis_masks branch of "
                    "ddim_step); 8 windows x (154-token cond + 77-token uncond) as ONE B=16 forward with two context segments; noise, "
                    "gather, guidance, MoCA ddim_step x 8, write-back, emission, FreeInit mix, shift in the same hipGraph; VAE decode excluded"}


def video_leg(dm, ae, device, T, H, W):
    """Extra: MEASURED wall-clock of one whole video (second half of BASELINE.json's metric), prompt mode of
    videocrafter_main.py:176-232 at full size -- base sampling (64 CFG DDIM steps, funcs.py:177-241, + decode of its 16
    frames), queue construction (prepare_latents), 148 outer MoCA-FIFO iterations (one hipGraph each: 8 batched windows, cond =
    2 prompts, mask injection, FreeInit shift) and the VAE decode of the 148 emitted frames.  Text encoding / Grounded-SAM-2
    are inputs (out of scope); file writing excluded.  Every emitted frame must be finite (asserted)."""
    import types
    from moca_video_amd.fifo import base_ddim_sampling, fifo_ddim_sampling, prepare_latents
    args = types.SimpleNamespace(num_inference_steps=64, video_length=T, lookahead_denoising=True, num_partitions=4,
                                 new_video_length=100)
    dm.first_stage_model, dm.scale_factor = ae, 0.18215
    # the emitted latents of this synthetic run are bounded but large (up to ~4e3, see ZeroDataDenoiser / tools/diag_video_finite.py):
    # the random-init decoder would overflow fp16 on z / 0.18 ~ 2e4.  post_quant_conv (a 4 -> 4 channel 1 x 1 conv in front of the
    # decoder, autoencoder.py:105) is scaled by 2^-12 for this leg (exact, undone below); the decoder's GroupNorms see the same
    # normalised activations and every FLOP still runs
    with torch.no_grad():
        ae.post_quant_conv.weight.mul_(2.0 ** -12); ae.post_quant_conv.bias.mul_(2.0 ** -12)
    ae._invalidate()
    g = torch.Generator(device=device).manual_seed(9)
    c1, c2, uc_emb = (torch.randn(1, 77, 1024, device=device, generator=g) for _ in range(3))
    fps = torch.tensor([10], device=device)
    cimg = torch.rand(1, 4, 1, H, W, device=device, generator=g)
    shape = [1, 4, T, H, W]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    base, sampler, samples = base_ddim_sampling(dm, {"c_crossattn": [c1], "fps": fps}, shape, 64, 1.0, 12.0, uc_emb=uc_emb)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    lat = prepare_latents(args, None, sampler, initial_latents=samples)
    # prompt mode, as videocrafter_main.py drives it: no masks -> ddim_step's segmentation branch on candidate masks (synthetic: the
    # Grounded-SAM-2 producer is outside the path), inside the iteration graph
    frames = fifo_ddim_sampling(args, dm, {"c_crossattn": [c1, c2], "fps": fps}, shape, sampler, cfg_scale=12.0, uc_emb=uc_emb,
                                latents=lat, conditioned_image=cimg, sam_masks=synthetic_sam_candidates(T, H, W), sam_capacity=2 * 8 * T,
                                decode=True, batch_windows=True, seed=9)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    n_finite = sum(int(bool(torch.isfinite(f).all())) for f in frames)
    dm.first_stage_model = None
    with torch.no_grad():
        ae.post_quant_conv.weight.mul_(2.0 ** 12); ae.post_quant_conv.bias.mul_(2.0 ** 12)
    ae._invalidate()
    res = {"video_s": round(t2 - t0, 2), "base_sampling_s": round(t1 - t0, 2), "fifo_148_iterations_incl_decode_s": round(t2 - t1, 2),
           "unet_steps": 2 * 64 + 148 * 16, "frames_decoded": 16 + 148, "frames_emitted": len(frames),
           "frame_shape": list(frames[0].shape), "frames_finite": n_finite, "base_latents_finite": bool(torch.isfinite(samples).all()),
           "note": "synthetic zero-data weights (random-init with 9 tensors set so that the UNet is the exact denoiser of an all-zero "
                   "dataset: bounded MoCA momentum dynamics, see ZeroDataDenoiser; operand statistics, hence clocks, are not those of a "
                   "trained model); measured, one prompt, 1 GPU, PROMPT mode: 64 CFG base steps + prepare_latents + 148 FIFO iterations "
                   "(one hipGraph each: 8 batched windows, 154-token cond / 77-token uncond, segmentation-branch injection on synthetic "
                   "candidate masks, FreeInit shift; engine construction + capture included) + VAE decode of every emitted frame"}
    assert n_finite == len(frames) and res["base_latents_finite"], f"non-finite frames in the measured video: {res}"
    return res


VAE_DD = dict(double_z=True, z_channels=4, resolution=512, in_channels=3, out_ch=3, ch=128, ch_mult=[1, 2, 4, 4],
              num_res_blocks=2, attn_resolutions=[], dropout=0.0)          # configs/inference_t2v_512_v2.0.yaml:56-70
VAE_FLOP_PER_FRAME = 1.5635e12      # conv/linear/bmm MACs x 2 of AutoencoderKL.decode on [1,4,40,64] (FlopCounterMode on the oracle)


def emulate_world_leg(dm, sampler, device, T, H, W, world=8, steps=5):
    """configs[4] on ONE GPU of an N-GPU node: rank 0's prompt rows (64 / N, strided as videocrafter_main.py:181) as B = 16 forwards
    of the one-graph DDIM step, no collective (there is none in the loop).  The N-GPU job's whole-job rate is N x this per-GPU rate
    minus the start-up broadcast (`broadcast_bytes`, once per job) -- stated as a projection, never as a measurement."""
    from moca_video_amd import dist as mdist
    from moca_video_amd.fifo_graph import BaseEngine
    rows = mdist.shard_indices(N_PROMPTS, 0, world)
    engines = []
    for i in range(0, len(rows), PROMPTS_PER_FORWARD):
        xs, cs, us = [], [], []
        for r in rows[i:i + PROMPTS_PER_FORWARD]:
            g = torch.Generator(device=device).manual_seed(321 + r)
            xs.append(torch.randn(1, 4, T, H, W, device=device, generator=g))
            cs.append(torch.randn(1, 77, 1024, device=device, generator=g))
            us.append(torch.randn(1, 77, 1024, device=device, generator=g))
        fps = torch.tensor([10] * len(xs), device=device)
        engines.append(BaseEngine(dm, sampler, torch.cat(xs), {"c_crossattn": [torch.cat(cs)], "fps": fps},
                                  {"c_crossattn": [torch.cat(us)], "fps": fps}, 12.0, seed=321 + i))

    def step():
        cur = torch.cuda.current_stream(device)
        for e in engines:
            e.step()
            cur.wait_stream(e.plan.stream)
    for _ in range(2):                                   # eager pass + capture pass
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    finite = all(bool(torch.isfinite(e.latents()).all()) for e in engines)
    packed = mdist.packed_operands(dm.model.diffusion_model._packed, [e.plan for e in engines])
    nbytes = int(sum(t.numel() * t.element_size() for t in packed))
    per_gpu = 2 * len(rows) * steps / dt
    del engines
    torch.cuda.empty_cache()
    return {"world_emulated": world, "prompt_rows_on_this_gpu": len(rows), "forwards_per_step": -(-len(rows) // PROMPTS_PER_FORWARD),
            "batch_per_forward": 2 * min(len(rows), PROMPTS_PER_FORWARD), "steps": steps, "ms_per_ddim_step": round(dt / steps * 1e3, 2),
            "unet_steps_per_s_this_gpu": round(per_gpu, 2), "output_finite": finite,
            "broadcast_bytes": nbytes, "projected_whole_job_unet_steps_per_s": round(world * per_gpu, 1),
            "note": "measured on ONE GPU: rank 0's share of BASELINE configs[4] (64 prompts strided over %d GPUs, 16x320x512, DDIM S=50, CFG 12) as "
                    "one-hipGraph DDIM steps; no collective exists in the loop, so the %d-GPU job runs %d such independent streams: the projection "
                    "is %d x this rate and excludes the one-time start-up broadcast of `broadcast_bytes` packed operands (C1) and the final "
                    "gather (C2); NOT a multi-GPU measurement" % (world, world, world, world)}


def vae_leg(device, H, W, frames=8, iters=5):
    """Extra: AutoencoderKL.decode of `frames` emitted latent frames [frames,4,40,64] -> [frames,3,320,512]
    (funcs.py:360 decodes one frame per FIFO iteration, 148 per video)."""
    from moca_video_amd import AutoencoderKL
    from moca_video_amd.weightgen import init_random_
    with torch.device(device):
        ae = AutoencoderKL(ddconfig=VAE_DD, lossconfig={"target": "torch.nn.Identity"}, embed_dim=4)
    ae = ae.to(device)
    init_random_(ae, 123)
    g = torch.Generator(device=device).manual_seed(11)
    z = torch.randn(frames, 4, H, W, device=device, generator=g) / 0.18215
    for _ in range(2):
        out = ae.decode(z)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        out = ae.decode(z)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    assert out.shape == (frames, 3, 8 * H, 8 * W) and bool(torch.isfinite(out).all())
    return {"ms_per_frame": round(dt / frames * 1e3, 3), "frames_per_launch": frames,
            "tflops": round(VAE_FLOP_PER_FRAME * frames / dt / 1e12, 1),
            "s_per_video_148_frames": round(148 * dt / frames, 3)}, ae


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(n, argv, timeout_s=3000):
    """`python bench.py --gpus N` without a launcher: start N fresh child processes of this file, one per GPU
    (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment, the reference idiom of hand-launched ranks,
    videocrafter_main.py:179-181).  The parent never touches the GPU (no HIP call, no torch.cuda call): it polls ALL children,
    forwards rank 0's stdout (the JSON line) and returns the worst exit code; the first child that fails (or the overall
    timeout) ends the others -- exact PIDs of children it started, never a pattern -- so a rank that dies during start-up does
    not leave the rest waiting in the rendezvous holding their GPUs."""
    import subprocess
    import tempfile
    port = _free_port()
    procs = []
    out0 = tempfile.TemporaryFile(mode="w+")
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        out = out0 if r == 0 else sys.stderr                    # only rank 0 owns stdout: ONE JSON line
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env, stdout=out))
    t_end = time.monotonic() + timeout_s
    rc = 0
    live = list(procs)
    while live:
        for p in list(live):
            code = p.poll()
            if code is not None:
                live.remove(p)
                rc = rc or code
        if rc != 0 or time.monotonic() > t_end:
            rc = rc or 124
            for p in live:                                       # exact PIDs of our own children
                p.terminate()
            for p in live:
                try:
                    p.wait(timeout=20)
                except subprocess.TimeoutExpired:
                    p.kill()
                    p.wait()
            break
        time.sleep(0.2)
    out0.seek(0)
    for line in out0.read().splitlines():
        # stdout carries the JSON line only; anything else rank 0 printed (gloo/RCCL banners) goes to stderr
        (sys.stdout if line.lstrip().startswith("{") else sys.stderr).write(line + "\n")
    sys.stdout.flush()
    return rc


def launcher_selftest(args):
    """`--selftest-cpu`: the multi-rank plumbing of this file WITHOUT the hot path (CPU, gloo): rendezvous from the
    environment, C1 exactly as the GPU job runs it (only rank 0 holds real parameters, the others NaN; timed broadcast;
    all-gathered checksums must agree -- `broadcast_and_verify`), prompt-row striding of config[4], barrier, K no-op "steps",
    max-over-ranks timing, C2 gather, rank 0 prints one JSON line.  Test infrastructure for tests/test_dist_cpu.py -- it
    measures nothing and says so in `metric`; a stubbed-out broadcast makes every rank exit non-zero."""
    from moca_video_amd import dist as mdist
    rank, local, world = mdist.init_from_env(backend="gloo")
    import functools
    import types
    from moca_video_amd import ops
    torch.manual_seed(100)
    m = torch.nn.Sequential(torch.nn.Linear(64, 64), torch.nn.Linear(64, 8))
    if rank != 0:
        with torch.no_grad():
            for p in m.parameters():
                p.fill_(float("nan"))
    # what UNetModel offers dist.broadcast_packed: `_packed` (id -> packed operands) and plans whose recorded steps reference them
    m._packed = {id(l): ops.pack_linear(l.weight.detach(), l.bias.detach(), device="cpu") for l in m}
    plan = types.SimpleNamespace(steps=[functools.partial(lambda *a, **k: None, None, pw, None) for pw in m._packed.values()])
    info = broadcast_and_verify(m, torch.device("cpu"), world, rank, [plan])
    rows = mdist.shard_indices(N_PROMPTS, rank, world)
    mdist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        pass
    mdist.barrier()
    dt = mdist.max_over_ranks(time.perf_counter() - t0, torch.device("cpu"))
    outs = mdist.gather_results(torch.tensor([float(rank)] + [float(r) for r in rows]), dst=0)
    if rank == 0:
        ranks = [int(o[0]) for o in outs]
        got_rows = sorted(int(v) for o in outs for v in o[1:])
        print(json.dumps({"metric": "launcher-selftest (no compute)", "value": 0.0, "unit": "none", "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "ranks_gathered": ranks,
                          "weights_equal_after_broadcast": info["param_checksums_equal"], "multi_gpu": info,
                          "rows_covered": got_rows == list(range(N_PROMPTS)), "elapsed_s": dt}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fifo", action="store_true", help="skip the extra batched-FIFO-iteration measurement")
    ap.add_argument("--cfg-mode", default="batched", choices=["batched", "concurrent"],
                    help="cond+uncond as one B=2 launch, or as two concurrent B=1 hipGraphs on two streams")
    ap.add_argument("--no-shared-prefix", action="store_true",
                    help="A/B: evaluate the two CFG branches as a plain B=2 batch instead of sharing the layers before the first cross-attention")
    ap.add_argument("--step-mode", default="graph", choices=["graph", "host"],
                    help="graph: one hipGraph per DDIM step (timestep, UNet, noise, guidance + update: what DDIMSampler.sample runs); "
                         "host: p_sample_ddim per step (UNet graph + four small launches issued by the host)")
    ap.add_argument("--no-weight-prefetch", action="store_true",
                    help="A/B: no moca_gemm_params.prefetch (spare blocks of a GEMM launch's grid that read the next weight-heavy "
                         "launch's weights into the Infinity Cache)")
    ap.add_argument("--cpu-threads", type=int, default=0)
    ap.add_argument("--frames", type=int, default=16)
    ap.add_argument("--height", type=int, default=40)
    ap.add_argument("--width", type=int, default=64)
    ap.add_argument("--no-video", action="store_true", help="skip the measured whole-video leg (base sampling + 148 FIFO iterations + decode)")
    ap.add_argument("--emulate-world", type=int, default=0,
                    help="1-GPU rehearsal of rank 0's share of the N-GPU configs[4] job (its 64/N prompt rows in B=16 forwards, no "
                         "collectives): `value` is then THIS GPU's UNet-steps/s on that workload; extra legs are skipped")
    ap.add_argument("--no-emulate-world", action="store_true",
                    help="skip the default leg that measures rank 0's share of the 8-GPU configs[4] job on this GPU (`emulate_world_8`)")
    ap.add_argument("--selftest-cpu", action="store_true", help="launcher/collective plumbing only (CPU, gloo): see launcher_selftest")
    args = ap.parse_args()

    # `python bench.py --gpus N` with no launcher around it: become the launcher (before anything touches the GPU)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    if args.selftest_cpu:
        return launcher_selftest(args)

    from moca_video_amd import dist as mdist
    from moca_video_amd import lib as mlib
    from moca_video_amd.sampler import DDIMSampler
    rank, local, world = mdist.init_from_env()
    if world != args.gpus:
        if rank == 0:
            print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)
    device = torch.device("cuda", local % torch.cuda.device_count())      # (more ranks than devices only in the 1-GPU gloo rehearsal)
    torch.cuda.set_device(device)
    lib = mlib.load()

    dm = build_model(device, seed=321, materialise=(rank == 0))           # ranks != 0: NaN until C1 delivers rank 0's packed operands
    multi = None
    unet = dm.model.diffusion_model
    unet.weight_prefetch = not args.no_weight_prefetch
    sampler = DDIMSampler(dm)
    sampler.cfg_mode = args.cfg_mode
    sampler.share_prefix = not args.no_shared_prefix
    S = 50
    sampler.make_schedule(S, ddim_eta=1.0, verbose=False)

    T, H, W = args.frames, args.height, args.width
    # N = 1: BASELINE.json configs[1], one prompt (B = 2 with the CFG branch).  N > 1: configs[4], the 64 prompt rows strided
    # over the ranks (videocrafter_main.py:181), each rank running its rows in batches of 8 prompts (B = 16 per forward);
    # prompt row r has seed 321 + r: latents and context are drawn per row, whatever rank owns it
    emu = args.emulate_world if world == 1 and args.emulate_world > 1 else 0
    rows = mdist.shard_indices(N_PROMPTS, 0, emu) if emu else ([0] if world == 1 else mdist.shard_indices(N_PROMPTS, rank, world))
    batches = []
    for i in range(0, len(rows), PROMPTS_PER_FORWARD):
        xs, cs, us = [], [], []
        for r in rows[i:i + PROMPTS_PER_FORWARD]:
            g = torch.Generator(device=device).manual_seed(321 + r)
            xs.append(torch.randn(1, 4, T, H, W, device=device, generator=g))
            cs.append(torch.randn(1, 77, 1024, device=device, generator=g))
            us.append(torch.randn(1, 77, 1024, device=device, generator=g))
        nb = len(xs)
        fps = torch.tensor([10] * nb, device=device)
        batches.append(dict(x=torch.cat(xs), cond={"c_crossattn": [torch.cat(cs)], "fps": fps},
                            uc={"c_crossattn": [torch.cat(us)], "fps": fps}, n=nb))
    x, ctx, cond = batches[0]["x"][:1], batches[0]["cond"]["c_crossattn"][0][:1], None
    cond = {"c_crossattn": [ctx], "fps": torch.tensor([10], device=device)}

    # One bench step = one DDIM step of `DDIMSampler.sample`'s loop (ddim.py:226-252) for every prompt batch of this rank.  With
    # --step-mode graph (default) the loop body is ONE hipGraph per batch (fifo_graph.BaseEngine, what `sample()` runs): timestep
    # rows, the shared-prefix UNet forward, device noise, guidance + DDIM update in place; --step-mode host calls p_sample_ddim.
    engines = None
    if args.step_mode == "graph" and args.cfg_mode == "batched" and sampler.share_prefix:
        from moca_video_amd.fifo_graph import BaseEngine
        engines = [BaseEngine(dm, sampler, b["x"], b["cond"], b["uc"], 12.0, seed=321 + i) for i, b in enumerate(batches)]
    if world > 1:
        # C1 (RCCL over xGMI) + checksum proof, on the PACKED set: every rank has built its plans (ranks != 0 from NaN placeholders)
        if engines is None:
            raise SystemExit("multi-GPU runs use --step-mode graph (the plans must exist before the packed broadcast)")
        multi = broadcast_and_verify(unet, device, world, rank, [e.plan for e in engines])

    def ddim_step(i, imgs):
        if engines is not None:
            cur = torch.cuda.current_stream(device)
            for e in engines:
                e.step()                                  # (enqueued on the engine's own stream, behind `cur`)
                cur.wait_stream(e.plan.stream)            # one forward at a time: the per-launch HIP events time a launch, not a queue
            return imgs
        index = S - 1 - (i % S)
        out = []
        for b, img in zip(batches, imgs):
            ts = torch.full((b["n"],), int(sampler.ddim_timesteps[index]), device=device, dtype=torch.long)
            img, _ = sampler.p_sample_ddim(img, b["cond"], ts, index=index, unconditional_guidance_scale=12.0,
                                           unconditional_conditioning=b["uc"])
            out.append(img)
        return out

    img = [b["x"] for b in batches]
    for i in range(max(args.warmup, 2)):       # >= 2: eager pass + hipGraph capture pass
        img = ddim_step(i, img)
    torch.cuda.synchronize()
    plans = [e.plan for e in engines] if engines is not None else list(unet._plans.values())
    graph_on = all(pl.graph is not None for pl in plans)

    # HIP events on the stream(s) the UNet graphs are launched on (torch.cuda.Event would only see
    # torch's current stream): bracket every UNet graph launch inside the timed region.
    ev = []

    def hook(pl):
        handle = C.c_void_p(pl.stream.cuda_stream)
        orig = pl._launch

        def timed_launch(h):
            a, b = C.c_void_p(), C.c_void_p()
            lib.moca_event_create(C.byref(a)); lib.moca_event_create(C.byref(b))
            lib.moca_event_record(a, handle)
            orig(h)
            lib.moca_event_record(b, handle)
            ev.append((a, b))
        pl._launch = timed_launch
        return orig
    originals = [hook(pl) for pl in plans]

    mdist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    img = [b["x"] for b in batches]
    for i in range(args.steps):
        img = ddim_step(i, img)
    torch.cuda.synchronize()
    if engines is not None:
        img = [e.latents() for e in engines]
    mdist.barrier()
    dt = time.perf_counter() - t0
    dt = mdist.max_over_ranks(dt, device)
    for pl, orig in zip(plans, originals):
        pl._launch = orig

    unet_ms = []
    for a, b in ev:
        ms = C.c_float()
        lib.moca_event_elapsed_ms(a, b, C.byref(ms))
        unet_ms.append(ms.value)
        lib.moca_event_destroy(a); lib.moca_event_destroy(b)
    img = torch.cat(img)                                                     # [prompts of this rank, 4, T, H, W]
    finite = bool(torch.isfinite(img).all().item())
    if world > 1 and len(rows) * world != N_PROMPTS:                         # uneven shards: pad to the longest for the gather
        pad = -(-N_PROMPTS // world) - len(rows)
        img = torch.cat([img, img[-1:].expand(pad, -1, -1, -1, -1)]) if pad > 0 else img
    outs = mdist.gather_results(img, dst=0)                                  # C2

    if rank != 0:
        return
    n_prompts = (len(rows) if emu else 1) if world == 1 else N_PROMPTS
    if multi is not None:
        multi.update({"prompts_total": N_PROMPTS, "prompts_per_rank": len(rows), "prompts_per_forward": batches[0]["n"],
                      "forwards_per_step_per_rank": len(batches), "gathered_result_tensors": len(outs),
                      "gathered_results_finite": all(bool(torch.isfinite(o).all()) for o in outs)})
    unet_steps = 2 * args.steps * n_prompts
    value = unet_steps / dt
    concurrent = args.cfg_mode == "concurrent"
    avg_launch_ms = sum(unet_ms) / max(len(unet_ms), 1)
    flop_unit = FLOP_PER_UNET_STEP * (T * H * W) / (16 * 40 * 64)
    if concurrent:
        # two B=1 graphs run overlapped on two streams: a launch's own span includes time it shares with the
        # other chain, so the rate is taken over both launches of a step: 2 units / mean(span of the pair)
        flop_per_launch = flop_unit
        achieved = 2 * flop_unit / (avg_launch_ms * 1e-3) / 1e12 if avg_launch_ms > 0 else 0.0
    else:
        flop_per_launch = 2 * batches[0]["n"] * flop_unit      # one launch = batched (cond + uncond) x prompts forward
        achieved = flop_per_launch / (avg_launch_ms * 1e-3) / 1e12 if avg_launch_ms > 0 else 0.0
    name, cus = mlib.device_info()
    traffic_bytes, traffic_src = traffic_from_profile()
    n_launches = max([len(pl.steps) for pl in plans] or [0])
    executed = plan_flops(plans[0]) if plans else 0.0
    executed_rate = executed / (avg_launch_ms * 1e-3) / 1e12 if avg_launch_ms > 0 and not concurrent else 0.0
    res = {
        "metric": "denoising UNet-steps/sec @16x320x512 fp16",
        "value": round(value, 3),
        "unit": "UNet-steps/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": max(args.warmup, 2),
        "ms_per_step": round(dt / args.steps * 1e3, 3),
        "higher_is_better": True,
        "scaling": "weak" if world == 1 else "strong",
        "vs_baseline": None,
        "dtype": "f16",
        "data": "synthetic",
        "config": {"workload": ("VideoCrafter2 3D-UNet 16x320x512 (latents [1,4,%d,%d,%d]), DDIM S=50 eta=1 CFG=12, single prompt; "
                                "step = 1 DDIM step = 2 UNet-steps (batched cond+uncond) + CFG + DDIM update" % (T, H, W)) if world == 1 else
                               ("BASELINE configs[4]: %d prompt rows strided over %d GPUs (videocrafter_main.py:181), VideoCrafter2 3D-UNet "
                                "16x320x512, DDIM S=50 eta=1 CFG=12; step = 1 DDIM step of ALL %d prompts = %d UNet-steps, each rank "
                                "running its %d rows as %d forward(s) of B=%d" % (N_PROMPTS, world, N_PROMPTS, 2 * N_PROMPTS, len(rows),
                                                                                  len(batches), 2 * batches[0]["n"])),
                   "unet_steps_per_step": 2 * n_prompts, "context_tokens": 77, "weights": "random-init, 1.41B params, fp16 packed" +
                   ("" if world == 1 else "; materialised and packed on rank 0 only, packed set RCCL-broadcast (C1), checksums all-gathered, no fp32 masters on the other ranks"),
                   "parallelism": f"dp{world} (independent prompts, no collective in the loop)",
                   "hipgraph_replay": graph_on, "cfg_mode": args.cfg_mode, "cfg_shared_prefix": sampler.share_prefix and args.cfg_mode == "batched", "weight_prefetch": unet.weight_prefetch, "step_mode": "graph" if engines is not None else "host", "device": name, "compute_units": cus, "output_finite": finite,
                   "hbm_peak_gib": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2)},
        "achieved_tflops": round(value / world * FLOP_PER_UNET_STEP / 1e12, 2),
        "roofline": {"bound": "mfma", "achieved": round(achieved, 2), "peak": PEAK_F16_MFMA_TFLOPS, "unit": "TFLOP/s",
                     "frac": round(achieved / PEAK_F16_MFMA_TFLOPS, 4), "traffic": traffic_bytes if not concurrent else None, "traffic_source": traffic_src if not concurrent else None,
                     "kernel": ("the whole DDIM-step hipGraph (%d launches: timestep rows + device noise + the UNet forward + guidance / "
                                "DDIM update; the implicit-GEMM conv/linear kernels gemm_w80s/gemm_glds/gemm_sqp are 84%% of its kernel "
                                "time, profiles/r05_bench_kernel_stats_summary.txt)" if engines is not None else
                                "UNet forward launch sequence (hipGraph of %d launches)") % n_launches +
                               (", two B=1 graphs on two streams" if concurrent else ", batch %d" % (2 * batches[0]["n"])),
                     "flop_per_launch": flop_per_launch, "avg_launch_ms": round(avg_launch_ms, 3), "launches": len(unet_ms),
                     # `achieved` / `frac` price the reference's ALGORITHMIC FLOPs.  The launch sequence executes fewer: the
                     # two guidance branches share everything before the first cross-attention and `Upsample` runs as four 2x2 convs
                     "executed_flop_per_launch": executed, "executed_tflops": round(executed_rate, 2),
                     "frac_executed": round(executed_rate / PEAK_F16_MFMA_TFLOPS, 4)},
    }
    if multi is not None:
        res["multi_gpu"] = multi
    if emu:
        res["emulated_world"] = emu
        res["config"]["workload"] = ("1-GPU rehearsal of rank 0 of the %d-GPU configs[4] job: %d prompt rows as %d forward(s) of B=%d per DDIM "
                                     "step; no collectives; value = this GPU alone" % (emu, len(rows), len(batches), 2 * batches[0]["n"]))
        print(json.dumps(res))
        return
    note = lambda m: print(f"[bench] {m}", file=sys.stderr, flush=True)     # (progress on stderr: a silent multi-minute run looks hung)
    note("headline measured: %.2f UNet-steps/s" % value)
    if world == 1 and not args.no_emulate_world:
        res["emulate_world_8"] = emulate_world_leg(dm, sampler, device, T, H, W)
        note("emulate_world_8 done")
    if world == 1 and not args.no_fifo:
        # (operand statistics move the clock -- the same iteration on the headline's random-init weights, beside the
        #  zero-data one the long video leg needs)
        res["fifo_random_init"] = fifo_leg(dm, device, T, H, W, iters=5, weights="random-init")
        zdd = ZeroDataDenoiser(dm)
        res["fifo"] = fifo_leg(dm, device, T, H, W)
        note("fifo leg done")
        res["fifo_prompt_mode"] = fifo_leg(dm, device, T, H, W, mode="prompt")
        res["vae_decode"], ae = vae_leg(device, H, W)
        note("fifo prompt-mode + vae legs done")
        res["fifo"]["projected_s_per_video_incl_vae_decode"] = round(
            res["fifo"]["projected_s_per_video_148_iterations"] + res["vae_decode"]["s_per_video_148_frames"], 1)
    if world == 1 and not args.no_fifo and not args.no_video:
        res["video"] = video_leg(dm, ae, device, T, H, W)
        note("video leg done")
    if world == 1 and not args.no_fifo:
        zdd.restore()                           # back to the headline weights
    if world == 1 and not args.no_cpu_baseline:
        # the host cores of ONE GPU's share of the box: min(affinity set, CPUs online / 8 GPUs per node) -- 32 of the 256 on the 8-GPU
        # hosts of this pool.  (All 256 threads on the fp32 oracle oversubscribe the convolutions: the leg then runs for > 7 minutes.)
        threads = args.cpu_threads or max(1, min(len(os.sched_getaffinity(0)), (os.cpu_count() or 8) // 8))
        ts = torch.full((1,), int(sampler.ddim_timesteps[S - 1]), device=device, dtype=torch.long)
        note("cpu_baseline: 4 runs of the fp32 oracle on %d threads (~20 s each)" % threads)
        cdt, y_cpu, ctimes = cpu_baseline(dm, x, ctx, ts, threads)
        y_gpu = dm.apply_model(x, ts, cond).float().cpu()
        err = ((y_gpu - y_cpu).abs().max() / y_cpu.abs().max()).item()
        res["cpu_baseline"] = {"value": round(1.0 / cdt, 5), "unit": "UNet-steps/s", "cores": threads, "kind": "port",
                               "cpu_model": cpu_model_name(), "cpus_online": os.cpu_count(),
                               "sample": "1 UNet-step (fp32 oracle of the reference UNet, [1,4,%d,%d,%d], same weights/inputs) per run; "
                                         "1 warm-up + %d timed runs (%s s), median; HIP-vs-oracle max rel err %.2e; threads = one GPU's share of "
                                         "the host (min(affinity set %d, %d CPUs online / 8 GPUs))" % (T, H, W, len(ctimes), "/".join("%.1f" % c for c in ctimes), err,
                                                                                                         len(os.sched_getaffinity(0)), os.cpu_count() or 0)}
    print(json.dumps(res))


if __name__ == "__main__":
    try:
        main()
    finally:
        import torch.distributed as _dist          # (not moca_video_amd.dist: the launcher parent must not load the HIP library)
        if _dist.is_available() and _dist.is_initialized():
            _dist.destroy_process_group()
