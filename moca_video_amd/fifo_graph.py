"""One outer iteration of `fifo_ddim_sampling` (scripts/evaluation/funcs.py:305-371) as ONE hipGraph.

The reference walks the 2n windows of an iteration one after the other (clone the window, two UNet forwards, ddim_step with
~100 tiny torch ops per frame, write the second half back), decodes, then shifts the queue by cloning it.  Here everything an
iteration needs lives on the device and depends on nothing the host has to look at:

  * the latent queue is a ring [C][Q][HW] f32 whose head / iteration counter / RNG seed sit in a 32-byte state block
    (`moca_fifo_state`); a shift is `head += 1`, the FreeInit-mixed frame overwrites the dequeued frame's slot;
  * timesteps, DDIM coefficients, mask-frame indices and enhancement factors depend on the window slot only (funcs.py:290-312):
    built once as device tables;
  * the noise of ddim.py:561 / funcs.py:92 comes from a Philox kernel keyed by (seed, iteration) -- or from the caller, for
    fixtures;
  * the 2n conditional windows (two prompts: 154 tokens) and their 2n unconditional copies (77 tokens) are ONE UNet forward of
    batch 4n with two context segments (`_Plan.segs`);
  * classifier-free guidance, the MoCA ddim_step of all windows, the write-back, the emission, the FreeInit mix and the queue /
    mask shift are five more launches on the same stream.

The recorded sequence (gather + ~640 UNet launches + tail) is run eagerly once, captured on the second iteration and replayed
as one `hipGraphLaunch` from then on; the host never synchronises inside the loop."""
from __future__ import annotations

import ctypes as C
import functools

import numpy as np
import torch

from . import lib as _l
from . import ops
from .freeinit import get_freq_filter
from .plan import _Plan
from .unet import UNetModel, same_fps


def fifo_windows(args):
    """Window schedule of one outer iteration (funcs.py:290-312): (start, mid, end) for rank = 2n-1 .. 0 (reversed so every
    window reads only not-yet-rewritten frames)."""
    f = args.video_length
    n = 2 * args.num_partitions if args.lookahead_denoising else args.num_partitions
    for rank in reversed(range(n)):
        start = rank * (f // 2) if args.lookahead_denoising else rank * f
        yield start, start + f // 2, start + f


def _ptr(t):
    return None if t is None else t.data_ptr()


class FifoEngine:
    """Device-resident MoCA-FIFO loop, prompt mode and DAVIS-video mode.  `supported(...)` says whether a call can run here;
    `fifo.fifo_ddim_sampling` falls back to its host-driven loop otherwise (a mask-producer callback, per-call mask lists, a model
    that is not ours).

    DAVIS mode (`anchor_moments`): the FreeInit anchor of every shift is a fresh posterior sample of the VAE encoding of the LAST
    DAVIS frame (funcs.py:101-108) -- the frame never changes, so its moments [1, 2z, h, w] are encoded ONCE by the caller and an
    iteration only draws `scale_factor * (mean + std * noise)` (`moca_gaussian_sample_f32`) in front of the mix."""

    @staticmethod
    def supported(model, cond, latents, davis_data=None, sam_masks_fn=None):
        unet = getattr(getattr(model, "model", None), "diffusion_model", None)
        return (isinstance(unet, UNetModel) and isinstance(cond, dict) and "c_crossattn" in cond
                and (davis_data is None or getattr(model, "first_stage_model", None) is not None)
                and sam_masks_fn is None and latents is not None and latents.is_cuda and latents.shape[0] == 1)

    def __init__(self, args, model, sampler, cond, uc, cfg_scale, latents, conditioned_image=None, masks=None, gamma=0.5,
                 n_slots=1, seed=0, anchor_moments=None, scale_factor=1.0, sam_capacity=0):
        """sam_capacity > 0 (and no `masks`): prompt mode with precomputed Grounded-SAM-2 candidates -- `ddim_step`'s segmentation
        branch (ddim.py:592-606 -> `_apply_segmentation` :739-903) runs inside the iteration graph on at most `sam_capacity`
        candidate masks per iteration, handed to `step(sam_masks=...)`."""
        self.unet = unet = model.model.diffusion_model
        dev = latents.device
        self.device = dev
        if unet._packed is None:
            unet._pack()
        f = args.video_length
        wins = list(fifo_windows(args))
        self.wins, self.nW, self.f = wins, len(wins), f
        nW = self.nW
        _, Cc, Q, H, W = latents.shape
        HW = H * W
        self.C, self.Q, self.H, self.W, self.HW = Cc, Q, H, W, HW
        self.lookahead = bool(args.lookahead_denoising)
        self.emit_frame = f // 2 if self.lookahead else 0
        guided = uc is not None and cfg_scale != 1.0
        self.reps = 2 if guided else 1
        # ---- timesteps / coefficient tables per window slot (funcs.py:290-294,311-312; ddim.py:405-430,565-582)
        timesteps = np.asarray(sampler.ddim_timesteps)
        indices = np.arange(args.num_inference_steps)
        if self.lookahead:
            timesteps = np.concatenate([np.full((f // 2,), timesteps[0]), timesteps])
            indices = np.concatenate([np.full((f // 2,), 0), indices])
        coef = np.zeros((nW, f, 6), np.float32)
        enh = np.ones((nW, f), np.float32)
        mframe = np.full((nW, f), -1, np.int32)
        t_rows = np.zeros((nW, f), np.int64)
        for w, (s0, _, e0) in enumerate(wins):
            t_rows[w] = timesteps[s0:e0]
            Fm = 0 if masks is None else max(0, min(e0, masks.shape[2]) - s0)
            coef[w], enh[w], mi = sampler.step_tables(indices[s0:e0], timesteps[s0:e0], H, Fm)
            mframe[w] = np.where(mi >= 0, s0 + mi, -1)
        to_dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        self.coef, self.enh, self.mframe = to_dev(coef), to_dev(enh), to_dev(mframe)
        self.win_start = to_dev(np.asarray([s0 for s0, _, _ in wins], np.int32))
        # ---- the batched UNet plan: [conditional windows | unconditional windows], one context segment each
        cc = torch.cat(cond["c_crossattn"], 1)
        ctxs = [cc.expand(nW, -1, -1)]
        if guided:
            ctxs.append(torch.cat(uc["c_crossattn"], 1).expand(nW, -1, -1))
        segs = tuple((nW, int(c.shape[1])) for c in ctxs)
        B = self.reps * nW
        # guided: the unconditional windows are the SAME latents with another context: one plan whose prefix (everything before the
        # first cross-attention) runs once for both branches
        # (the prefix adds ONE fps embedding: branches with different fps run as a plain batch of 2 nW videos instead)
        fps_c = cond.get("fps", 16)
        shared = guided and same_fps([fps_c, uc.get("fps", fps_c)])
        self.gather_reps = 1 if shared or not guided else 2
        self.plan = plan = _Plan(unet, B, f, H, W, segs, torch.float32, dev, shared_x=shared)

        def fps_rows(fp):
            if isinstance(fp, int):
                return torch.full((nW * f,), fp, dtype=torch.int64, device=dev)
            return torch.as_tensor(fp, device=dev).reshape(-1)[:1].to(torch.int64).expand(nW * f)
        fr = [fps_rows(fps_c)] + ([fps_rows(uc.get("fps", fps_c))] if guided else [])
        with torch.cuda.stream(plan.stream):
            plan.t_rows.copy_(to_dev(t_rows).reshape(-1).repeat(self.reps))
            plan.fps_rows.copy_(torch.cat(fr))
            r = 0
            for c in ctxs:
                n = c.shape[0] * c.shape[1]
                plan.ctx[r:r + n].copy_(c.reshape(n, -1))
                r += n
        # ---- device state
        f32 = dict(dtype=torch.float32, device=dev)
        self.queue = latents[0].to(torch.float32).reshape(Cc, Q, HW).clone()
        st = _l.FifoState(0, 0, seed & 0xffffffff, (seed >> 32) & 0xffffffff, 0)
        self.state = torch.frombuffer(bytearray(bytes(st)), dtype=torch.int32).to(dev)
        n_win = nW * Cc * f * HW
        self.noise = torch.zeros(n_win + 2 * Cc * HW, **f32)         # [nW][C][f][HW] window noise | [C][HW] enqueued noise | [C][HW] anchor draw
        self.moments = None
        if anchor_moments is not None:
            self.moments = anchor_moments.to(dev, torch.float32).reshape(2 * Cc, HW).contiguous()
        self.momentum = torch.zeros(nW, Cc, f, HW, **f32)            # ddim.py:395-397
        self.pred_x0 = torch.empty(nW, Cc, f, HW, **f32)
        self.x_prev = torch.empty(nW, Cc, f, HW, **f32)
        self.anchor = torch.empty(Cc, HW, **f32)
        self.newframe = torch.empty(Cc, HW, **f32)
        self.n_slots = max(1, int(n_slots))
        self.emitted = torch.zeros(self.n_slots, Cc, HW, **f32)
        self.lpf = get_freq_filter((1, Cc, 1, H, W), dev, "gaussian", 1, 0.25, 0.25).reshape(-1, 1, H, W)[0].contiguous()   # funcs.py:95
        lib = _l.load()
        self.mix_ws = torch.empty(int(lib.moca_freq_mix_ws_bytes(Cc, 1, H, W)) // 4, **f32)
        self.mask = self.mask_sums = self.cond = None
        self.sam_capacity = int(sam_capacity) if masks is None else 0
        self._t_host = t_rows                                     # [nW][f] timesteps (host copy: which frames take the <= 300 branch)
        if self.sam_capacity > 0:
            self.sam_cand = torch.zeros(self.sam_capacity, HW, **f32)
            self.sam_tab = torch.zeros(2, nW * f, dtype=torch.int32, device=dev)          # [0] pool offset, [1] candidate count
            self.sam_eff = torch.zeros(nW, f, HW, **f32)
            self.sam_idx = torch.full((nW * f,), -1, dtype=torch.int32, device=dev)
            self._sam_stage = [(torch.empty(self.sam_capacity, HW, dtype=torch.float32).pin_memory(),
                                torch.zeros(2, nW * f, dtype=torch.int32).pin_memory(), torch.cuda.Event()) for _ in range(2)]
            self._sam_turn = 0
            self.cond = self._cond_image(conditioned_image, Cc, HW, dev)
        if masks is not None:
            if masks.shape[2] != Q:
                raise ValueError("the device-resident loop wants one mask frame per queue frame")
            self.mask = masks[0, 0].to(torch.float32).reshape(Q, HW).clone()
            self.mask_sums = torch.empty(Q, **f32)
            _l.check(lib.moca_mask_frame_sums_f32(_l.ptr(self.mask), _l.ptr(self.mask_sums), Q, HW,
                                                  C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), "moca_mask_frame_sums_f32")
            self.cond = self._cond_image(conditioned_image, Cc, HW, dev)
        torch.cuda.current_stream(dev).synchronize()
        # ---- the launch sequence of one iteration = [noise, gather] + UNet + [guidance + step + write-back, FreeInit mix, advance]
        p = _l.FifoStepParams()
        p.state, p.x = self.state.data_ptr(), plan.x_in.data_ptr()
        eps = plan.out.reshape(B, -1)
        p.eps_c = eps[:nW].data_ptr()
        p.eps_u = eps[nW:].data_ptr() if guided else None
        p.noise, p.momentum, p.queue = self.noise.data_ptr(), self.momentum.data_ptr(), self.queue.data_ptr()
        p.x_prev, p.pred_x0 = self.x_prev.data_ptr(), self.pred_x0.data_ptr()
        p.coef, p.win_start = self.coef.data_ptr(), self.win_start.data_ptr()
        p.mask, p.mask_sums = _ptr(self.mask), _ptr(self.mask_sums)
        p.mask_frame, p.enh, p.cond = self.mframe.data_ptr(), self.enh.data_ptr(), _ptr(self.cond)
        if self.sam_capacity > 0:
            p.sam_eff, p.sam_idx = self.sam_eff.data_ptr(), self.sam_idx.data_ptr()
        p.cfg_scale = float(cfg_scale)
        p.beta, p.one_minus_beta = float(np.float32(sampler.beta)), float(np.float32(1 - sampler.beta))
        p.gamma, p.one_minus_gamma = float(np.float32(gamma)), float(np.float32(1 - gamma))
        p.nW, p.C, p.Q, p.f, p.HW = nW, Cc, Q, f, HW
        p.wb_from = f // 2 if self.lookahead else 0
        self._params = p
        st_, q_ = _l.ptr(self.state), _l.ptr(self.queue)
        S = lambda: C.c_void_p(ops.current_stream())

        def pre():
            _l.check(lib.moca_fifo_randn_f32(st_, _l.ptr(self.noise), self.noise.numel(), S()), "moca_fifo_randn_f32")
            # the FreeInit anchor is frame 0 AFTER the write-backs (funcs.py:88): only with lookahead is it untouched by them
            early_anchor = self.moments is None and self.lookahead
            _l.check(lib.moca_fifo_gather_windows_f32(st_, q_, _l.ptr(plan.x_in), _l.ptr(self.anchor) if early_anchor else None,
                                                      _l.ptr(self.win_start), nW, self.gather_reps, Cc, Q, f, HW, S()), "moca_fifo_gather_windows_f32")

        def post():
            if self.sam_capacity > 0:      # which candidate masks each window frame injects (pre_masks / IoU / > 80 % bookkeeping)
                _l.check(lib.moca_sam_select_masks_f32(_l.ptr(self.sam_cand), _l.ptr(self.sam_tab[0]), _l.ptr(self.sam_tab[1]),
                                                       _l.ptr(plan.t_rows), _l.ptr(self.sam_eff), _l.ptr(self.sam_idx), nW, f, HW, S()),
                         "moca_sam_select_masks_f32")
            _l.check(lib.moca_fifo_step_windows_f32(C.byref(p), S()), "moca_fifo_step_windows_f32")
            if self.moments is None and not self.lookahead:  # rank 0 rewrote frame 0 (funcs.py:353-354): read the anchor now
                _l.check(lib.moca_fifo_gather_windows_f32(st_, q_, None, _l.ptr(self.anchor), None, 0, 1, Cc, Q, f, HW, S()),
                         "moca_fifo_gather_windows_f32")
            if self.moments is not None:                     # anchor = get_first_stage_encoding(posterior of the last DAVIS frame) (:108)
                _l.check(lib.moca_gaussian_sample_f32(_l.ptr(self.moments), _l.ptr(self.noise[n_win + Cc * HW:]), _l.ptr(self.anchor), 1, Cc, HW,
                                                      float(scale_factor), S()), "moca_gaussian_sample_f32")
            _l.check(lib.moca_freq_mix_3d_f32(_l.ptr(self.anchor), _l.ptr(self.noise[n_win:n_win + Cc * HW]), _l.ptr(self.lpf), _l.ptr(self.newframe),
                                              Cc, 1, H, W, _l.ptr(self.mix_ws), S()), "moca_freq_mix_3d_f32")
            _l.check(lib.moca_fifo_advance_f32(st_, q_, _l.ptr(self.newframe), _l.ptr(self.emitted), self.n_slots, self.emit_frame,
                                               _l.ptr(self.mask), _l.ptr(self.mask_sums), Cc, Q, HW, S()), "moca_fifo_advance_f32")
        self.n_unet_launches = len(plan.steps)
        plan.steps = [pre] + plan.steps + [post]
        self.n_iter = 0
        sampler.momentum = self.momentum[nW - 1].view(1, Cc, f, H, W)      # what the reference's last call (rank 0) leaves behind

    @staticmethod
    def _cond_image(conditioned_image, Cc, HW, dev):
        if conditioned_image is None:
            return torch.zeros(Cc, HW, dtype=torch.float32, device=dev)                      # ddim.py:573-574
        ci = conditioned_image.to(dev)
        if ci.shape[1] != Cc:
            if ci.shape[1] != 3:
                raise ValueError(f"Conditional image must have 3 or 4 channels, got {ci.shape[1]}")
            ci = torch.cat([ci, torch.ones_like(ci[:, :1])], dim=1)                          # :575-578
        return ci.to(torch.float32).reshape(-1, Cc, HW)[0].contiguous()

    def _upload_sam(self, sam_masks):
        """pack the candidates of this iteration -- `sam_masks[w][i]` = [n,H,W] masks Grounded-SAM-2 returns for frame i of window w
        (reference call order), None / empty = no box -- into the pool and enqueue the copy on the plan's stream.  Frames with
        t > 300 never reach the producer (ddim.py:592) and are not uploaded.  Host tensors go through a pinned double buffer (no
        host synchronisation unless the copy of two iterations ago is still in flight); tensors already on the device are copied
        there (stream-ordered, nothing read back: only their shapes are looked at)."""
        pool, tab, ev = self._sam_stage[self._sam_turn]
        self._sam_turn ^= 1
        ev.synchronize()
        tab.zero_()
        used, host_runs, dev_copies = 0, [], []
        if sam_masks is not None:
            for w in range(self.nW):
                cw = sam_masks[w]
                for i in range(self.f):
                    if self._t_host[w, i] > 300 or cw is None or i >= len(cw) or cw[i] is None:
                        continue
                    m = torch.as_tensor(cw[i]).detach().reshape(-1, self.HW)
                    n = m.shape[0]
                    if n == 0:
                        continue
                    if used + n > self.sam_capacity:
                        raise ValueError(f"more than sam_capacity = {self.sam_capacity} candidate masks in one iteration")
                    if m.is_cuda:
                        dev_copies.append((used, n, m))
                    else:
                        pool[used:used + n].copy_(m.to(torch.float32))
                        host_runs.append((used, n))
                    tab[0, w * self.f + i], tab[1, w * self.f + i] = used, n
                    used += n
        for a, n in host_runs:
            self.sam_cand[a:a + n].copy_(pool[a:a + n], non_blocking=True)
        for a, n, m in dev_copies:
            self.sam_cand[a:a + n].copy_(m.to(self.device, torch.float32), non_blocking=True)
        self.sam_tab.copy_(tab, non_blocking=True)
        ev.record(self.plan.stream)

    # ------------------------------------------------------------------------------------------------------------------
    def step(self, noise=None, shift_noise=None, anchor_noise=None, sam_masks=None):
        """one outer iteration (enqueued, not synchronised).  `noise` = list over windows (reference order: rank 2n-1 .. 0) of
        [1,C,f,H,W] tensors, `shift_noise` [1,C,H,W] and (DAVIS mode) `anchor_noise` [1,C,1,H,W] fix the draws (all or none);
        `sam_masks` (engines built with sam_capacity): this iteration's candidate masks, see `_upload_sam`."""
        plan = self.plan
        cur = torch.cuda.current_stream(self.device)
        plan.stream.wait_stream(cur)
        with torch.cuda.stream(plan.stream):
            if self.sam_capacity > 0:
                self._upload_sam(sam_masks)
            elif sam_masks is not None:
                raise ValueError("this engine was built without sam_capacity")
            if noise is not None:
                n_win = self.nW * self.C * self.f * self.HW
                self.noise[:n_win].view(self.nW, -1).copy_(torch.stack([n.reshape(-1) for n in noise]).to(self.device, torch.float32))
                chw = self.C * self.HW
                self.noise[n_win:n_win + chw].copy_(shift_noise.reshape(-1).to(self.device, torch.float32))
                if anchor_noise is not None:
                    self.noise[n_win + chw:].copy_(anchor_noise.reshape(-1).to(self.device, torch.float32))
                self.state[4:5].fill_(1)                                 # ext_noise (the advance clears it)
            handle = plan.stream.cuda_stream
            ops.set_stream(handle)
            try:
                plan._launch(handle)
            finally:
                ops.set_stream(None)
        plan.n_runs += 1
        self.n_iter += 1

    def sync_to(self, stream=None):
        (stream or torch.cuda.current_stream(self.device)).wait_stream(self.plan.stream)

    def latents(self):
        """the queue in frame order [1,C,Q,H,W] (a copy; the ring itself never moves)"""
        self.sync_to()
        head = self.n_iter % self.Q
        return torch.roll(self.queue, -head, dims=1).reshape(1, self.C, self.Q, self.H, self.W)

    def mask_queue(self):
        self.sync_to()
        head = self.n_iter % self.Q
        return None if self.mask is None else torch.roll(self.mask, -head, dims=0).reshape(1, 1, self.Q, self.H, self.W)

    def emitted_frames(self, i0, i1):
        """latent frames emitted by iterations i0 .. i1-1 as [1,C,i1-i0,H,W] (they must still be in the slot ring)"""
        self.sync_to()
        assert self.n_iter - i0 <= self.n_slots and i1 <= self.n_iter
        idx = [i % self.n_slots for i in range(i0, i1)]
        return self.emitted[idx].permute(1, 0, 2).reshape(1, self.C, len(idx), self.H, self.W)

    def window_outputs(self):
        """(x_prev, pred_x0) of the last iteration's windows, [nW][1,C,f,H,W] views in the reference's call order"""
        self.sync_to()
        v = lambda t: [t[w].view(1, self.C, self.f, self.H, self.W) for w in range(self.nW)]
        return v(self.x_prev), v(self.pred_x0)

    def close(self):
        self.plan.close()


class BaseEngine:
    """One step of base sampling -- `DDIMSampler.ddim_sampling`'s loop body (ddim.py:226-252: two UNet calls on the same latents, guidance,
    the DDIM update with `use_scale`, fresh noise) -- as ONE hipGraph: [timestep rows of step i] + the shared-prefix UNet forward of
    the B latents x (conditional, unconditional) + [noise, guidance + update in place, i += 1].  The latents live in the plan's input
    buffer, the schedule in device tables indexed by the state block's iteration counter; the host only replays."""

    @staticmethod
    def supported(model, x, cond, uc, scale):
        unet = getattr(getattr(model, "model", None), "diffusion_model", None)
        ok = isinstance(unet, UNetModel) and x.is_cuda and isinstance(cond, dict) and isinstance(uc, dict) and scale != 1.0
        return ok and set(cond.keys()) == set(uc.keys()) <= {"c_crossattn", "fps"} and \
            getattr(model.model, "conditioning_key", None) == "crossattn" and 2 * x.shape[0] <= 64 and \
            same_fps([cond.get("fps", 16), uc.get("fps", 16)])      # (the shared prefix adds ONE fps embedding)

    def __init__(self, model, sampler, x, cond, uc, cfg_scale, seed=0, keep_pred_x0=False):
        self.unet = unet = model.model.diffusion_model
        dev = x.device
        self.device = dev
        if unet._packed is None:
            unet._pack()
        B, Cc, T, H, W = x.shape
        self.shape = (B, Cc, T, H, W)
        S = len(sampler.ddim_timesteps)
        self.S = S
        cc, cu = torch.cat(cond["c_crossattn"], 1), torch.cat(uc["c_crossattn"], 1)
        segs = ((B, int(cc.shape[1])), (B, int(cu.shape[1])))
        self.plan = plan = _Plan(unet, 2 * B, T, H, W, segs, torch.float32, dev, shared_x=True)

        def fps_rows(fp):
            if isinstance(fp, int):
                return torch.full((B * T,), fp, dtype=torch.int64, device=dev)
            fp = torch.as_tensor(fp, device=dev).reshape(-1).to(torch.int64)
            return (fp if fp.shape[0] == B else fp[:1].expand(B)).repeat_interleave(T)
        f32 = np.float32
        coef = np.zeros((S, 8), np.float32)
        use_scale = bool(sampler.use_scale)
        for i in range(S):                                   # the 0-dim fp32 tensors of ddim.py:331-343, per schedule index
            a_t, a_prev, sig = f32(sampler.ddim_alphas[i]), f32(sampler.ddim_alphas_prev[i]), f32(sampler.ddim_sigmas[i])
            coef[i, :5] = (np.sqrt(a_t), np.sqrt(a_prev), sig, f32(sampler.ddim_sqrt_one_minus_alphas[i]), np.sqrt(f32(1.) - a_prev - sig * sig))
            coef[i, 5] = f32(sampler.ddim_scale_arr[i]) if use_scale else 1.0
            coef[i, 6] = f32(sampler.ddim_scale_arr_prev[i]) if use_scale else 1.0
        self.coef = torch.from_numpy(coef).to(dev)
        self.t_table = torch.from_numpy(np.asarray(sampler.ddim_timesteps, dtype=np.int64)).to(dev)
        self.state = torch.zeros(8, dtype=torch.int32, device=dev)
        n = x.numel()
        self.noise = torch.zeros(n, dtype=torch.float32, device=dev)
        self.pred_x0 = torch.empty(n, dtype=torch.float32, device=dev) if keep_pred_x0 else None
        self._fps_rows = fps_rows
        self.reset(x, cond, uc, seed)
        lib = _l.load()
        eps = plan.out.reshape(2 * B, -1)
        st_ = _l.ptr(self.state)
        S_ = lambda: C.c_void_p(ops.current_stream())

        def pre():
            _l.check(lib.moca_base_set_timestep(st_, _l.ptr(self.t_table), S, _l.ptr(plan.t_rows), plan.t_rows.numel(), S_()), "moca_base_set_timestep")
            _l.check(lib.moca_fifo_randn_f32(st_, _l.ptr(self.noise), n, S_()), "moca_fifo_randn_f32")

        def post():
            _l.check(lib.moca_base_ddim_step_f32(st_, _l.ptr(plan.x_in), _l.ptr(eps[:B]), _l.ptr(eps[B:]), _l.ptr(self.noise), _l.ptr(self.pred_x0),
                                                 _l.ptr(self.coef), S, float(cfg_scale), 1 if use_scale else 0, n, S_()), "moca_base_ddim_step_f32")
        plan.steps = [pre] + plan.steps + [post]

    def reset(self, x, cond, uc, seed=0):
        """start a new trajectory on the same plan: latents x_T, contexts, fps, iteration 0, a new noise stream"""
        plan, dev = self.plan, self.device
        B = self.shape[0]
        if not same_fps([cond.get("fps", 16), uc.get("fps", 16)]):
            raise ValueError("BaseEngine shares the UNet prefix between the two guidance branches: their fps must be equal")
        cc, cu = (torch.cat(c["c_crossattn"], 1).expand(B, -1, -1) for c in (cond, uc))
        st = _l.FifoState(0, 0, seed & 0xffffffff, (seed >> 32) & 0xffffffff, 0)
        cur = torch.cuda.current_stream(dev)
        plan.stream.wait_stream(cur)
        with torch.cuda.stream(plan.stream):
            self.state.copy_(torch.frombuffer(bytearray(bytes(st)), dtype=torch.int32))
            plan.x_in.copy_(x.to(torch.float32))
            plan.fps_rows.copy_(torch.cat([self._fps_rows(cond.get("fps", 16)), self._fps_rows(uc.get("fps", 16))]))
            plan.set_context([cc, cu])
        cur.wait_stream(plan.stream)
        self.n_iter = 0

    def step(self, noise=None):
        """one DDIM step (enqueued, not synchronised); `noise` [B,C,T,H,W] fixes the draw"""
        plan = self.plan
        cur = torch.cuda.current_stream(self.device)
        plan.stream.wait_stream(cur)
        with torch.cuda.stream(plan.stream):
            if noise is not None:
                self.noise.copy_(noise.reshape(-1).to(self.device, torch.float32))
                self.state[4:5].fill_(1)
            handle = plan.stream.cuda_stream
            ops.set_stream(handle)
            try:
                plan._launch(handle)
            finally:
                ops.set_stream(None)
        plan.n_runs += 1
        self.n_iter += 1

    def latents(self):
        torch.cuda.current_stream(self.device).wait_stream(self.plan.stream)
        return self.plan.x_in.clone()

    def last_pred_x0(self):
        torch.cuda.current_stream(self.device).wait_stream(self.plan.stream)
        return None if self.pred_x0 is None else self.pred_x0.view(self.shape).clone()

    def close(self):
        self.plan.close()
