"""Recorded forward of the MI355X UNet: walks the reference-shaped module tree ONCE per
input signature, allocates every intermediate from a pool and records the launch
sequence (C-ABI calls with bound device pointers).  Replays are a single hipGraph launch.

Data layout in HBM: every activation is a row-major [B*T*H*W][C] fp16 matrix
(channels-last, frame outermost).  Spatial ops see contiguous channels per pixel; the
temporal conv / temporal attention reach the other frames of a pixel with a constant
row stride of H*W, so the reference's `(b t) c h w <-> b c t h w <-> (b h w) t c`
rearranges (openaimodel3d.py:43-45,231-233; attention.py:268,275,335-338,352,367) do
not exist here.
"""
from __future__ import annotations

import ctypes as C
import functools
import os
import warnings

import torch

from . import lib as _l
from . import ops
from .unet import (_BasicTransformerBlock, _CatMap, _Downsample, _FMap, _Pool, _ResBlock, _SpatialTransformer,
                   _TemporalTransformer, _Upsample)

N_CU = 256


def gemm_splits(M, pw):
    """split-K factor: only when the launch cannot give every CU a tile (mirrors the tile choice of moca_gemm_f16)."""
    nk = pw.w.shape[1] // 64
    tiles = 0
    if pw.N % 160 == 0 and not pw.geglu and M > 160:        # w80 kernel: 320 x 160 tiles when they fill the chip
        tiles = ((M + 319) // 320) * (pw.N // 160)
        if tiles < 200:
            tiles = 0
    if tiles == 0:
        if M > 128 and (pw.N % 128 == 0 or pw.N % 160 == 0):
            tm, bn = 256, (128 if pw.N % 128 == 0 else 160)
        else:
            tm, bn = 128, (128 if pw.N % 128 == 0 else 64)
        tiles = ((M + tm - 1) // tm) * (pw.N // bn)
    if tiles >= (N_CU * 3) // 4:
        return 1
    cap = N_CU // tiles
    if tiles >= 40:
        # measured at M = N = 1280 (50 tiles, tools/bench_l3.py): K = 1280 runs 17.9 us unsplit against 22-25 us with any split
        # (the slabs and the reduce launch cost more than the idle CUs); K = 2560..5120 is best at 4 splits (one round of 200
        # blocks, 20 % less slab traffic than 5); only the 3x3 convs (K >= 11520) want every CU
        if nk <= 24:
            return 1
        if nk <= 100:
            return max(1, min(cap, 4))
    return max(1, min(cap, nk // 8))


# weight prefetch (moca_gemm_params.prefetch): weights of a launch worth warming the memory-side cache for, and the least work a host
# launch must have so that its spare blocks finish before its tiles do (same-box A/B: profiles/r04_ab_weight_prefetch_in_kernel.txt;
# the environment variables exist for such A/B runs only)
PREFETCH_MIN_BYTES = int(float(os.environ.get("MOCA_PREFETCH_MIN_MB", "4")) * (1 << 20))
PREFETCH_HOST_FLOP = float(os.environ.get("MOCA_PREFETCH_HOST_GF", "50")) * 1e9
# virtual torch.cat in the output blocks (A/B switch: MOCA_VCAT=0 materialises every concat as before round 5)
VIRTUAL_CAT = os.environ.get("MOCA_VCAT", "1") != "0"
# split-K reduce inside the GroupNorm that consumes it (the 5 x 8-latent level of the B = 2 forward; A/B switch MOCA_SKGN=0)
SPLITK_GN = os.environ.get("MOCA_SKGN", "1") != "0"
# GroupNorm of a transformer entry folded into its proj_in as per-statistics-group weights (moca_groupnorm_fold_weights_f16 +
# moca_gemm_params.wgroup_rows): the normalised tensor is never written.  MOCA_GNFOLD=0: A/B
GN_FOLD = os.environ.get("MOCA_GNFOLD", "1") != "0"
SPLITK_GN_ALL = os.environ.get("MOCA_SKGN", "1") == "2"     # (A/B: also the 16-frame GroupNorms of the temporal convs)


class _LNRef:
    """a LayerNorm that exists only as statistics: x (fp16 [M][C]) + the row partial sums its producer left behind
    (f32 [nparts][M][2]); consumed by a MOCA_EP_LNFOLD GEMM"""
    __slots__ = ("x", "part", "nparts")

    def __init__(self, x, part, nparts):
        self.x, self.part, self.nparts = x, part, nparts


class _PlanBase:
    """pool + recorded launch list + hipGraph capture/replay, shared by the UNet plan and the VAE-decoder plan"""
    _defer_slabs = False     # (UNet plan only: every feature map it creates reaches gn() or _drop_colsum(), which release the deferred workspace)

    def __init__(self, model, device):
        self.model = model
        self.device = device
        self.P = model._packed
        self.pool = _Pool(device)
        self.stream = torch.cuda.Stream(device=device)
        self.steps = []
        self._pinned = set()
        self.graph = None
        self.graph_failed = False
        self.n_runs = 0
        self._gstat_buf, self._gstat_used, self._last_gemm_step = None, 0, None
        self._last_slabs = None
        self._held = {}          # data_ptr -> [pending deferred reduces that read it, buffer released meanwhile]
        self._gstat_full = []
        self.reps = 1            # > 1: the batch is `reps` context variants of the same Bx latents (_Plan: shared prefix)
        self._prefetch_at = {}   # during the build: index of a recorded GEMM step -> weights of later launches to prefetch in front of it
        self.prefetch_on = getattr(model, "weight_prefetch", True)

    def _note_gemm(self, pw):
        """called right before a GEMM step is recorded: weight-heavy -> its weights ride on the PREVIOUS GEMM launch (moca_gemm_params.prefetch:
        spare blocks of that launch's grid stream them into the Infinity Cache while its tiles compute; the small launches in between do
        not matter)"""
        if self._last_gemm_step is not None and not pw.derived and pw.w.numel() * pw.w.element_size() >= PREFETCH_MIN_BYTES:
            self._prefetch_at.setdefault(self._last_gemm_step, []).append(pw.w)
        self._last_gemm_step = len(self.steps)

    def _finish_prefetch(self):
        """end of the build (every re-targeting of recorded steps is done): the host launches get their `prefetch` operand (one per
        launch: the largest of the weights that asked for it)"""
        if self.prefetch_on:
            for j, ws in self._prefetch_at.items():
                prod = self.steps[j]
                kw = dict(prod.keywords)
                # the spare blocks must be done before the host's tiles are, or they prolong it: only hosts with enough work
                if 2.0 * kw["M"] * prod.args[1].N * prod.args[1].K < PREFETCH_HOST_FLOP:
                    continue
                kw["prefetch"] = max(ws, key=lambda w: w.numel())
                self.steps[j] = functools.partial(prod.func, *prod.args, **kw)
        self._prefetch_at = {}

    def close(self):
        """release the instantiated hipGraph (moca_graph_destroy); the plan falls back to eager launches if used again"""
        if self.graph is not None:
            g, self.graph = self.graph, None
            try:
                torch.cuda.synchronize(self.device)
                _l.load().moca_graph_destroy(g)
            except Exception:
                pass

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ emit helpers
    def _emit(self, fn, *args, **kw):
        self.steps.append(functools.partial(fn, *args, **kw))

    def _release(self, *bufs):
        for b in bufs:
            if b is None:
                continue
            if b.data_ptr() in self._pinned:
                continue
            held = self._held.get(b.data_ptr())
            if held is not None:               # an operand a deferred split-K reduce still has to read (see _gemm): back to the pool
                held[1] = b                    # only once the consumer that performs the reduce has been recorded (_unhold)
                continue
            self.pool.put(b)

    def _unhold(self, slabs):
        """the deferred reduce of `slabs` has been recorded (or dropped): operands released meanwhile go back to the pool NOW, i.e.
        after the fused GroupNorm took its output buffer -- it can never be handed one of the operands it re-reads"""
        for ptr in slabs[2]:
            held = self._held.get(ptr)
            if held is None:
                continue
            held[0] -= 1
            if held[0] == 0:
                del self._held[ptr]
                if held[1] is not None:
                    self.pool.put(held[1])

    def _splits(self, M, pw):
        return gemm_splits(M, pw)

    def _gemm(self, a, pw, M, *, out_cols=None, want_colsum=False, **kw):
        """returns `out`, or `(out, colsum)` with want_colsum: colsum = (f32 [row tiles][N][2], rows per tile) when this launch
        can leave the GroupNorm statistics of its output behind (moca_gemm_colsum_rows), else None"""
        n_out = pw.n_out if pw.geglu else pw.N
        out = self.pool.get(M, n_out)
        splits = self._splits(M, pw)
        ws = None
        if splits > 1:
            ws = self.pool.get(splits * M, pw.N, torch.float32)
        cs = None
        if want_colsum:
            rows = ops.gemm_colsum_rows(a, pw, M=M, splits=splits, **kw)
            if rows > 0:
                cs = (self.pool.get((M + rows - 1) // rows, 2 * pw.N, torch.float32), rows)
        self._note_gemm(pw)
        self._emit(ops.gemm, a, pw, out, M=M, splits=splits, splitk_ws=ws, colsum=None if cs is None else cs[0], **kw)
        self._last_slabs = None
        if ws is not None:
            if want_colsum and SPLITK_GN and self._defer_slabs:   # the workspace stays reserved until the consumer is known
                ptrs = []                                       # ... and so do the operands a fused reduce re-reads there
                for t in (kw.get("residual"), kw.get("rowadd")):
                    if t is not None and t.data_ptr() not in self._pinned:
                        self._held.setdefault(t.data_ptr(), [0, None])[0] += 1
                        ptrs.append(t.data_ptr())
                self._last_slabs = (ws, self._last_gemm_step, tuple(ptrs))   # (_FMap.slabs; released by gn() / _drop_colsum())
            else:
                self.pool.put(ws)
        return (out, cs) if want_colsum else out

    def linear(self, a, M, pw, residual=None, lda=None, want_colsum=False):
        return self._gemm(a, pw, M, lda=lda if lda is not None else a.stride(-2), residual=residual, want_colsum=want_colsum)

    def conv(self, fm, pw, *, stride=1, up=0, rowadd=None, rowadd_div=1, residual=None):
        """3x3 conv; the output carries its GroupNorm statistics when the launch can produce them"""
        if up:
            oH, oW = fm.H * 2, fm.W * 2
        elif stride == 2:
            oH, oW = (fm.H - 1) // 2 + 1, (fm.W - 1) // 2 + 1
        else:
            oH, oW = fm.H, fm.W
        M = fm.F * oH * oW
        out, cs = self._gemm(fm.buf, pw, M, mode=_l.MOCA_A_CONV3X3, conv=(fm.C, fm.H, fm.W, oH, oW, stride, up),
                             rowadd=rowadd, rowadd_div=rowadd_div, residual=residual, want_colsum=True)
        o = _FMap(out, fm.F, oH, oW, pw.N, cs, src=self._last_gemm_step if cs is not None else None)
        o.slabs = self._last_slabs
        return o

    def upconv(self, fm, conv):
        """`Upsample` (openaimodel3d.py:96-106: nearest x2, then conv3x3).  With enough low-resolution rows to fill the chip per
        launch: four 2 x 2 convs on the low-resolution grid, one per output parity, each writing its quarter of the pixels (4/9 of
        the FLOPs: the 3x3 taps that read the same input pixel are summed at pack time).  Otherwise the 3x3 conv whose gather
        reads in[y // 2][x // 2]."""
        phases = self.P.get(("up_phases", id(conv)))
        M = fm.F * fm.H * fm.W
        if phases is None or M < 5120 or self._splits(M, phases[0]) != 1:       # (below: split-k per phase costs more than it saves)
            return self.conv(fm, self.P[id(conv)], up=1)
        out = self.pool.get(4 * M, phases[0].N)
        kw = dict(M=M, mode=_l.MOCA_A_CONV3X3, conv=(fm.C, fm.H, fm.W, fm.H, fm.W, 1, 0), splits=1)
        idx = []
        for ph, pw in enumerate(phases):
            self._note_gemm(pw)
            idx.append(len(self.steps))
            self._emit(ops.gemm, fm.buf, pw, out, up_phase=ph + 1, **kw)
        up = _FMap(out, fm.F, fm.H * 2, fm.W * 2, phases[0].N, None)
        # the four launches can leave GroupNorm statistics of the upsampled map behind (a consumer that is a virtual concat re-targets
        # them): a row tile of the LOW-resolution grid lies inside one frame
        up.cs_rows = ops.gemm_colsum_rows(fm.buf, phases[0], **kw)
        if up.cs_rows > 0:
            up.src = tuple(idx)
        return up

    def tconv(self, fm, pw, residual=None):
        out, cs = self._gemm(fm.buf, pw, fm.M, mode=_l.MOCA_A_TCONV3, tconv=(fm.C, self.T, fm.H * fm.W), residual=residual,
                             want_colsum=True)
        o = _FMap(out, fm.F, fm.H, fm.W, pw.N, cs, src=self._last_gemm_step if cs is not None else None)
        o.slabs = self._last_slabs
        return o

    def gn(self, fm, gb, *, fps, eps, silu, x_dead=False):
        """GroupNorm(32) (+SiLU).  With `fm.colsum` (left by the producing GEMM) and row tiles that do not straddle frames the
        statistics pass over x disappears: finalize-from-column-sums + apply; otherwise the three-launch / slab path."""
        if isinstance(fm, _CatMap):                            # the virtual concat: both sources read in place, statistics finished
            y = self.pool.get(fm.M, fm.C)
            self._emit(ops.groupnorm_gstat_cat, fm.h.buf, fm.skip.buf, y, gb[0], gb[1], fm.gcat, fm.gb, F=fm.F, HW=fm.H * fm.W,
                       C1=fm.h.C, C2=fm.skip.C, frames_per_stat=fps, eps=eps, silu=silu, Fb=fm.Fb)
            assert fps == 1
            return y
        if getattr(fm, "slabs", None) is not None:
            # the producer ran split-K: its reduce launch moves into this GroupNorm (x_dead: nobody else reads fm.buf -- it is not written)
            slabs = fm.slabs
            wsk, src = slabs[0], slabs[1]
            prod = self.steps[src]
            kw = dict(prod.keywords)
            fm.slabs = None
            # (only per-frame statistics: 1024 slabs fill the chip and the launch beats reduce + GroupNorm by 2.3 us; the 64 slabs of the
            #  16-frame GroupNorms are slower fused, 18.6 against 7.9 + 7.9 us -- profiles/r05_ab_splitk_groupnorm.txt)
            if (fps == 1 or SPLITK_GN_ALL) and \
                    ops.gemm_splitk_groupnorm_ok(prod.args[0], prod.args[1], HW=fm.H * fm.W, frames_per_stat=fps, **kw):
                y = self.pool.get(fm.M, fm.C)
                kw["slabs"] = True
                self.steps[src] = functools.partial(prod.func, *prod.args, **kw)
                self._emit(ops.gemm_splitk_groupnorm, prod.args[0], prod.args[1], prod.args[2], y, gb[0], gb[1], HW=fm.H * fm.W,
                           frames_per_stat=fps, eps=eps, silu=silu, write_x=not x_dead, **kw)
                self.pool.put(wsk)
                self._unhold(slabs)
                return y
            self.pool.put(wsk)
            self._unhold(slabs)
        y = self.pool.get(fm.M, fm.C)
        HW = fm.H * fm.W
        ws = self.pool.get(1, ops.groupnorm_ws_floats(fm.F, HW, fm.C), torch.float32)
        cs = fm.colsum
        if fm.gstat is not None and fm.gstat[1] == fps:       # (the concat that produced fm accumulated the statistics)
            self._emit(ops.groupnorm_gstat, fm.buf, y, gb[0], gb[1], fm.gstat[0], F=fm.F, HW=HW, Cn=fm.C, frames_per_stat=fps,
                       eps=eps, silu=silu)
        elif cs is not None and (fps * HW) % cs[1] == 0 and fm.src is not None:
            # the producer is re-targeted: instead of per-tile column sums it accumulates the FINISHED statistics of this
            # GroupNorm (fixed-point atomics per (statistics group, channel group), MOCA_EP_GSTAT) -- no finalize launch
            prod = self.steps[fm.src]
            slot = self._gstat_slot((fm.F // fps) * 64)
            kw = dict(prod.keywords)
            kw["colsum"] = None
            kw["gstat"] = (slot, fps * HW)
            self.steps[fm.src] = functools.partial(prod.func, *prod.args, **kw)
            self.pool.put(cs[0])
            fm.colsum = None
            fm.gstat_own = (slot, fps)
            self._emit(ops.groupnorm_gstat, fm.buf, y, gb[0], gb[1], slot, F=fm.F, HW=HW, Cn=fm.C, frames_per_stat=fps,
                       eps=eps, silu=silu)
        elif cs is not None and (fps * HW) % cs[1] == 0:      # no row tile straddles two statistics groups
            fm.cs_used = True
            self._emit(ops.groupnorm_colsum, fm.buf, y, gb[0], gb[1], cs[0], tile_rows=cs[1], F=fm.F, HW=HW, Cn=fm.C,
                       frames_per_stat=fps, eps=eps, silu=silu, ws=ws)
        else:
            self._emit(ops.groupnorm, fm.buf, y, gb[0], gb[1], F=fm.F, HW=HW, Cn=fm.C, frames_per_stat=fps,
                       eps=eps, silu=silu, ws=ws)
        self.pool.put(ws)
        return y

    def _gstat_slot(self, n_doubles):
        """64-bit fixed-point accumulators of one GEMM -> GroupNorm pair, carved from ONE buffer that a single memset zeroes at the start of
        every run (_run_steps)"""
        if self._gstat_buf is None or self._gstat_used + n_doubles > self._gstat_buf.numel():
            if self._gstat_buf is not None:                # the chunk is full: keep it (its slots are in use), open another one
                self._gstat_full.append(self._gstat_buf[:self._gstat_used])
            self._gstat_buf = torch.zeros(max(1 << 20, n_doubles), dtype=torch.int64, device=self.device)      # 8 MiB chunks
            self._gstat_used = 0
        s = self._gstat_buf[self._gstat_used:self._gstat_used + n_doubles]
        self._gstat_used += n_doubles
        return s

    def _drop_colsum(self, fm):
        """the statistics buffer of a feature map goes back to the pool once its GroupNorm consumer has been recorded"""
        if getattr(fm, "slabs", None) is not None:             # a split-K producer whose consumer was no GroupNorm: its own reduce stays
            self.pool.put(fm.slabs[0])
            self._unhold(fm.slabs)
            fm.slabs = None
        if fm.colsum is not None:
            if not fm.cs_used and fm.src is not None:          # nobody read the column sums (the consumer was a conv / a concat /
                prod = self.steps[fm.src]                      # a GroupNorm whose statistics groups the row tiles straddle):
                kw = dict(prod.keywords)                       # the producer goes back to the plain store loop
                kw["colsum"] = None
                self.steps[fm.src] = functools.partial(prod.func, *prod.args, **kw)
            self.pool.put(fm.colsum[0])
            fm.colsum = None

    def ln(self, x, M, Cn, gb):
        y = self.pool.get(M, Cn)
        self._emit(ops.layernorm, x, y, gb[0], gb[1], M=M, Cn=Cn, eps=1e-5)
        return y

    def _run_steps(self):
        if self._prefetch_at:                  # plans whose build does not end in _finish_prefetch (VAE decoder, single blocks)
            self._finish_prefetch()
        for full in self._gstat_full:
            ops.memset_zero(full)
        if self._gstat_used:
            ops.memset_zero(self._gstat_buf[:self._gstat_used])
        for s in self.steps:
            s()

    def _launch(self, handle):
        lib = _l.load()
        if self.graph is not None:
            _l.check(lib.moca_graph_launch(self.graph, C.c_void_p(handle)), "moca_graph_launch")
            return
        use_graph = self.model.use_graph and not self.graph_failed
        if self.n_runs == 0 or not use_graph:
            self._run_steps()          # first pass eager: surfaces argument errors outside of capture
            return
        # second pass: record the same launch sequence into a hipGraph, then replay it
        rc = lib.moca_graph_begin(C.c_void_p(handle))
        if rc != 0:
            self.graph_failed = True
            warnings.warn("moca_video_amd: hipStreamBeginCapture failed; staying on eager HIP launches")
            self._run_steps()
            return
        try:
            self._run_steps()
        finally:
            g = C.c_void_p()
            rc = lib.moca_graph_end(C.c_void_p(handle), C.byref(g))
        if rc != 0 or not g.value:
            self.graph_failed = True
            warnings.warn("moca_video_amd: hipGraph instantiate failed; staying on eager HIP launches")
            self._run_steps()
            return
        self.graph = g
        _l.check(lib.moca_graph_launch(self.graph, C.c_void_p(handle)), "moca_graph_launch")


class _Plan(_PlanBase):
    _defer_slabs = True

    def __init__(self, model, B, T, H, W, L, in_dtype, device, shared_x=False):
        """shared_x: the B videos are `len(L)` context variants of the SAME B / len(L) latents (the two `apply_model` calls of
        classifier-free guidance, ddim.py:298-299,366-369, on one x): `x_in` holds the distinct latents only, everything up to the
        first cross-attention (conv_in, init_attn, the first ResBlock, the first SpatialTransformer's self-attention and to_q:
        8 % of the forward) is computed once and repeated where the contexts first enter (`_expand`)."""
        super().__init__(model, device)
        self.B, self.T, self.H, self.W, self.L = B, T, H, W, L
        self.BT = B * T
        # context segments (videos, tokens) in batch order: one segment normally; the batched FIFO call carries the conditional
        # windows (two prompts, 154 tokens) and the unconditional ones (77) in ONE forward: every cross-attention then runs one
        # launch per segment on its own rows of the shared K|V projection -- no padded or masked keys
        self.segs = [(B, L)] if isinstance(L, int) else [(int(n), int(l)) for n, l in L]
        assert sum(n for n, _ in self.segs) == B
        self.ctx_rows = sum(n * l for n, l in self.segs)
        m = model
        if shared_x:
            assert len(self.segs) > 1 and all(n == self.segs[0][0] for n, _ in self.segs)
            self.reps = len(self.segs)
        self.Bx = B // self.reps
        self.x_in = torch.empty(self.Bx, m.in_channels, T, H, W, dtype=in_dtype, device=device)
        self.t_rows = torch.empty(self.BT, dtype=torch.int64, device=device)
        self.fps_rows = torch.empty(self.BT, dtype=torch.int64, device=device)
        self.ctx = torch.empty(self.ctx_rows, m.context_dim, dtype=torch.float16, device=device)
        self.out = torch.empty(B, m.out_channels, T, H, W, dtype=in_dtype, device=device)
        self._build()

    # ------------------------------------------------------------------ blocks
    def res_block(self, mod, x):
        """ResBlock._forward, openaimodel3d.py:208-234"""
        P, HW = self.P, x.H * x.W
        g1 = self.gn(x, P[id(mod.in_layers[0])], fps=1, eps=1e-5, silu=True)
        off, width = self.model._emb_cols[id(mod)]
        emb_out = self.emb_all[:, off:off + width]               # this block's columns of the fused emb_layers GEMM
        h1 = self.conv(_FMap(g1, x.F, x.H, x.W, x.C), P[id(mod.in_layers[2])], rowadd=emb_out, rowadd_div=HW)
        self._release(g1)
        g2 = self.gn(h1, P[id(mod.out_layers[0])], fps=1, eps=1e-5, silu=True, x_dead=True)
        self._release(h1.buf)
        self._drop_colsum(h1)
        if isinstance(mod.skip_connection, torch.nn.Identity):
            sk = x.buf
        elif isinstance(x, _CatMap):                             # 1x1 conv over the virtual concat: two A sources, split at h's channels
            pw = P[id(mod.skip_connection)]
            sk = self.pool.get(x.M, pw.N)
            self._note_gemm(pw)
            self._emit(ops.gemm, x.h.buf, pw, sk, M=x.M, lda=x.h.buf.stride(-2), splits=1, a2=(x.skip.buf, x.h.C))
        else:
            sk = self.linear(x.buf, x.M, P[id(mod.skip_connection)])
        h2 = self.conv(_FMap(g2, x.F, x.H, x.W, mod.cout), P[id(mod.out_layers[3])], residual=sk)
        self._release(g2)
        if isinstance(x, _CatMap) or sk is not x.buf:
            self._release(sk)
        if not mod.use_temporal_conv:
            return h2
        # TemporalConvBlock.forward, openaimodel3d.py:269-276 (GroupNorm statistics over (C/32, T, H, W))
        tc = mod.temopral_conv
        cur = h2
        for i, (name, idx) in enumerate((("conv1", 2), ("conv2", 3), ("conv3", 3), ("conv4", 3))):
            sq = getattr(tc, name)
            g = self.gn(cur, P[id(sq[0])], fps=self.T, eps=1e-5, silu=True, x_dead=cur is not h2)
            self._drop_colsum(cur)
            nxt = self.tconv(_FMap(g, cur.F, cur.H, cur.W, cur.C), P[id(sq[idx])], residual=h2.buf if i == 3 else None)
            self._release(g)
            if cur is not h2:
                self._release(cur.buf)
            cur = nxt
        self._release(h2.buf)
        return cur

    # ---- LayerNorm folded into the linear that consumes it ------------------------------------------------------------
    def _folded(self, key, build):
        """LayerNorm-folded packed weights of one consumer (built on first use, kept beside the plain ones)"""
        k = ("lnfold", key)
        if k not in self.P:
            self.P[k] = build()
        return self.P[k]

    def _fold_pw(self, kind, mod, norm):
        """W' = W diag(gamma), b' = b + W beta for the linear `mod` that reads LayerNorm `norm` (ops.fold_layernorm)"""
        g, b = self.P[id(norm)]
        dev = self.device

        def build():
            if kind == "qkv":                                    # fused to_q | to_k | to_v (no biases, attention.py:54-56)
                w = torch.cat([mod.to_q.weight.detach(), mod.to_k.weight.detach(), mod.to_v.weight.detach()], dim=0).to(dev)
                wf, bf = ops.fold_layernorm(w, None, g, b)
                return ops.finish_lnfold(ops.pack_linear(wf, bf, device=dev))
            if kind == "q":
                wf, bf = ops.fold_layernorm(mod.to_q.weight.detach().to(dev), None, g, b)
                return ops.finish_lnfold(ops.pack_linear(wf, bf, device=dev))
            wf, bf = ops.fold_layernorm(mod.weight.detach().to(dev), mod.bias.detach().to(dev), g, b)      # GEGLU.proj
            return ops.finish_lnfold(ops.pack_geglu(wf, bf, device=dev))
        return self._folded(id(mod), build)

    def linear_of_ln(self, l, M, pw, fold_pw):
        """linear(LayerNorm(x)): `l` is either the LayerNorm output (a tensor) or an _LNRef (x + row partial sums left by its
        producer): then the fold runs in this GEMM's epilogue on the folded weights and no LayerNorm pass exists"""
        if isinstance(l, _LNRef):
            pwf = fold_pw()
            out = self.pool.get(M, pwf.n_out if pwf.geglu else pwf.N)
            self._note_gemm(pwf)
            self._emit(ops.gemm, l.x, pwf, out, M=M, lda=l.x.stride(-2), splits=1, lnfold=(l.part, l.nparts, 1e-5))
            return out
        return self.linear(l, M, pw)

    def _release_ln(self, l):
        self._release(l.part if isinstance(l, _LNRef) else l)

    def _tattn_pw(self, att, norm, heads, folded):
        """to_q|to_k|to_v packed per head for MOCA_EP_TATTN (LayerNorm-folded when the input arrives as an _LNRef)"""
        dev = self.device

        def build():
            wq, wk, wv = (m.weight.detach().to(dev) for m in (att.to_q, att.to_k, att.to_v))
            if not folded:
                return ops.pack_qkv_per_head(wq, wk, wv, heads, device=dev)
            g, b = self.P[id(norm)]
            wf, bf = ops.fold_layernorm(torch.cat([wq, wk, wv], dim=0), None, g, b)
            c = wq.shape[0]
            return ops.finish_lnfold(ops.pack_qkv_per_head(wf[:c], wf[c:2 * c], wf[2 * c:], heads, bias=bf, device=dev))
        k = ("tqkv_fold" if folded else "tqkv", id(att))
        if k not in self.P:
            self.P[k] = build()
        return self.P[k]

    def _attn_temporal_fused(self, att, l, M, Cn, heads, HW, norm):
        """q|k|v projection + attention over the frame axis in ONE launch (MOCA_EP_TATTN) where its 320-row tiles (16 frames x 20
        pixels, one head) come in whole rounds of the chip; None -> the caller runs projection + temporal_attention.
        Used wherever the kernel applies (round-2 same-device A/B of the whole CFG step: never 37.2 ms, only where the tile count is a
        multiple of 256 or >= 1024 36.4, everywhere 36.0 -- the launch and the q|k|v round trip it removes outweigh the partial last
        round of tiles at the 640- and 1280-channel levels)."""
        if self.T != 16 or HW % 20:
            return None
        folded = isinstance(l, _LNRef)
        x = l.x if folded else l
        pw = self._tattn_pw(att, norm, heads, folded)
        scale = att.dim_head ** -0.5
        kw = dict(M=M, lda=x.stride(-2), splits=1, tattn=(self.T, HW, scale))
        if not ops.gemm_tattn_ok(x, pw, lnfold=(None, l.nparts, 1e-5) if folded else None, **kw):
            return None
        o = self.pool.get(M, Cn)
        self._note_gemm(pw)
        self._emit(ops.gemm, x, pw, o, lnfold=(l.part, l.nparts, 1e-5) if folded else None, **kw)
        return o

    def _attn_self(self, att, l, M, Cn, heads, spatial, F, HW, norm=None):
        P = self.P
        if not spatial:
            o = self._attn_temporal_fused(att, l, M, Cn, heads, HW, norm)
            if o is not None:
                return o
        qkv = self.linear_of_ln(l, M, P[id(att)], lambda: self._fold_pw("qkv", att, norm))   # [M][3C] fused to_q|to_k|to_v
        o = self.pool.get(M, Cn)
        q, k, v = qkv[:, :Cn], qkv[:, Cn:2 * Cn], qkv[:, 2 * Cn:]
        scale = att.dim_head ** -0.5
        if spatial:
            self._emit(ops.attention, q, k, v, o, Bq=F, heads=heads, Nq=HW, Nk=HW, ldq=3 * Cn, ldk=3 * Cn, ldv=3 * Cn,
                       ldo=Cn, kv_div=1, scale=scale)
        else:
            self._emit(ops.temporal_attention, q, k, v, o, B=M // (self.T * HW), T=self.T, HW=HW, heads=heads, ld_qkv=3 * Cn, ldo=Cn,
                       scale=scale)
        self._release(qkv)
        return o

    def _attn_cross(self, att, l, M, Cn, heads, F, HW, norm=None, expand=False):
        """expand: `l` (and so q) still has the rows of the Bx distinct latents; every context segment attends with the SAME q rows
        and writes its own rows of o, which comes out with reps x M rows"""
        q = self.linear_of_ln(l, M, self.P[id(att)], lambda: self._fold_pw("q", att, norm))
        off, inner = self.model._kv_cols[id(att)]                # one K/V per video (context.repeat_interleave, :547),
        ld = self.kv_all.shape[1]                                # all layers' K|V projected by one GEMM up front
        o = self.pool.get(M * (self.reps if expand else 1), Cn)
        r0 = k0 = 0
        segs = self.segs
        if not expand and len(segs) > 1 and all(Ls == segs[0][1] for _, Ls in segs):
            segs = [(sum(nv for nv, _ in segs), segs[0][1])]     # equal context lengths: the segments are one contiguous batch
        for nv, Ls in segs:                                      # (one launch per context segment: rows of q / o, rows of K|V)
            rows = nv * self.T * HW
            kv = self.kv_all[k0:k0 + nv * Ls]
            self._emit(ops.attention, q[0:rows] if expand else q[r0:r0 + rows], kv[:, off:off + inner],
                       kv[:, off + inner:off + 2 * inner], o[r0:r0 + rows], Bq=nv * self.T, heads=heads, Nq=HW, Nk=Ls, ldq=Cn, ldk=ld,
                       ldv=ld, ldo=Cn, kv_div=self.T, scale=att.dim_head ** -0.5)
            r0 += rows
            k0 += nv * Ls
        self._release(q)
        return o

    def _expand(self, buf, rows, cols):
        """[rows][cols] -> reps copies (the branches of a shared-latents batch separate here)"""
        out = self.pool.get(rows * self.reps, cols)
        self._emit(ops.repeat, buf.reshape(-1)[:rows * cols], out, reps=self.reps)
        return out

    def linear_ln(self, a, M, pw, gb, residual=None, consumer=None, wgroup=None):
        """`out = linear(a) (+ residual)` and `l = LayerNorm(out)` (eps 1e-5, attention.py:199-201).
        (1) `consumer` = a callable returning the LayerNorm-folded weights of the ONE linear that reads l: when this launch can
        leave row sums behind (MOCA_EP_ROWSUM) and the consumer's kernel has the fold epilogue (MOCA_EP_LNFOLD), l is an
        _LNRef and LayerNorm never touches memory; (2) ONE launch writing out and LayerNorm(out) where the 160 x 320 tiling
        applies (N = 320: a block owns whole rows and normalises them in its store loop, MOCA_EP_LN); (3) otherwise the linear
        followed by the LayerNorm kernel."""
        lda = a.stride(-2)
        splits = self._splits(M, pw)
        wk = {} if wgroup is None else {"wgroup": wgroup}      # (`pw` = per-row-group weights: a GroupNorm folded in, see transformer())
        ln_epilogue = ops.gemm_ln_ok(a, pw, M=M, lda=lda, residual=residual, splits=splits, ln=(gb[0], gb[1], None, 1e-5), **wk)
        # Measured (same device, whole CFG step, tools/ab_run7.sh): fold wherever possible 36.9 ms, fold only where neither the
        # LayerNorm store loop (N = 320) nor more than two row partials apply 37.3, never 37.6.  In isolation the consumers pay
        # 3-12 % for the two FMAs per accumulator and the statistics loads (tools/bench_lnfold.py; worst at K = 320 and with the
        # 10 partials of the 1280-channel level), less than the LayerNorm pass and the second output they replace cost in the
        # graph.  So: fold wherever producer and consumer kernels allow.
        if consumer is not None and splits == 1:
            cols = ops.gemm_rowsum_cols(a, pw, M=M, lda=lda, residual=residual, splits=1, rowsum=True, **wk)
            nparts = pw.N // cols if cols > 0 else 0
            if cols > 0:
                pwf = consumer()
                if ops.gemm_lnfold_ok(a, pwf, M=M, lda=pw.N, splits=self._splits(M, pwf), lnfold=(None, nparts, 1e-5)) and \
                        self._splits(M, pwf) == 1:
                    out = self.pool.get(M, pw.N)
                    part = self.pool.get(nparts * M, 2, torch.float32)
                    self._note_gemm(pw)
                    self._emit(ops.gemm, a, pw, out, M=M, lda=lda, residual=residual, splits=1, rowsum=part, **wk)
                    return out, _LNRef(out, part, nparts)
        if ln_epilogue:
            out, l = self.pool.get(M, pw.N), self.pool.get(M, pw.N)
            self._note_gemm(pw)
            self._emit(ops.gemm, a, pw, out, M=M, lda=lda, residual=residual, splits=1, ln=(gb[0], gb[1], l, 1e-5), **wk)
            return out, l
        assert wgroup is None or splits == 1
        out = self._gemm(a, pw, M, lda=lda, residual=residual, **wk)
        return out, self.ln(out, M, pw.N, gb)

    def tblock(self, blk, h, l, M, Cn, heads, spatial, F, HW, next_gb=None, next_consumer=None):
        """BasicTransformerBlock._forward, attention.py:216-220.  `l` = norm1(h), already computed by the producer of h;
        returns (h_out, norm1 of the NEXT block applied to it or None, rows, frames) -- rows / frames grow by `reps` when this
        block holds the first cross-attention of a shared-latents plan."""
        P = self.P
        geglu = blk.ff.net[0].proj
        consumers = (self._ln_consumer(blk.attn2, blk.norm2), lambda: self._fold_pw("geglu", geglu, blk.norm3))
        for (att, cur, nxt), cons in zip(((blk.attn1, blk.norm1, blk.norm2), (blk.attn2, blk.norm2, blk.norm3)), consumers):
            if att.is_self:
                o = self._attn_self(att, l, M, Cn, heads, spatial, F, HW, norm=cur)
            else:
                expand = self.reps > 1 and F == self.Bx * self.T      # first use of the contexts: the shared prefix ends here
                o = self._attn_cross(att, l, M, Cn, heads, F, HW, norm=cur, expand=expand)
                if expand:
                    h2 = self._expand(h, M, Cn)
                    self._release(h)
                    h, M, F = h2, M * self.reps, F * self.reps
            self._release_ln(l)
            nh, l = self.linear_ln(o, M, P[id(att.to_out[0])], P[id(nxt)], residual=h, consumer=cons)
            self._release(o, h)
            h = nh
        ff = self.linear_of_ln(l, M, P[id(geglu)], consumers[1])  # GEGLU fused into the epilogue
        self._release_ln(l)
        if next_gb is not None:
            nh, l = self.linear_ln(ff, M, P[id(blk.ff.net[2])], next_gb, residual=h, consumer=next_consumer)
        else:
            nh, l = self.linear(ff, M, P[id(blk.ff.net[2])], residual=h), None
        self._release(ff, h)
        return nh, l, M, F

    def _ln_consumer(self, att, norm):
        """the folded weights of the projection that reads `norm` in front of attention `att`"""
        return lambda: self._fold_pw("qkv" if att.is_self else "q", att, norm)

    def _gn_fold(self, fm, gb, pw, *, fps, eps):
        """GroupNorm(32, no activation) -> Linear as ONE GEMM on per-statistics-group weights, where (a) the producer of `fm` can leave the
        FINISHED statistics behind (the condition of gn()'s single-launch path: a GEMM whose row tiles lie inside one statistics group, or
        a concat that accumulated them) and (b) the linear runs on a staggered kernel whose row tiles lie inside one group
        (moca_gemm_wgroup_ok).  Returns (per-group PackedWeight, (rows per group, stride)) or None (the caller runs gn() + the linear)."""
        if not GN_FOLD or isinstance(fm, _CatMap) or getattr(fm, "slabs", None) is not None or fm.C % 32 or pw.geglu:
            return None
        HW, rows = fm.H * fm.W, fps * fm.H * fm.W
        cs = fm.colsum
        have = fm.gstat is not None and fm.gstat[1] == fps
        can = cs is not None and rows % cs[1] == 0 and fm.src is not None and not isinstance(fm.src, tuple)
        # worth it only while the derived weights (written once, read by the group's row tiles) are small against the tensor the GroupNorm
        # pass would read and write: rows per group >= 4 N.  (640-channel level, per-frame statistics: 640 rows per group -- 256 frames x
        # 819 KB of weights at B = 16, as many bytes as the pass itself: measured no gain; that norm stays a launch)
        if not (have or can) or fm.F % fps or rows < 4 * pw.N:
            return None
        stride = pw.N * pw.w.stride(0)
        if not ops.gemm_wgroup_ok(fm.buf, pw, M=fm.M, lda=fm.buf.stride(-2), splits=self._splits(fm.M, pw), wgroup=(rows, stride)):
            return None
        n_sg = fm.F // fps
        if have:
            slot = fm.gstat[0]
        else:                                        # re-target the producer: finished statistics instead of per-tile column sums (as gn())
            prod = self.steps[fm.src]
            slot = self._gstat_slot(n_sg * 64)
            kw = dict(prod.keywords)
            kw["colsum"] = None
            kw["gstat"] = (slot, rows)
            self.steps[fm.src] = functools.partial(prod.func, *prod.args, **kw)
            self.pool.put(cs[0])
            fm.colsum = None
            fm.gstat_own = (slot, fps)
        wg = self.pool.get(n_sg * pw.N, pw.w.stride(0))
        bg = self.pool.get(1, n_sg * pw.N, torch.float32)
        self._emit(ops.groupnorm_fold_weights, pw, gb[0], gb[1], slot, wg, bg, n_sg=n_sg, count=rows * (fm.C // 32), eps=eps)
        pwg = ops.PackedWeight(wg, bg.reshape(-1), pw.N, pw.K, pw.n_out)
        pwg.derived = True
        return pwg, (rows, stride)

    def transformer(self, mod, x, spatial):
        """SpatialTransformer.forward attention.py:262-278 / TemporalTransformer.forward :331-373"""
        P = self.P
        blocks = list(mod.transformer_blocks)
        fold = self._gn_fold(x, P[id(mod.norm)], P[id(mod.proj_in)], fps=1 if spatial else self.T, eps=1e-6)
        if fold is not None:
            # `x = self.norm(x)` (attention.py:262-268 / :333-341) lives in proj_in's per-statistics-group weights: no GroupNorm launch,
            # the normalised tensor is never written
            pwg, wgroup = fold
            h, l = self.linear_ln(x.buf, x.M, pwg, P[id(blocks[0].norm1)], consumer=self._ln_consumer(blocks[0].attn1, blocks[0].norm1),
                                  wgroup=wgroup)
            self._release(pwg.w, pwg.bias)
        else:
            n = self.gn(x, P[id(mod.norm)], fps=1 if spatial else self.T, eps=1e-6, silu=False)
            h, l = self.linear_ln(n, x.M, P[id(mod.proj_in)], P[id(blocks[0].norm1)],
                                  consumer=self._ln_consumer(blocks[0].attn1, blocks[0].norm1))
            self._release(n)
        M, Fr, xres = x.M, x.F, x.buf
        for i, blk in enumerate(blocks):
            last = i + 1 >= len(blocks)
            nxt = None if last else P[id(blocks[i + 1].norm1)]
            ncons = None if last else self._ln_consumer(blocks[i + 1].attn1, blocks[i + 1].norm1)
            h, l, M, Fr = self.tblock(blk, h, l, M, mod.inner, mod.heads, spatial, Fr, x.H * x.W, next_gb=nxt, next_consumer=ncons)
        if Fr != x.F:                                            # the shared prefix ended inside: the outer residual too
            xres = self._expand(x.buf, x.M, x.C)
        out, cs = self.linear(h, M, P[id(mod.proj_out)], residual=xres, want_colsum=True)
        self._release(h)
        if xres is not x.buf:
            self._release(xres)
        o = _FMap(out, Fr, x.H, x.W, x.C, cs, src=self._last_gemm_step if cs is not None else None)
        o.slabs = self._last_slabs          # (split-K proj_out, e.g. M = 640 at B = 1: released by gn() / _drop_colsum())
        return o

    def run_seq(self, seq, h):
        """TimestepEmbedSequential.forward, openaimodel3d.py:36-48"""
        for layer in seq:
            if isinstance(layer, _ResBlock):
                nh = self.res_block(layer, h)
            elif isinstance(layer, _SpatialTransformer):
                nh = self.transformer(layer, h, True)
            elif isinstance(layer, _TemporalTransformer):
                nh = self.transformer(layer, h, False)
            elif isinstance(layer, _Downsample):
                nh = self.conv(h, self.P[id(layer.op)], stride=2)
            elif isinstance(layer, _Upsample):
                nh = self.upconv(h, layer.conv)
            else:
                raise TypeError(type(layer))
            if isinstance(h, _CatMap):
                self._drop_colsum(h.h)          # (whatever _virtual_cat did not re-target: column sums nobody read, a deferred workspace)
                self._drop_colsum(h.skip)
                self._release(h.h.buf, h.skip.buf)
            else:
                self._release(h.buf)
                self._drop_colsum(h)
            h = nh
        return h

    # ------------------------------------------------------------------ whole forward
    def _embed_mlp(self, sq, rows_t):
        """timestep_embedding -> Linear -> SiLU -> Linear (openaimodel3d.py:362-372,536-543)"""
        m, P, BT = self.model, self.P, self.BT
        te = self.pool.get(BT, m.model_channels)
        self._emit(ops.timestep_embedding, rows_t, te, n=BT, dim=m.model_channels)
        e1 = self.linear(te, BT, P[id(sq[0])])
        s1 = self.pool.get(BT, e1.shape[1])
        self._emit(ops.silu_add_rows, e1, 1, None, 1, s1, rows=BT, Cn=e1.shape[1], silu=True)
        e2 = self.linear(s1, BT, P[id(sq[2])])
        self._release(te, e1, s1)
        return e2

    def _build(self):
        m, P, B, T, H, W = self.model, self.P, self.B, self.T, self.H, self.W
        BT = self.BT
        emb = self._embed_mlp(m.time_embed, self.t_rows)
        femb = self._embed_mlp(m.fps_embedding, self.fps_rows) if m.fps_cond else None
        # every consumer of `emb` is ResBlock.emb_layers = SiLU -> Linear (openaimodel3d.py:166-172)
        self.emb_silu = self.pool.get(BT, emb.shape[1])
        self._emit(ops.silu_add_rows, emb, 1, femb, 1, self.emb_silu, rows=BT, Cn=emb.shape[1], silu=True)
        self._release(emb, femb)
        self._pinned.add(self.emb_silu.data_ptr())
        self.emb_all = self.linear(self.emb_silu, BT, P["emb_all"])          # [BT][sum Cout]: every ResBlock's emb_layers
        self._pinned.add(self.emb_all.data_ptr())
        self.kv_all = None
        if "ctx_kv_all" in P:
            self.kv_all = self.linear(self.ctx, self.ctx_rows, P["ctx_kv_all"])  # [sum B_i L_i][sum 2C]: every cross-attention K|V
            self._pinned.add(self.kv_all.data_ptr())

        FT = self.Bx * T                                         # frames of the distinct latents (= BT unless shared_x)
        x8 = self.pool.get(FT * H * W, 8)
        self._emit(ops.ncthw_to_nhwc, self.x_in, x8, B=self.Bx, Cin=m.in_channels, T=T, HW=H * W, Cpad=8)
        h = self.conv(_FMap(x8, FT, H, W, 8), P[id(m.input_blocks[0][0])])
        self._release(x8)
        hs = []
        for i, module in enumerate(m.input_blocks):
            if i == 0:
                if m.addition_attention:
                    nh = self.transformer(m.init_attn[0], h, False)
                    self._release(h.buf)
                    self._drop_colsum(h)
                    h = nh
            else:
                h = self.run_seq(module, h)
            hs.append(h)
            self._pinned.add(h.buf.data_ptr())
        h = self.run_seq(m.middle_block, h)
        if h.F != BT:                                            # (no cross-attention anywhere: nothing ever separated the branches)
            h = _FMap(self._expand(h.buf, h.M, h.C), BT, h.H, h.W, h.C)
        for module in m.output_blocks:
            skip = hs.pop()
            if skip.F != h.F:                                    # a skip connection from the shared prefix
                self._pinned.discard(skip.buf.data_ptr())
                wide = self._expand(skip.buf, skip.M, skip.C)
                self._drop_colsum(skip)
                self._release(skip.buf)
                own, own_F = skip.gstat_own, skip.F
                skip = _FMap(wide, h.F, skip.H, skip.W, skip.C)
                skip.gstat_own, skip.own_F = own, own_F          # (the copies share the statistics of the frames they repeat)
            vc = self._virtual_cat(module, h, skip)
            if vc is not None:
                self._pinned.discard(skip.buf.data_ptr())
                h = self.run_seq(module, vc)                     # (run_seq releases h.buf and skip.buf behind the ResBlock)
                continue
            cat = self.pool.get(h.M, h.C + skip.C)
            gst = None
            Cc = h.C + skip.C
            if isinstance(module[0], _ResBlock) and Cc % 32 == 0 and \
                    64 <= (Cc // 8) * max(1, 256 // (Cc // 8)) and Cc // 8 <= 1024:
                # torch.cat(dim=1), :571, leaving the statistics of ResBlock.in_layers[0] (per-frame GroupNorm) behind
                gst = (self._gstat_slot(h.F * 64), 1)
                self._emit(ops.concat_channels_gstat, h.buf, skip.buf, cat, gst[0], F=h.F, HW=h.H * h.W, C1=h.C, C2=skip.C,
                           frames_per_stat=1)
            else:
                self._emit(ops.concat_channels, h.buf, skip.buf, cat, rows=h.M, C1=h.C, C2=skip.C)   # torch.cat(dim=1), :571
            self._pinned.discard(skip.buf.data_ptr())
            self._drop_colsum(h)          # (a concat consumed h: its producer goes back to the plain store loop)
            self._drop_colsum(skip)
            self._release(h.buf, skip.buf)
            h = self.run_seq(module, _FMap(cat, h.F, h.H, h.W, Cc, gstat=gst))
        g = self.gn(h, P[id(m.out[0])], fps=1, eps=1e-5, silu=True)
        self._release(h.buf)
        self._drop_colsum(h)
        o = self.conv(_FMap(g, h.F, h.H, h.W, h.C), P[id(m.out[2])])
        self._release(g)
        self._emit(ops.nhwc_to_ncthw, o.buf, o.C, self.out, B=B, Cout=m.out_channels, T=T, HW=H * W)
        self._release(o.buf)
        self._finish_prefetch()

    def _virtual_cat(self, module, h, skip):
        """torch.cat([h, skip], dim=1) (openaimodel3d.py:571) WITHOUT the copy, where both consumers can read two sources: the
        ResBlock's in_layers GroupNorm (moca_groupnorm_gstat_cat_f16) and its skip_connection 1x1 conv (moca_gemm_params.a2, the
        staggered kernels: the 320- / 640-channel levels).  The statistics of the concat's 32 groups come from the producers: h's
        last GEMM is re-targeted to accumulate ITS share in the concat's grouping (gstat_cpg); skip's share is its own finished
        per-frame statistics when its producer already left them for the input path (merged by the GroupNorm), else its producer
        is re-targeted likewise; a source without such a producer (an `Upsample`, the repeat that ends the shared prefix) gets a
        read-only statistics pass.  Returns a _CatMap, or None (the caller materialises the concat)."""
        if not VIRTUAL_CAT or not isinstance(module[0], _ResBlock) or isinstance(module[0].skip_connection, torch.nn.Identity):
            return None
        Cc, HW = h.C + skip.C, h.H * h.W
        if Cc % 32 or h.C % 64 or skip.C % 64 or Cc // 8 > 1024 or h.F != skip.F:
            return None
        pw = self.P[id(module[0].skip_connection)]
        if self._splits(h.M, pw) != 1 or not ops.gemm_cat_ok(h.buf, pw, M=h.M, lda=h.buf.stride(-2), splits=1, a2=(skip.buf, h.C)):
            return None
        gw = Cc // 32
        slot = self._gstat_slot(h.F * 64)

        def retarget(fm, coff):
            """fm's producer accumulates fm's share of the concat's per-frame statistics (a row tile must lie inside one frame)"""
            rows = fm.colsum[1] if fm.colsum is not None else fm.cs_rows
            phases = isinstance(fm.src, tuple)                   # an `Upsample` as four 2 x 2 convs on the low-resolution grid
            grp = HW // 4 if phases else HW                     # GEMM rows per frame
            if fm.src is None or rows <= 0 or grp % rows or fm.gstat_own is not None or fm.cs_used:
                return False
            for i in (fm.src if phases else (fm.src,)):
                prod = self.steps[i]
                kw = dict(prod.keywords)
                if kw.get("gstat") is not None or kw.get("rowsum") is not None:
                    return False
                kw["colsum"] = None
                kw["gstat"] = (slot, grp, gw, coff)
                self.steps[i] = functools.partial(prod.func, *prod.args, **kw)
            if fm.colsum is not None:
                self.pool.put(fm.colsum[0])
                fm.colsum = None
            fm.src = None
            return True

        if not retarget(h, 0):
            self._drop_colsum(h)
            self._emit(ops.gstat_accum, h.buf, slot, F=h.F, HW=HW, Cn=h.C, frames_per_stat=1, cpg=gw, coff=0)
        gb, Fb = None, 0
        cpg2 = skip.C // 32
        if skip.gstat_own is not None and skip.gstat_own[1] == 1 and skip.C % 32 == 0 and gw % cpg2 == 0 and (h.C % gw) % cpg2 == 0:
            gb, Fb = skip.gstat_own[0], skip.own_F
        elif not retarget(skip, h.C):
            self._drop_colsum(skip)
            self._emit(ops.gstat_accum, skip.buf, slot, F=skip.F, HW=HW, Cn=skip.C, frames_per_stat=1, cpg=gw, coff=h.C)
        return _CatMap(h, skip, slot, gb, Fb)

    def set_context(self, context):
        """context [B, L, D], or one [n_i, L_i, D] tensor per segment"""
        if torch.is_tensor(context):
            self.ctx.copy_(context.reshape(self.ctx_rows, -1), non_blocking=True)
            return
        r = 0
        for (nv, Ls), c in zip(self.segs, context):
            self.ctx[r:r + nv * Ls].copy_(c.reshape(nv * Ls, -1), non_blocking=True)
            r += nv * Ls

    def launch_async(self, x, t_rows, fps_rows, context, cur):
        """enqueue one forward on this plan's stream (ordered after `cur`); the caller joins"""
        self.stream.wait_stream(cur)
        with torch.cuda.stream(self.stream):
            self.x_in.copy_(x, non_blocking=True)
            self.t_rows.copy_(t_rows, non_blocking=True)
            self.fps_rows.copy_(fps_rows, non_blocking=True)
            self.set_context(context)
            handle = self.stream.cuda_stream
            ops.set_stream(handle)
            try:
                self._launch(handle)
            finally:
                ops.set_stream(None)
            out = self.out.clone()
        self.n_runs += 1
        return out

    def run(self, x, t_rows, fps_rows, context):
        cur = torch.cuda.current_stream(self.device)
        out = self.launch_async(x, t_rows, fps_rows, context, cur)
        out.record_stream(cur)
        cur.wait_stream(self.stream)
        return out

