"""Run ONE block of the UNet through the recorded-plan machinery of `plan.py` -- a `ResBlock` (with its
`TemporalConvBlock`), a `SpatialTransformer`, a `TemporalTransformer` (linear or the Conv1d-projected `init_attn`
flavour), a `Downsample` or an `Upsample` -- with the reference's per-block call signature:

    ResBlock.forward(x, emb, batch_size)              openaimodel3d.py:195-234
    SpatialTransformer.forward(x, context)            attention.py:262-278
    TemporalTransformer.forward(x)                    attention.py:331-373
    Downsample / Upsample .forward(x)                 openaimodel3d.py:56-121

Same kernels, same launch recording and the same hipGraph replay as the whole-UNet plan (the block methods ARE
`_Plan.res_block / transformer / conv`); what this adds is only the packing of a single block and the inputs the UNet
would have prepared for it (SiLU(emb) -> the fused emb_layers GEMM, the context K/V GEMM).  Used by the block-level
parity tests (tests/test_unet_gpu.py) and handy for profiling one block in isolation."""
from __future__ import annotations

import torch
import torch.nn as nn

from . import ops
from .plan import _Plan, _PlanBase
from .unet import (_Downsample, _FMap, _ResBlock, _SpatialTransformer, _TemporalTransformer, _Upsample, pack_tree)


class _Host(nn.Module):
    """what `_PlanBase` / `_Plan` read from the model object"""

    def __init__(self, block):
        super().__init__()
        self.block = block
        self.use_graph = True
        self._packed = None


class BlockRunner:
    def __init__(self, block, B, T, H, W, L=0, context_dim=None):
        if not isinstance(block, (_ResBlock, _SpatialTransformer, _TemporalTransformer, _Downsample, _Upsample)):
            raise TypeError(type(block))
        dev = next(block.parameters()).device
        if dev.type != "cuda":
            raise RuntimeError("BlockRunner runs on an MI355X only (no CPU path)")
        self.block, self.B, self.T, self.H, self.W, self.L = block, B, T, H, W, L
        host = _Host(block)
        host._packed, host._emb_cols, host._kv_cols = pack_tree(host, dev)
        pl = _Plan.__new__(_Plan)
        _PlanBase.__init__(pl, host, dev)
        pl.B, pl.T, pl.H, pl.W, pl.L, pl.BT = B, T, H, W, L, B * T
        pl.segs, pl.ctx_rows = [(B, L)], B * L
        self.plan = pl
        P = host._packed
        if isinstance(block, _ResBlock):
            cin = block.cin
        elif isinstance(block, _Downsample):
            cin = block.op.weight.shape[1]
        elif isinstance(block, _Upsample):
            cin = block.conv.weight.shape[1]
        else:
            cin = block.ch
        self.cin = cin
        M = B * T * H * W
        self.x_in = torch.empty(M, cin, dtype=torch.float16, device=dev)
        self.emb_in = self.ctx_in = None
        if "emb_all" in P:                      # ResBlock: emb_layers = SiLU -> Linear (openaimodel3d.py:166-172)
            emb_ch = block.emb_layers[1].weight.shape[1]
            self.emb_in = torch.empty(B * T, emb_ch, dtype=torch.float16, device=dev)
            silu = pl.pool.get(B * T, emb_ch)
            pl._emit(ops.silu_add_rows, self.emb_in, 1, None, 1, silu, rows=B * T, Cn=emb_ch, silu=True)
            pl.emb_all = pl.linear(silu, B * T, P["emb_all"])
            pl._pinned.add(pl.emb_all.data_ptr())
        pl.kv_all = None
        if "ctx_kv_all" in P:                   # cross-attention K|V of the context, one row block per video
            self.ctx_in = torch.empty(B * L, context_dim, dtype=torch.float16, device=dev)
            pl.kv_all = pl.linear(self.ctx_in, B * L, P["ctx_kv_all"])
            pl._pinned.add(pl.kv_all.data_ptr())
        pl._pinned.add(self.x_in.data_ptr())
        self.out = pl.run_seq([block], _FMap(self.x_in, B * T, H, W, cin))
        pl._pinned.add(self.out.buf.data_ptr())

    @torch.no_grad()
    def __call__(self, x, emb=None, context=None):
        """x: [B*T, C, H, W] (the reference's `(b t) c h w`); emb: [B*T, emb_ch]; context: [B, L, context_dim].
        Returns [B*T, C', H', W'] fp32.  Call 1 runs the recorded launches eagerly, call 2 captures the hipGraph,
        later calls replay it."""
        pl = self.plan
        cur = torch.cuda.current_stream(x.device)
        pl.stream.wait_stream(cur)
        with torch.cuda.stream(pl.stream):
            self.x_in.copy_(x.permute(0, 2, 3, 1).reshape(-1, self.cin))
            if self.emb_in is not None:
                self.emb_in.copy_(emb)
            if self.ctx_in is not None:
                self.ctx_in.copy_(context.reshape(self.B * self.L, -1))
            ops.set_stream(pl.stream.cuda_stream)
            try:
                pl._launch(pl.stream.cuda_stream)
            finally:
                ops.set_stream(None)
            o = self.out
            y = o.buf.view(o.F, o.H, o.W, o.C).permute(0, 3, 1, 2).float()
        pl.n_runs += 1
        cur.wait_stream(pl.stream)
        return y
