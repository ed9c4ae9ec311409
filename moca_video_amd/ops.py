"""Thin tensor-level wrappers over the C-ABI (moca_video_amd/lib.py) plus the weight
pre-packing that turns reference-shaped parameters (nn.Conv2d / nn.Conv3d / nn.Linear
state-dict tensors) into the K-contiguous fp16 matrices the implicit-GEMM kernel reads.

Every function launches on `ops.current_stream()` (a raw hipStream_t owned by a
torch.cuda.Stream) and raises on failure.  Nothing here computes on the CPU or
through torch kernels: torch only allocates the buffers.
"""
from __future__ import annotations

import ctypes as C
import os
import torch

from . import lib as _l

_stream = None  # raw hipStream_t (int) all launches go to


def set_stream(handle):
    global _stream
    _stream = handle


def current_stream():
    if _stream is None:
        # torch's current stream on the current device
        return torch.cuda.current_stream().cuda_stream
    return _stream


def _st():
    return C.c_void_p(current_stream())


def _round_up(x, m):
    return (x + m - 1) // m * m


# --------------------------------------------------------------------------------------
# weight packing (runs once at load_state_dict time; plain torch ops on the weights)
# --------------------------------------------------------------------------------------
class PackedWeight:
    """fp16 [Npad][Kpad] K-contiguous weight + fp32 bias, as moca_gemm_f16 wants them."""
    __slots__ = ("w", "bias", "N", "K", "n_out", "geglu", "wsum", "derived")

    def __init__(self, w, bias, N, K, n_out, geglu=False):
        self.w, self.bias, self.N, self.K, self.n_out, self.geglu = w, bias, N, K, n_out, geglu
        self.wsum = None         # LayerNorm-folded weights only (fold_layernorm): f32 [N], row sums of the packed fp16 W'
        self.derived = False     # True: written by a launch of the forward itself (groupnorm_fold_weights) -- nothing to prefetch


def _finish(w2d: torch.Tensor, bias, device, n_out=None, geglu=False) -> PackedWeight:
    n, k = w2d.shape
    npad, kpad = _round_up(n, 64), _round_up(k, 64)
    w = torch.zeros(npad, kpad, dtype=torch.float16, device=device)
    w[:n, :k] = w2d.to(device=device, dtype=torch.float16)
    b = None
    if bias is not None:
        b = torch.zeros(npad, dtype=torch.float32, device=device)
        b[:n] = bias.to(device=device, dtype=torch.float32)
    return PackedWeight(w, b, npad, k, n if n_out is None else n_out, geglu)


def pack_linear(weight, bias=None, device="cuda"):
    """nn.Linear weight [N][K] (or Conv1d k=1 [N][K][1])."""
    w = weight.reshape(weight.shape[0], -1)
    return _finish(w, bias, device)


def pack_linear_cat(weights, biases=None, device="cuda"):
    """Fuse several linears sharing an input (to_q/to_k/to_v) into one [sum N][K] GEMM."""
    w = torch.cat([x.reshape(x.shape[0], -1) for x in weights], dim=0)
    b = None
    if biases is not None:
        b = torch.cat(list(biases), dim=0)
    return _finish(w, b, device)


def pack_conv3x3(weight, bias=None, cpad=None, device="cuda"):
    """nn.Conv2d weight [N][C][3][3] -> [N][(ky,kx,c)], channels zero padded to cpad."""
    n, c = weight.shape[0], weight.shape[1]
    cpad = c if cpad is None else cpad
    w = torch.zeros(n, 3, 3, cpad, dtype=weight.dtype, device=weight.device)
    w[..., :c] = weight.permute(0, 2, 3, 1)
    return _finish(w.reshape(n, 9 * cpad), bias, device)


def pack_upconv_phases(weight, bias=None, device="cuda"):
    """`Upsample` = nearest x2 + conv3x3 (openaimodel3d.py:96-106) as four 2 x 2 convs on the low-resolution grid, one per output
    parity (a, b): output pixel (2i+a, 2j+b) reads upsampled rows 2i+a-1 .. 2i+a+1 = input rows {i-1, i, i} (a = 0) or {i, i, i+1}
    (a = 1), so the 3x3 taps that fall on the same input pixel are SUMMED (in fp32, then rounded to fp16 like every weight): tap r of
    phase a covers ky in ((0,), (1, 2)) for a = 0 and ((0, 1), (2,)) for a = 1, the same along x.  The zero padding of the upsampled
    image coincides with the zero padding of the input (a summed pair never straddles the border).  Returns the four PackedWeights
    [N][(r, s, c)] in phase order 1 + 2a + b; 4/9 of the multiply-adds of the 3x3 conv on the upsampled image."""
    n, c = weight.shape[0], weight.shape[1]
    w32 = weight.detach().to(torch.float32)
    rows = (((0,), (1, 2)), ((0, 1), (2,)))
    out = []
    for a in range(2):
        for b in range(2):
            w = torch.zeros(n, 2, 2, c, dtype=torch.float32, device=weight.device)
            for r in range(2):
                for s_ in range(2):
                    for ky in rows[a][r]:
                        for kx in rows[b][s_]:
                            w[:, r, s_, :] += w32[:, :, ky, kx]
            out.append(_finish(w.reshape(n, 4 * c), bias, device))
    return out


def pack_conv1x1(weight, bias=None, device="cuda"):
    return _finish(weight.reshape(weight.shape[0], weight.shape[1]), bias, device)


def pack_tconv3(weight, bias=None, device="cuda"):
    """nn.Conv3d weight [N][C][3][1][1] -> [N][(kt,c)]."""
    n, c = weight.shape[0], weight.shape[1]
    w = weight.reshape(n, c, 3).permute(0, 2, 1).reshape(n, 3 * c)
    return _finish(w, bias, device)


def pack_geglu(weight, bias, device="cuda"):
    """GEGLU.proj weight [2*inner][K]: value rows first, gate rows second (attention.py:382).
    Re-ordered in 64-row groups (32 value rows, then their 32 gate rows) so that one wave's
    two 32-wide accumulator tiles hold value and gate of the same output columns."""
    two_inner, k = weight.shape
    inner = two_inner // 2
    assert inner % 32 == 0
    idx = torch.arange(inner, device=weight.device).reshape(inner // 32, 32)
    order = torch.cat([idx, idx + inner], dim=1).reshape(-1)
    p = _finish(weight[order], bias[order] if bias is not None else None, device, n_out=inner, geglu=True)
    return p


def fold_layernorm(weight, bias, gamma, beta):
    """Linear(LayerNorm(x)) as a linear on x plus per-row statistics (MOCA_EP_LNFOLD):
    W' = W * diag(gamma), b' = b + W @ beta.  Returns (W' f32, b' f32); pack them with pack_linear / pack_geglu and call
    finish_lnfold() on the result."""
    w = weight.reshape(weight.shape[0], -1).float()
    g, be = gamma.to(w.device).float(), beta.to(w.device).float()
    b = w @ be
    if bias is not None:
        b = b + bias.to(w.device).float()
    return w * g[None, :], b


def finish_lnfold(pw: PackedWeight) -> PackedWeight:
    """wsum[n] = sum_k of the PACKED fp16 W'[n][k] (what the MFMAs multiply with), so that acc - mean * wsum is exact algebra"""
    pw.wsum = pw.w.float().sum(dim=1).contiguous()
    return pw


# --------------------------------------------------------------------------------------
# kernel wrappers
# --------------------------------------------------------------------------------------
def _gemm_params(a, pw: PackedWeight, out, *, M, lda=None, mode=_l.MOCA_A_LINEAR, rowadd=None, rowadd_div=1,
                 residual=None, conv=None, tconv=None, out_f32=False, splits=1, splitk_ws=None, gelu=False, colsum=None, ln=None,
                 force_small=False, rowsum=None, lnfold=None, gstat=None, tattn=None, up_phase=0, prefetch=None, a2=None, slabs=False,
                 wgroup=None):
    p = _l.GemmParams()
    if wgroup is not None:                   # (rows per group, fp16 elements between two groups' matrices): `pw` = the per-group weights of
        p.wgroup_rows, p.wgroup_stride = wgroup   # groupnorm_fold_weights (pw.w [n_sg * N][ldw], pw.bias [n_sg * N])
    p.up_phase = up_phase
    if a2 is not None:                       # (second A source fp16 [M][lda2], columns of `a`): A = the virtual cat([a, a2], channels)
        p.a2, p.lda2, p.k1 = a2[0].data_ptr(), a2[0].stride(-2), a2[1]
    if prefetch is not None:                 # tensor the launch's spare blocks stream into the memory-side cache (the NEXT heavy launch's weights)
        p.prefetch, p.prefetch_kib = prefetch.data_ptr(), (prefetch.numel() * prefetch.element_size()) >> 10
    p.a, p.w, p.out = a.data_ptr(), pw.w.data_ptr(), (out.data_ptr() if out is not None else None)
    p.bias = pw.bias.data_ptr() if pw.bias is not None else None
    p.rowadd = rowadd.data_ptr() if rowadd is not None else None
    p.residual = residual.data_ptr() if residual is not None else None
    p.splitk_ws = splitk_ws.data_ptr() if splitk_ws is not None else None
    p.M, p.N, p.K = M, pw.N, pw.K
    p.lda = lda if lda is not None else (a.stride(-2) if a.dim() >= 2 else pw.K)
    p.ldw = pw.w.stride(0)
    p.ldo = out.stride(-2) if out is not None else pw.N
    p.ldr = residual.stride(-2) if residual is not None else 0
    p.ld_rowadd = rowadd.stride(-2) if rowadd is not None else 0
    p.rowadd_div = rowadd_div
    p.a_mode = mode
    if conv is not None:
        p.C, p.inH, p.inW, p.outH, p.outW, p.stride, p.up = conv[:7]
        p.nopad_lo = conv[7] if len(conv) > 7 else 0
    if tconv is not None:
        p.C, p.T, p.HW = tconv
    p.flags = (_l.MOCA_EP_GEGLU if pw.geglu else 0) | (_l.MOCA_EP_OUT_F32 if out_f32 else 0) | \
              ((_l.MOCA_EP_GELU | _l.MOCA_FORCE_SMALL_TILE) if gelu else 0) | (_l.MOCA_EP_COLSUM if colsum is not None else 0) | \
              (_l.MOCA_FORCE_SMALL_TILE if force_small else 0)
    p.colsum = colsum.data_ptr() if colsum is not None else None
    if ln is not None:                       # (gamma f32 [N], beta f32 [N], ln_out fp16 [M][ld] or None when only probing, eps)
        p.flags |= _l.MOCA_EP_LN
        p.ln_gamma, p.ln_beta = ln[0].data_ptr(), ln[1].data_ptr()
        p.ln_out = ln[2].data_ptr() if ln[2] is not None else None
        p.ld_ln = ln[2].stride(-2) if ln[2] is not None else pw.N
        p.ln_eps = ln[3]
    if rowsum is not None:                   # f32 [N / cols][M][2] (cols = gemm_rowsum_cols), or True when only probing
        p.flags |= _l.MOCA_EP_ROWSUM
        p.rowsum = rowsum.data_ptr() if torch.is_tensor(rowsum) else None
    if tattn is not None:                    # (T, HW, softmax scale): projection + temporal attention in one launch (MOCA_EP_TATTN)
        p.flags |= _l.MOCA_EP_TATTN
        p.T, p.HW, p.tattn_scale = tattn
    if gstat is not None:                    # (i64 [M / rows][32][2] fixed-point accumulators, zeroed before the launch; rows per statistics group
        p.flags |= _l.MOCA_EP_GSTAT         #  [, columns per channel group, channel of column 0]: one source of a virtual concat)
        p.gstat = gstat[0].data_ptr()
        p.gstat_rows = gstat[1]
        if len(gstat) > 2:
            p.gstat_cpg, p.gstat_coff = gstat[2], gstat[3]
    if lnfold is not None:                   # (row partials f32 [nparts][M][2] or None when only probing, nparts, eps); pw.wsum required
        p.flags |= _l.MOCA_EP_LNFOLD
        p.lnf_part = lnfold[0].data_ptr() if lnfold[0] is not None else None
        p.lnf_nparts = lnfold[1]
        p.ln_eps = lnfold[2]
        p.lnf_wsum = pw.wsum.data_ptr() if pw.wsum is not None else None
    if slabs:                                # split-K without the reduce launch: gemm_splitk_groupnorm() finishes the slabs
        p.flags |= _l.MOCA_EP_SLABS
    p.splits = splits
    return p


def gemm(a, pw: PackedWeight, out, **kw):
    """out[M][:] = epilogue(gather(a) @ pw.w^T).  conv = (C, inH, inW, outH, outW, stride, up);
    tconv = (C, T, HW); colsum = f32 [ceil(M/320)][N][2] buffer for the consumer GroupNorm's statistics
    (only where gemm_colsum_rows(...) > 0)."""
    p = _gemm_params(a, pw, out, **kw)
    _l.check(_l.load().moca_gemm_f16(C.byref(p), _st()), "moca_gemm_f16")
    return out


def gemm_ln_ok(a, pw: PackedWeight, **kw):
    """can this call also write the LayerNorm of its output rows (MOCA_EP_LN)?"""
    p = _gemm_params(a, pw, None, **kw)
    return bool(_l.load().moca_gemm_ln_ok(C.byref(p)))


def gemm_rowsum_cols(a, pw: PackedWeight, **kw):
    """columns per column tile of the MOCA_EP_ROWSUM output of this call (N / cols partial sums per row), 0 if it cannot"""
    p = _gemm_params(a, pw, None, **kw)
    return int(_l.load().moca_gemm_rowsum_cols(C.byref(p)))


def gemm_lnfold_ok(a, pw: PackedWeight, **kw):
    """does the kernel this call runs on have the MOCA_EP_LNFOLD epilogue?"""
    p = _gemm_params(a, pw, None, **kw)
    return bool(_l.load().moca_gemm_lnfold_ok(C.byref(p)))


def gemm_wgroup_ok(a, pw: PackedWeight, **kw):
    """can this call take per-row-group weights (`wgroup`: a GroupNorm folded into the linear that consumes it)?"""
    p = _gemm_params(a, pw, None, **kw)
    return bool(_l.load().moca_gemm_wgroup_ok(C.byref(p)))


def groupnorm_fold_weights(pw: PackedWeight, gamma, beta, gstat, wg, bg, *, n_sg, count, eps):
    """wg fp16 [n_sg * N][ldw], bg f32 [n_sg * N] = the weights / bias of `Linear(GroupNorm(x))` per statistics group, from the finished
    statistics of a MOCA_EP_GSTAT producer (include/moca_hip.h); returns them as a PackedWeight for `gemm(..., wgroup=(rows, N * ldw))`"""
    _l.check(_l.load().moca_groupnorm_fold_weights_f16(_l.ptr(pw.w), _l.ptr(pw.bias) if pw.bias is not None else None, _l.ptr(gamma), _l.ptr(beta),
                                                       _l.ptr(gstat), _l.ptr(wg), _l.ptr(bg), n_sg, pw.N, pw.K, pw.w.stride(0), count, eps, _st()),
             "moca_groupnorm_fold_weights_f16")
    out = PackedWeight(wg, bg, pw.N, pw.K, pw.n_out)
    out.derived = True
    return out


def gemm_tattn_ok(a, pw: PackedWeight, **kw):
    """can this call run as MOCA_EP_TATTN (fused q|k|v projection + temporal attention)?"""
    p = _gemm_params(a, pw, None, **kw)
    return bool(_l.load().moca_gemm_tattn_ok(C.byref(p)))


def pack_qkv_per_head(wq, wk, wv, heads, bias=None, device="cuda"):
    """to_q | to_k | to_v of a self-attention re-ordered per head for MOCA_EP_TATTN: rows [64 q_h, 64 k_h, 64 v_h] for h = 0..heads-1
    (weights [heads*64][K] each, or already LayerNorm-folded; `bias` = the folded bias in the plain q|k|v order or None)"""
    C_ = wq.shape[0]
    assert C_ == heads * 64 and wk.shape[0] == C_ and wv.shape[0] == C_
    w = torch.stack([wq.reshape(heads, 64, -1), wk.reshape(heads, 64, -1), wv.reshape(heads, 64, -1)], dim=1).reshape(3 * C_, -1)
    b = None
    if bias is not None:
        b = torch.stack([bias[:C_].reshape(heads, 64), bias[C_:2 * C_].reshape(heads, 64), bias[2 * C_:].reshape(heads, 64)], dim=1).reshape(-1)
    return _finish(w, b, device)


def gemm_splitk_groupnorm_ok(a, pw: PackedWeight, *, HW, frames_per_stat, **kw):
    """can the split-K reduce of this call be fused into the GroupNorm that consumes it (gemm_splitk_groupnorm)?"""
    p = _gemm_params(a, pw, None, slabs=True, **kw)
    return bool(_l.load().moca_gemm_splitk_groupnorm_ok(C.byref(p), HW, frames_per_stat))


def gemm_splitk_groupnorm(a, pw: PackedWeight, out, y, gamma, beta, *, HW, frames_per_stat, eps, silu, write_x, **kw):
    """finish a `gemm(..., slabs=True)` call: x = its output (written to `out` only with write_x), y = GroupNorm(+SiLU)(x); `kw` = the
    keywords of that gemm call"""
    kw = {k: v for k, v in kw.items() if k not in ("slabs", "colsum", "prefetch")}
    p = _gemm_params(a, pw, out, slabs=True, **kw)
    _l.check(_l.load().moca_gemm_splitk_groupnorm_f16(C.byref(p), _l.ptr(y), _l.ptr(gamma), _l.ptr(beta), HW, frames_per_stat, eps,
                                                      1 if silu else 0, 1 if write_x else 0, _st()), "moca_gemm_splitk_groupnorm_f16")
    return y


def gemm_cat_ok(a, pw: PackedWeight, **kw):
    """can this linear read its A operand from two sources (a2 = (second source, columns of the first): the virtual torch.cat)?"""
    p = _gemm_params(a, pw, None, **kw)
    return bool(_l.load().moca_gemm_cat_ok(C.byref(p)))


def gemm_colsum_rows(a, pw: PackedWeight, **kw):
    """rows per row tile of the MOCA_EP_COLSUM output of this call, 0 if it cannot produce column sums"""
    p = _gemm_params(a, pw, None, **kw)
    return int(_l.load().moca_gemm_colsum_rows(C.byref(p)))


def groupnorm(x, y, gamma, beta, *, F, HW, Cn, frames_per_stat, eps, silu, ws):
    _l.check(_l.load().moca_groupnorm_nhwc_f16(_l.ptr(x), _l.ptr(y), _l.ptr(gamma), _l.ptr(beta), F, HW, Cn,
                                               frames_per_stat, eps, 1 if silu else 0, _l.ptr(ws), _st()),
             "moca_groupnorm_nhwc_f16")
    return y


def groupnorm_colsum(x, y, gamma, beta, colsum, *, tile_rows, F, HW, Cn, frames_per_stat, eps, silu, ws):
    _l.check(_l.load().moca_groupnorm_colsum_f16(_l.ptr(x), _l.ptr(y), _l.ptr(gamma), _l.ptr(beta), _l.ptr(colsum), tile_rows, F, HW, Cn,
                                                 frames_per_stat, eps, 1 if silu else 0, _l.ptr(ws), _st()),
             "moca_groupnorm_colsum_f16")
    return y


def groupnorm_gstat(x, y, gamma, beta, gstat, *, F, HW, Cn, frames_per_stat, eps, silu):
    """GroupNorm whose statistics the producing GEMM accumulated (MOCA_EP_GSTAT): one launch"""
    _l.check(_l.load().moca_groupnorm_gstat_f16(_l.ptr(x), _l.ptr(y), _l.ptr(gamma), _l.ptr(beta), _l.ptr(gstat), F, HW, Cn,
                                                frames_per_stat, eps, 1 if silu else 0, _st()), "moca_groupnorm_gstat_f16")
    return y


def concat_channels_gstat(a, b, out, gstat, *, F, HW, C1, C2, frames_per_stat):
    """torch.cat(dim=channels) + the GroupNorm statistics of the result added to gstat (i64 [F / frames_per_stat][32][2])"""
    _l.check(_l.load().moca_concat_channels_gstat_f16(_l.ptr(a), _l.ptr(b), _l.ptr(out), F, HW, C1, C2, frames_per_stat,
                                                      _l.ptr(gstat), _st()), "moca_concat_channels_gstat_f16")
    return out


def groupnorm_gstat_cat(a, b, y, gamma, beta, gstat_cat, gstat_b, *, F, HW, C1, C2, frames_per_stat, eps, silu, Fb=0):
    """GroupNorm(+SiLU) of the virtual cat([a, b], channels): statistics from gstat_cat (the concat's grouping) [+ b's own gstat_b,
    which covers Fb frames when b is F / Fb copies of them]"""
    _l.check(_l.load().moca_groupnorm_gstat_cat_f16(_l.ptr(a), _l.ptr(b), _l.ptr(y), _l.ptr(gamma), _l.ptr(beta), _l.ptr(gstat_cat),
                                                    _l.ptr(gstat_b) if gstat_b is not None else None, Fb, F, HW, C1, C2, frames_per_stat,
                                                    eps, 1 if silu else 0, _st()), "moca_groupnorm_gstat_cat_f16")
    return y


def gstat_accum(x, gstat, *, F, HW, Cn, frames_per_stat, cpg, coff):
    """statistics only: x's share of a virtual concat's GroupNorm statistics (channel c -> group (coff + c) / cpg)"""
    _l.check(_l.load().moca_gstat_accum_f16(_l.ptr(x), F, HW, Cn, frames_per_stat, cpg, coff, _l.ptr(gstat), _st()),
             "moca_gstat_accum_f16")
    return gstat


def memset_zero(t):
    _l.check(_l.load().moca_memset_zero(_l.ptr(t), t.numel() * t.element_size(), _st()), "moca_memset_zero")
    return t


def groupnorm_ws_floats(F, HW, Cn):
    return int(_l.load().moca_groupnorm_ws_bytes(F, HW, Cn)) // 4


def layernorm(x, y, gamma, beta, *, M, Cn, eps=1e-5):
    _l.check(_l.load().moca_layernorm_f16(_l.ptr(x), _l.ptr(y), _l.ptr(gamma), _l.ptr(beta), M, Cn, eps, _st()),
             "moca_layernorm_f16")
    return y


def attention(q, k, v, out, *, Bq, heads, Nq, Nk, ldq, ldk, ldv, ldo, kv_div, scale):
    _l.check(_l.load().moca_attention_f16(_l.ptr(q), _l.ptr(k), _l.ptr(v), _l.ptr(out), Bq, heads, Nq, Nk,
                                          ldq, ldk, ldv, ldo, kv_div, scale, _st()), "moca_attention_f16")
    return out


def temporal_attention(q, k, v, out, *, B, T, HW, heads, ld_qkv, ldo, scale):
    _l.check(_l.load().moca_temporal_attention_f16(_l.ptr(q), _l.ptr(k), _l.ptr(v), _l.ptr(out), B, T, HW, heads,
                                                   ld_qkv, ldo, scale, _st()), "moca_temporal_attention_f16")
    return out


def ncthw_to_nhwc(x, y, *, B, Cin, T, HW, Cpad):
    _l.check(_l.load().moca_ncthw_to_nhwc_f16(_l.ptr(x), 1 if x.dtype == torch.float32 else 0, _l.ptr(y), B, Cin, T, HW,
                                              Cpad, _st()), "moca_ncthw_to_nhwc_f16")
    return y


def nhwc_to_ncthw(y, ld, x, *, B, Cout, T, HW):
    _l.check(_l.load().moca_nhwc_to_ncthw(_l.ptr(y), ld, _l.ptr(x), 1 if x.dtype == torch.float32 else 0, B, Cout, T, HW,
                                          _st()), "moca_nhwc_to_ncthw")
    return x


def concat_channels(a, b, out, *, rows, C1, C2):
    _l.check(_l.load().moca_concat_channels_f16(_l.ptr(a), _l.ptr(b), _l.ptr(out), rows, C1, C2, _st()),
             "moca_concat_channels_f16")
    return out


def repeat(src, dst, *, reps):
    """dst = reps back-to-back copies of src"""
    nb = src.numel() * src.element_size()
    assert dst.numel() * dst.element_size() == nb * reps
    _l.check(_l.load().moca_repeat_f16(_l.ptr(src), _l.ptr(dst), nb, reps, _st()), "moca_repeat_f16")
    return dst


def timestep_embedding(t, out, *, n, dim, max_period=10000.0):
    _l.check(_l.load().moca_timestep_embedding_f16(_l.ptr(t), _l.ptr(out), n, dim, max_period, _st()),
             "moca_timestep_embedding_f16")
    return out


def silu_add_rows(a, div_a, b, div_b, out, *, rows, Cn, silu):
    _l.check(_l.load().moca_silu_add_rows_f16(_l.ptr(a), div_a, _l.ptr(b), div_b, _l.ptr(out), rows, Cn,
                                              1 if silu else 0, _st()), "moca_silu_add_rows_f16")
    return out


def channel_mix(z, w, bias, out, *, B, Cin, T, HW, Cout, Cpad, inv_scale):
    _l.check(_l.load().moca_channel_mix_f16(_l.ptr(z), 1 if z.dtype == torch.float32 else 0, _l.ptr(w), _l.ptr(bias), _l.ptr(out),
                                            B, Cin, T, HW, Cout, Cpad, inv_scale, _st()), "moca_channel_mix_f16")
    return out


def softmax_rows(s, p, *, R, N, scale):
    _l.check(_l.load().moca_softmax_rows_f16(_l.ptr(s), _l.ptr(p), R, N, s.stride(-2), p.stride(-2), scale, _st()),
             "moca_softmax_rows_f16")
    return p


def gaussian_sample(moments, noise, out, *, n, z, hw, scale):
    """out = scale * (mean + exp(0.5 * clamp(logvar, -30, 20)) * noise); noise None -> the mode"""
    _l.check(_l.load().moca_gaussian_sample_f32(_l.ptr(moments), _l.ptr(noise), _l.ptr(out), n, z, hw, scale, _st()),
             "moca_gaussian_sample_f32")
    return out


def attention_causal(q, k, v, out, *, B, heads, N, ldq, ldk, ldv, ldo, scale):
    _l.check(_l.load().moca_attention_causal_f16(_l.ptr(q), _l.ptr(k), _l.ptr(v), _l.ptr(out), B, heads, N, ldq, ldk, ldv, ldo,
                                                 scale, _st()), "moca_attention_causal_f16")
    return out


def embed_tokens(tokens, table, pos, out, *, n_tokens, L, Cn, vocab):
    _l.check(_l.load().moca_embed_tokens_f16(_l.ptr(tokens), _l.ptr(table), _l.ptr(pos), _l.ptr(out), n_tokens, L, Cn, vocab, _st()),
             "moca_embed_tokens_f16")
    return out
