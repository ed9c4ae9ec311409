"""Kernel-level parity: every C-ABI kernel of libmoca_hip.so against a plain PyTorch fp32
restatement of the same op, computed on the same (fp16-rounded) inputs.  These run on the
GPU box only (`-m gpu`).  Tolerances: fp16 outputs -> max|err| <= 3e-3 * max|ref| (one fp16
rounding is 4.9e-4 relative; K up to ~3000 fp32-accumulated products); fp32 sampler /
FreeInit kernels -> 1e-5 relative."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from moca_video_amd import lib as L  # noqa: E402
from moca_video_amd import ops  # noqa: E402

DEV = "cuda"
TOL16 = 3e-3


def relerr(got, ref):
    got, ref = got.float(), ref.float()
    return ((got - ref).abs().max() / ref.abs().max().clamp_min(1e-12)).item()


def check(got, ref, tol, what):
    assert torch.isfinite(got.float()).all(), f"{what}: non-finite output"
    e = relerr(got, ref)
    assert e <= tol, f"{what}: rel max err {e:.3e} > {tol:.1e}"


@pytest.fixture(autouse=True)
def _stream():
    ops.set_stream(None)
    yield
    torch.cuda.synchronize()


def rnd(*shape, scale=1.0, dtype=torch.float16, seed=[0]):
    seed[0] += 1
    g = torch.Generator(device="cpu").manual_seed(1234 + seed[0])
    return (torch.randn(*shape, generator=g) * scale).to(DEV).to(dtype)


# ---------------------------------------------------------------- GEMM: linear
@pytest.mark.parametrize("M,N,K,splits", [(300, 192, 328, 1), (1000, 256, 2880, 1), (128, 64, 64, 1),
                                         (640, 1280, 2304, 1), (640, 1280, 2304, 4), (77, 640, 1024, 3),
                                         (16, 1280, 320, 1)])
def test_gemm_linear(M, N, K, splits):
    a = rnd(M, K)
    w = rnd(N, K, scale=K ** -0.5)
    b = rnd(N, dtype=torch.float32)
    res = rnd(M, N)
    pw = ops.pack_linear(w, b)
    out = torch.empty(M, pw.N, dtype=torch.float16, device=DEV)
    ws = torch.empty(splits * M * pw.N, dtype=torch.float32, device=DEV) if splits > 1 else None
    ops.gemm(a, pw, out, M=M, residual=res if pw.N == N else None, splits=splits, splitk_ws=ws)
    ref = a.float() @ w.float().t() + b
    if pw.N == N:
        ref = ref + res.float()
    check(out[:, :N], ref, TOL16, f"gemm linear {M}x{N}x{K} s{splits}")


def test_gemm_rowadd_f32out():
    M, N, K, div = 600, 128, 256, 100
    a, w = rnd(M, K), rnd(N, K, scale=K ** -0.5)
    ra = rnd(M // div, N)
    pw = ops.pack_linear(w, None)
    out = torch.empty(M, N, dtype=torch.float32, device=DEV)
    ops.gemm(a, pw, out, M=M, rowadd=ra, rowadd_div=div, out_f32=True)
    ref = a.float() @ w.float().t() + ra.float().repeat_interleave(div, dim=0)
    check(out, ref, 1e-3, "gemm rowadd f32")


@pytest.mark.parametrize("M,res,rows,bias,pad", [(8192, True, True, True, 0), (8192, False, False, False, 0), (10016, True, False, True, 64),
                                               (81920, True, True, True, 0), (81920, False, True, True, 32), (40960, "inplace", True, True, 0)])
def test_gemm_weight_stationary_320(M, res, rows, bias, pad):
    """gemm_ws.hip: the weight-stationary streaming kernel of the 320 -> 320 linears (W as MFMA fragments in registers, A and the residual
    through an LDS-DMA ring, register epilogue) against fp32 torch AND against the tiled kernel it replaces (knob off): one strip per
    block, uneven strips per block (313 strips on 256 blocks), 10 strips per block; bias / residual / row sums; padded row strides;
    `out` aliasing `residual` (attention.py:217-219: x = attn(norm(x)) + x in place)."""
    K = N = 320
    a_full, w = rnd(M, K + pad), rnd(N, K, scale=K ** -0.5)
    a = a_full[:, :K]
    b = rnd(N, dtype=torch.float32) if bias else None
    pw = ops.pack_linear(w, b)
    r_full = rnd(M, N + pad) if res else None
    r = r_full[:, :N] if res else None
    ref = a.float() @ w.float().t() + (b if bias else 0.0) + (r.float() if res else 0.0)
    old = L.set_tuning(L.MOCA_TUNE_GEMM_WS, 2)
    assert ops.gemm_rowsum_cols(a, pw, M=M, residual=r, rowsum=True) == 80, "the weight-stationary kernel leaves one row partial per column group"
    L.set_tuning(L.MOCA_TUNE_GEMM_WS, old)
    outs = []
    for knob in (2, 0):                      # weight-stationary kernel wherever it applies; tiled kernel
        old = L.set_tuning(L.MOCA_TUNE_GEMM_WS, knob)
        try:
            cols = ops.gemm_rowsum_cols(a, pw, M=M, residual=r, rowsum=True)
            o_full = torch.full((M, N + pad), float("nan"), dtype=torch.float16, device=DEV)
            out = o_full[:, :N]
            rr = r
            if res == "inplace":
                out = r.clone()
                rr = out
            part = torch.full((N // cols * M, 2), float("nan"), dtype=torch.float32, device=DEV) if rows else None
            ops.gemm(a, pw, out, M=M, residual=rr, rowsum=part)
            check(out, ref, TOL16, f"ws={knob} linear {M}x320x320 res={res} rows={rows}")
            if pad:
                assert torch.isnan(o_full[:, N:]).all(), "wrote outside the output columns"
            if rows:
                ps = part.view(N // cols, M, 2).sum(0)
                of = out.float()
                assert relerr(ps[:, 0], of.sum(1)) < 1e-4 and relerr(ps[:, 1], (of * of).sum(1)) < 1e-4, "row sums of the stored values"
            outs.append(out.clone())
        finally:
            L.set_tuning(L.MOCA_TUNE_GEMM_WS, old)
    assert relerr(outs[0], outs[1]) < 1e-3, "weight-stationary vs tiled kernel"
    old = L.set_tuning(L.MOCA_TUNE_GEMM_WS, 2)
    try:
        # refused shapes fall through to the tiled kernels: M not a multiple of 32
        assert ops.gemm_rowsum_cols(a[:M - 8], pw, M=M - 8, rowsum=True) != 80
        # repeatable to the bit (no atomics, fixed strip -> block map)
        o2 = torch.empty(M, N, dtype=torch.float16, device=DEV)
        ops.gemm(a, pw, o2, M=M, residual=None if res == "inplace" else r)
        o3 = torch.empty_like(o2)
        ops.gemm(a, pw, o3, M=M, residual=None if res == "inplace" else r)
        assert torch.equal(o2, o3)
    finally:
        L.set_tuning(L.MOCA_TUNE_GEMM_WS, old)


@pytest.mark.parametrize("Fr,HW,C,fps,mode", [(32, 2560, 320, 1, "rowsum"), (32, 2560, 320, 16, "ln"), (32, 640, 640, 1, "rowsum"),
                                              (32, 640, 640, 16, "plain")])
def test_gemm_groupnorm_folded_into_per_group_weights(Fr, HW, C, fps, mode):
    """`proj_in(GroupNorm(x))` of SpatialTransformer / TemporalTransformer (attention.py:238-242,262-268 / :297-302,333-341) as ONE GEMM on
    per-statistics-group weights (moca_groupnorm_fold_weights_f16 + moca_gemm_params.wgroup_rows): against fp32 torch and against the
    two-launch path (statistics-fed GroupNorm apply, then the linear); per-frame and per-video statistics; with the row sums / the
    LayerNorm store loop the consumer of proj_in asks for; poisoned statistics give NaN rows for exactly that group."""
    M, N, eps = Fr * HW, C, 1e-6
    x = (rnd(M, C) * 1.7 + 0.6).half()
    x.view(Fr, HW, C)[1] *= 3.0                                  # (a frame of another scale: the groups must not mix)
    gm, be = rnd(C, dtype=torch.float32) * 0.3 + 1.0, rnd(C, dtype=torch.float32) * 0.3
    w, b = rnd(N, C, scale=C ** -0.5), rnd(N, dtype=torch.float32)
    pw = ops.pack_linear(w, b)
    n_sg = Fr // fps
    gst = torch.zeros(n_sg * 64, dtype=torch.int64, device=DEV)
    ops.gstat_accum(x, gst, F=Fr, HW=HW, Cn=C, frames_per_stat=fps, cpg=C // 32, coff=0)
    ldw = pw.w.stride(0)
    wg = torch.empty(n_sg * pw.N, ldw, dtype=torch.float16, device=DEV)
    bg = torch.empty(n_sg * pw.N, dtype=torch.float32, device=DEV)
    pwg = ops.groupnorm_fold_weights(pw, gm, be, gst, wg, bg, n_sg=n_sg, count=fps * HW * (C // 32), eps=eps)
    wgroup = (fps * HW, pw.N * ldw)
    assert ops.gemm_wgroup_ok(x, pwg, M=M, wgroup=wgroup)
    xr = x.float().view(n_sg, fps * HW, C).permute(0, 2, 1)
    y_ref = F.group_norm(xr, 32, gm, be, eps).permute(0, 2, 1).reshape(M, C)
    ref = y_ref @ w.float().t() + b
    out = torch.empty(M, N, dtype=torch.float16, device=DEV)
    kw = {}
    if mode == "rowsum":
        cols = ops.gemm_rowsum_cols(x, pwg, M=M, rowsum=True, wgroup=wgroup)
        assert cols > 0
        part = torch.empty(N // cols * M, 2, dtype=torch.float32, device=DEV)
        kw["rowsum"] = part
    elif mode == "ln":
        lg, lb = rnd(N, dtype=torch.float32) * 0.2 + 1.0, rnd(N, dtype=torch.float32) * 0.2
        assert ops.gemm_ln_ok(x, pwg, M=M, ln=(lg, lb, None, 1e-5), wgroup=wgroup)
        lo = torch.empty(M, N, dtype=torch.float16, device=DEV)
        kw["ln"] = (lg, lb, lo, 1e-5)
    ops.gemm(x, pwg, out, M=M, wgroup=wgroup, **kw)
    check(out, ref, TOL16, "Linear(GroupNorm(x)) on per-group weights")
    if C == 320 and mode != "ln":                      # the weight-stationary kernel deals its strips by weight group: same call, knob 2
        old = L.set_tuning(L.MOCA_TUNE_GEMM_WS, 2)
        try:
            assert ops.gemm_wgroup_ok(x, pwg, M=M, wgroup=wgroup)
            kw2 = dict(kw)
            if mode == "rowsum":
                cols2 = ops.gemm_rowsum_cols(x, pwg, M=M, rowsum=True, wgroup=wgroup)
                assert cols2 == 80
                kw2["rowsum"] = torch.empty(N // cols2 * M, 2, dtype=torch.float32, device=DEV)
            outw = torch.empty_like(out)
            ops.gemm(x, pwg, outw, M=M, wgroup=wgroup, **kw2)
            check(outw, ref, TOL16, "Linear(GroupNorm(x)) on per-group weights, weight-stationary kernel")
            assert relerr(outw, out) < 1e-3
        finally:
            L.set_tuning(L.MOCA_TUNE_GEMM_WS, old)
    if mode == "rowsum":
        ps, of = part.view(N // cols, M, 2).sum(0), out.float()
        assert relerr(ps[:, 0], of.sum(1)) < 1e-4 and relerr(ps[:, 1], (of * of).sum(1)) < 1e-4
    if mode == "ln":
        check(lo, F.layer_norm(out.float(), (N,), lg, lb, 1e-5), TOL16, "LayerNorm store loop behind the folded GroupNorm")
    # the two-launch path it replaces
    y = torch.empty_like(x)
    ops.groupnorm_gstat(x, y, gm, be, gst, F=Fr, HW=HW, Cn=C, frames_per_stat=fps, eps=eps, silu=False)
    out2 = torch.empty_like(out)
    ops.gemm(y, pw, out2, M=M)
    assert relerr(out, out2) < 2e-3
    e_fold, e_two = relerr(out, ref), relerr(out2, ref)
    print(f"[parity] folded {e_fold:.2e} vs two launches {e_two:.2e}")
    # refusals: a group that does not hold whole row tiles; the weight-stationary kernel never takes such a call
    assert not ops.gemm_wgroup_ok(x, pwg, M=M, wgroup=(fps * HW - 160 + 32, pw.N * ldw))
    # poison: group 1's statistics out of range -> its rows NaN, the others untouched
    gst2 = gst.clone()
    gst2.view(n_sg, 32, 2)[1, 3, 1] = 1 << 62
    ops.groupnorm_fold_weights(pw, gm, be, gst2, wg, bg, n_sg=n_sg, count=fps * HW * (C // 32), eps=eps)
    out3 = torch.empty_like(out)
    ops.gemm(x, pwg, out3, M=M, wgroup=wgroup)
    o3 = out3.view(n_sg, fps * HW, N)
    assert torch.isnan(o3[1]).all() and torch.equal(o3[0], out.view(n_sg, fps * HW, N)[0]) and (n_sg < 3 or torch.equal(o3[2], out.view(n_sg, fps * HW, N)[2]))


@pytest.mark.parametrize("Fr,HW,fps,res,cpg,coff", [(32, 2560, 1, True, 0, 0), (32, 2560, 16, True, 0, 0), (16, 1280, 1, False, 0, 0),
                                                     (32, 2560, 1, True, 20, 320), (32, 2560, 1, True, 30, 0), (256, 2560, 1, True, 0, 0)])
def test_gemm_weight_stationary_320_groupnorm_statistics(Fr, HW, fps, res, cpg, coff):
    """MOCA_EP_GSTAT on the weight-stationary kernel (proj_out + the GroupNorm that follows, openaimodel3d.py:149; also as ONE SOURCE of a
    virtual concat: gstat_cpg / gstat_coff).  The kernel deals its strips BY STATISTICS GROUP (several blocks per group, or whole groups per
    block: 256 frames on 256 CUs), keeps the column sums in registers over a group's strips and flushes them once: the finished
    fixed-point statistics against fp32 sums of what the launch stored (before the fp16 rounding, as the tiled kernels count them), the
    GroupNorm fed by them against torch, and the tiled kernel's accumulators."""
    M, K, N = Fr * HW, 320, 320
    a, w, b = rnd(M, K), rnd(N, K, scale=K ** -0.5), rnd(N, dtype=torch.float32)
    pw = ops.pack_linear(w, b)
    r = rnd(M, N) if res else None
    ref = a.float() @ w.float().t() + b + (r.float() if res else 0.0)
    n_sg = Fr // fps
    accs, outs = [], []
    for knob in (2, 0):
        old = L.set_tuning(L.MOCA_TUNE_GEMM_WS, knob)
        try:
            gst = torch.zeros(n_sg * 64, dtype=torch.int64, device=DEV)
            out = torch.empty(M, N, dtype=torch.float16, device=DEV)
            g = (gst, fps * HW) if cpg == 0 else (gst, fps * HW, cpg, coff)
            ops.gemm(a, pw, out, M=M, residual=r, gstat=g)
            check(out, ref, TOL16, f"ws={knob} linear +gstat")
            accs.append(gst.clone()); outs.append(out.clone())
        finally:
            L.set_tuning(L.MOCA_TUNE_GEMM_WS, old)
    assert relerr(outs[0], outs[1]) < 1e-3
    sc = torch.tensor([2.0 ** -20, 2.0 ** -12], dtype=torch.float64, device=DEV)
    got = accs[0].view(n_sg, 32, 2).double() * sc
    cw = cpg if cpg else N // 32
    groups = (coff + torch.arange(N, device=DEV)) // cw                       # consumer channel group of every column
    xg = ref.view(n_sg, fps * HW, N).double()
    want = torch.zeros(n_sg, 32, 2, dtype=torch.float64, device=DEV)
    want[:, :, 0].index_add_(1, groups, xg.sum(1))
    want[:, :, 1].index_add_(1, groups, (xg * xg).sum(1))
    assert relerr(got[..., 0], want[..., 0]) < 1e-3 and relerr(got[..., 1], want[..., 1]) < 1e-3, "fixed-point statistics vs fp32 sums"
    tiled = accs[1].view(n_sg, 32, 2).double() * sc
    assert relerr(got, tiled) < 1e-4, "weight-stationary vs tiled accumulators"
    if cpg == 0:                                                              # the GroupNorm that consumes them
        gm, be = rnd(N, dtype=torch.float32) * 0.2 + 1.0, rnd(N, dtype=torch.float32) * 0.2
        y = torch.empty_like(outs[0])
        ops.groupnorm_gstat(outs[0], y, gm, be, accs[0], F=Fr, HW=HW, Cn=N, frames_per_stat=fps, eps=1e-5, silu=True)
        xr = outs[0].float().view(n_sg, fps * HW, N).permute(0, 2, 1)
        gref = F.silu(F.group_norm(xr, 32, gm, be, 1e-5)).permute(0, 2, 1).reshape(M, N)
        check(y, gref, TOL16, "GroupNorm from the weight-stationary kernel's statistics")


@pytest.mark.parametrize("splits", [1, 2])
def test_gemm_geglu(splits):
    M, K, inner = 520, 320, 1280
    a = rnd(M, K)
    w = rnd(2 * inner, K, scale=K ** -0.5)
    b = rnd(2 * inner, dtype=torch.float32, scale=0.1)
    pw = ops.pack_geglu(w, b)
    out = torch.empty(M, inner, dtype=torch.float16, device=DEV)
    ws = torch.empty(splits * M * pw.N, dtype=torch.float32, device=DEV) if splits > 1 else None
    ops.gemm(a, pw, out, M=M, splits=splits, splitk_ws=ws)
    y = a.float() @ w.float().t() + b
    ref = y[:, :inner] * F.gelu(y[:, inner:])
    check(out, ref, TOL16, "gemm geglu")


# ---------------------------------------------------------------- GEMM: conv
def nhwc(x):   # [F,C,H,W] -> [F,H,W,C] contiguous
    return x.permute(0, 2, 3, 1).contiguous()


@pytest.mark.parametrize("Fr,C,H,W,N,stride,up", [(3, 64, 10, 12, 128, 1, 0), (2, 320, 8, 8, 320, 1, 0),
                                                  (2, 64, 10, 12, 64, 2, 0), (2, 64, 5, 6, 128, 1, 1),
                                                  (1, 8, 16, 16, 320, 1, 0), (16, 128, 5, 8, 128, 1, 0)])
def test_gemm_conv3x3(Fr, C, H, W, N, stride, up):
    x = rnd(Fr, C, H, W)
    w = rnd(N, C, 3, 3, scale=(9 * C) ** -0.5)
    b = rnd(N, dtype=torch.float32)
    pw = ops.pack_conv3x3(w, b)
    xin = x.float()
    if up:
        xin = F.interpolate(xin, scale_factor=2, mode="nearest")
    ref = F.conv2d(xin, w.float(), b, stride=stride, padding=1)
    oH, oW = ref.shape[-2:]
    M = Fr * oH * oW
    out = torch.empty(M, pw.N, dtype=torch.float16, device=DEV)
    ops.gemm(nhwc(x), pw, out, M=M, mode=L.MOCA_A_CONV3X3, conv=(C, H, W, oH, oW, stride, up))
    got = out.view(Fr, oH, oW, pw.N)[..., :N].permute(0, 3, 1, 2)
    check(got, ref, TOL16, f"conv3x3 s{stride} up{up}")


@pytest.mark.parametrize("B,T,HW,C,N", [(2, 5, 30, 64, 64), (1, 16, 40, 320, 320), (1, 8, 64, 128, 128)])
def test_gemm_tconv(B, T, HW, C, N):
    x = rnd(B, C, T, HW, 1)
    w = rnd(N, C, 3, 1, 1, scale=(3 * C) ** -0.5)
    b = rnd(N, dtype=torch.float32)
    pw = ops.pack_tconv3(w, b)
    ref = F.conv3d(x.float(), w.float(), b, padding=(1, 0, 0))          # [B,N,T,HW,1]
    xr = x[..., 0].permute(0, 2, 3, 1).contiguous()                      # [B,T,HW,C]
    M = B * T * HW
    out = torch.empty(M, pw.N, dtype=torch.float16, device=DEV)
    res = rnd(M, pw.N)
    ops.gemm(xr, pw, out, M=M, mode=L.MOCA_A_TCONV3, tconv=(C, T, HW), residual=res)
    got = out.view(B, T, HW, pw.N).permute(0, 3, 1, 2)
    check(got, ref[..., 0] + res.float().view(B, T, HW, pw.N).permute(0, 3, 1, 2), TOL16, "tconv3")


@pytest.fixture
def tune():
    """kernel-choice knobs (include/moca_hip.h MOCA_TUNE_*: which kernel runs a shape, never what it computes), restored afterwards"""
    saved = []

    def set_(knob, value):
        saved.append((knob, L.set_tuning(knob, value)))
    yield set_
    for knob, old in reversed(saved):
        L.set_tuning(knob, old)


# ---------------------------------------------------------------- norms
@pytest.mark.parametrize("Fr,HW,C,fps,silu,eps", [(4, 100, 320, 1, True, 1e-5), (4, 100, 320, 2, True, 1e-5),
                                                   (16, 40, 1280, 16, False, 1e-6), (2, 2560, 320, 1, False, 1e-6),
                                                   (3, 37, 64, 1, True, 1e-5), (2, 64, 2560, 1, True, 1e-5),
                                                   (8, 64, 960, 8, True, 1e-5)])
@pytest.mark.parametrize("path", [0, 2])     # MOCA_TUNE_GN_SLAB: 0 = three-launch streaming path, 2 = single-launch slab path
def test_groupnorm(Fr, HW, C, fps, silu, eps, path, tune):
    tune(L.MOCA_TUNE_GN_SLAB, path)
    x = (rnd(Fr, HW, C) * 1.5 + 0.7).half()
    g = rnd(C, dtype=torch.float32) * 0.2 + 1.0
    b = rnd(C, dtype=torch.float32) * 0.2
    y = torch.empty_like(x)
    ws = torch.empty(ops.groupnorm_ws_floats(Fr, HW, C), dtype=torch.float32, device=DEV)
    ops.groupnorm(x, y, g, b, F=Fr, HW=HW, Cn=C, frames_per_stat=fps, eps=eps, silu=silu, ws=ws)
    # reference: [B, C, fps*HW]
    xr = x.float().view(Fr // fps, fps * HW, C).permute(0, 2, 1)
    ref = F.group_norm(xr, 32, g, b, eps)
    if silu:
        ref = F.silu(ref)
    ref = ref.permute(0, 2, 1).reshape(Fr, HW, C)
    check(y, ref, TOL16, "groupnorm")


@pytest.mark.parametrize("mode,Fr,HW,C,N,fps,with_res", [("tconv", 32, 1280, 320, 320, 16, True), ("conv", 8, 2560, 64, 640, 1, False),
                                                            ("linear", 16, 1280, 320, 640, 8, True),
                                                            ("tconv", 32, 160, 1280, 1280, 16, True),      # 256-row kernel (BN 128)
                                                            ("linear", 32, 160, 1280, 1280, 16, False),
                                                            ("conv", 16, 160, 128, 320, 8, True)])         # 256-row kernel (BN 160)
def test_gemm_colsum_feeds_groupnorm(mode, Fr, HW, C, N, fps, with_res):
    """MOCA_EP_COLSUM: the 320-row GEMM leaves per-(row tile, column) sums / sums of squares of what it stores; the GroupNorm
    that consumes them (finalize-from-column-sums + apply) must equal GroupNorm of the stored tensor (ref: torch)."""
    M = Fr * HW
    b = rnd(N, dtype=torch.float32)
    res = rnd(M, N) if with_res else None
    if mode == "linear":
        a, w = rnd(M, C), rnd(N, C, scale=C ** -0.5)
        pw, kw = ops.pack_linear(w, b), {}
        ref = a.float() @ w.float().t() + b
    elif mode == "conv":
        H, W = (40, HW // 40) if HW % 40 == 0 and HW >= 1600 else (10, HW // 10)
        x = rnd(Fr, C, H, W)
        w = rnd(N, C, 3, 3, scale=(9 * C) ** -0.5)
        pw = ops.pack_conv3x3(w, b)
        a = nhwc(x).reshape(M, C)
        kw = dict(mode=L.MOCA_A_CONV3X3, conv=(C, H, W, H, W, 1, 0))
        ref = F.conv2d(x.float(), w.float(), b, padding=1).permute(0, 2, 3, 1).reshape(M, N)
    else:
        T = 16
        x = rnd(Fr // T, C, T, HW, 1)
        w = rnd(N, C, 3, 1, 1, scale=(3 * C) ** -0.5)
        pw = ops.pack_tconv3(w, b)
        a = x.permute(0, 2, 3, 4, 1).reshape(M, C).contiguous()
        kw = dict(mode=L.MOCA_A_TCONV3, tconv=(C, T, HW))
        ref = F.conv3d(x.float(), w.float(), b, padding=(1, 0, 0)).permute(0, 2, 3, 4, 1).reshape(M, N)
    if res is not None:
        ref = ref + res.float()
    rows = ops.gemm_colsum_rows(a, pw, M=M, residual=res, **kw)
    assert rows in (160, 256, 320) and M % rows == 0, "this shape is expected on the 320 x 160 / 160 x 320 / 256-row kernel"
    assert (rows == 256) == (HW == 160)
    out = torch.empty(M, N, dtype=torch.float16, device=DEV)
    cs = torch.full((M // rows, 2 * N), float("nan"), dtype=torch.float32, device=DEV)
    ops.gemm(a, pw, out, M=M, residual=res, colsum=cs, **kw)
    check(out, ref, TOL16, f"gemm+colsum {mode}")
    tiles = ref.view(M // rows, rows, N)
    cs = cs.view(M // rows, N, 2)
    assert relerr(cs[..., 0], tiles.sum(1)) < 2e-3 and relerr(cs[..., 1], (tiles * tiles).sum(1)) < 2e-3
    g = rnd(N, dtype=torch.float32) * 0.2 + 1.0
    be = rnd(N, dtype=torch.float32) * 0.2
    y = torch.empty_like(out)
    ws = torch.empty(ops.groupnorm_ws_floats(Fr, HW, N), dtype=torch.float32, device=DEV)
    ops.groupnorm_colsum(out, y, g, be, cs, tile_rows=rows, F=Fr, HW=HW, Cn=N, frames_per_stat=fps, eps=1e-5, silu=True, ws=ws)
    xr = out.float().view(Fr // fps, fps * HW, N).permute(0, 2, 1)
    gref = F.silu(F.group_norm(xr, 32, g, be, 1e-5)).permute(0, 2, 1).reshape(Fr * HW, N)
    check(y, gref, TOL16, f"groupnorm from column sums ({mode})")
    # MOCA_EP_GSTAT: the same launch accumulating the FINISHED statistics (64-bit fixed-point atomics per (statistics group,
    # channel group): order independent, so a second launch reproduces the accumulators bit for bit);
    # the GroupNorm is then a single apply launch
    n_sg = Fr // fps
    gst = torch.zeros(n_sg * 64, dtype=torch.int64, device=DEV)
    out2 = torch.empty_like(out)
    ops.gemm(a, pw, out2, M=M, residual=res, gstat=(gst, fps * HW), **kw)
    assert torch.equal(out2, out)
    xg = out.float().view(n_sg, fps * HW, 32, N // 32)
    gs = gst.view(n_sg, 32, 2).double() * torch.tensor([2.0 ** -20, 2.0 ** -12], dtype=torch.float64, device=DEV)   # fixed point
    assert relerr(gs[..., 0], xg.sum(dim=(1, 3))) < 1e-3 and relerr(gs[..., 1], (xg * xg).sum(dim=(1, 3))) < 1e-3
    gst2 = torch.zeros_like(gst)
    ops.gemm(a, pw, out2, M=M, residual=res, gstat=(gst2, fps * HW), **kw)
    assert torch.equal(gst2, gst)
    y3 = torch.full_like(y, float("nan"))
    ops.groupnorm_gstat(out, y3, g, be, gst, F=Fr, HW=HW, Cn=N, frames_per_stat=fps, eps=1e-5, silu=True)
    check(y3, gref, TOL16, f"groupnorm from accumulated statistics ({mode})")
    # a shape the 320-row kernel does not take reports 0 (the plan then keeps the three-launch GroupNorm)
    assert ops.gemm_colsum_rows(rnd(100, 64), ops.pack_linear(rnd(128, 64), None), M=100) == 0


@pytest.mark.parametrize("M,K,with_res", [(40000, 320, True), (33000, 1280, False)])
def test_gemm_layernorm_epilogue(M, K, with_res):
    """MOCA_EP_LN: the 160 x 320 tiling writes the linear's output (+residual) AND its LayerNorm in one launch
    (attention.py:199-201,216-219); M tail rows, both outputs against torch."""
    N = 320
    a, w = rnd(M, K), rnd(N, K, scale=K ** -0.5)
    b = rnd(N, dtype=torch.float32)
    res = rnd(M, N) if with_res else None
    g = rnd(N, dtype=torch.float32) * 0.2 + 1.0
    be = rnd(N, dtype=torch.float32) * 0.2
    pw = ops.pack_linear(w, b)
    assert ops.gemm_ln_ok(a, pw, M=M, residual=res, ln=(g, be, None, 1e-5))
    out = torch.empty(M, N, dtype=torch.float16, device=DEV)
    ln = torch.full((M, N), float("nan"), dtype=torch.float16, device=DEV)
    ops.gemm(a, pw, out, M=M, residual=res, ln=(g, be, ln, 1e-5))
    ref = a.float() @ w.float().t() + b
    if res is not None:
        ref = ref + res.float()
    check(out, ref, TOL16, "linear (+res) of the LN launch")
    check(ln, F.layer_norm(ref, (N,), g, be, 1e-5), TOL16, "LayerNorm epilogue")
    # not available for other widths / short M: the plan falls back to linear + layernorm kernel
    assert not ops.gemm_ln_ok(rnd(40000, 640), ops.pack_linear(rnd(640, 640), None), M=40000, ln=(g, be, None, 1e-5))
    assert not ops.gemm_ln_ok(rnd(1000, 320), pw, M=1000, ln=(g, be, None, 1e-5))


@pytest.mark.parametrize("M,C", [(500, 320), (333, 640), (200, 1280), (64, 512), (10, 2560),
                                 (16390, 320), (16385, 1280), (20001, 2560)])     # >= 16384 rows: four rows per wave (+ tails)
def test_layernorm(M, C):
    x = (rnd(M, C) * 2 + 0.5).half()
    g = rnd(C, dtype=torch.float32) * 0.2 + 1.0
    b = rnd(C, dtype=torch.float32) * 0.2
    y = torch.empty_like(x)
    ops.layernorm(x, y, g, b, M=M, Cn=C)
    check(y, F.layer_norm(x.float(), (C,), g, b, 1e-5), TOL16, "layernorm")


# ---------------------------------------------------------------- attention
def sdpa_ref(q, k, v, scale):
    s = torch.einsum("bhid,bhjd->bhij", q.float(), k.float()) * scale
    return torch.einsum("bhij,bhjd->bhid", s.softmax(-1), v.float())


@pytest.mark.parametrize("Bq,heads,Nq,Nk,kv_div", [(4, 5, 300, 300, 1), (2, 5, 2560, 2560, 1), (4, 10, 160, 77, 2),
                                                   (6, 5, 100, 154, 3), (2, 20, 40, 40, 1), (1, 1, 33, 1, 1),
                                                   (32, 5, 2560, 77, 16), (32, 10, 640, 77, 16), (3, 2, 700, 96, 1), (2, 4, 130, 65, 2)])
def test_attention(Bq, heads, Nq, Nk, kv_div):
    C = heads * 64
    Bk = Bq // kv_div
    q = rnd(Bq, Nq, C)
    kv = rnd(Bk, Nk, 2 * C)
    k, v = kv[..., :C], kv[..., C:]
    out = torch.empty(Bq, Nq, C, dtype=torch.float16, device=DEV)
    ops.attention(q, k, v, out, Bq=Bq, heads=heads, Nq=Nq, Nk=Nk, ldq=C, ldk=2 * C, ldv=2 * C, ldo=C, kv_div=kv_div,
                  scale=0.125)
    qh = q.view(Bq, Nq, heads, 64).permute(0, 2, 1, 3)
    kh = k.reshape(Bk, Nk, heads, 64).permute(0, 2, 1, 3).repeat_interleave(kv_div, dim=0)
    vh = v.reshape(Bk, Nk, heads, 64).permute(0, 2, 1, 3).repeat_interleave(kv_div, dim=0)
    ref = sdpa_ref(qh, kh, vh, 0.125).permute(0, 2, 1, 3).reshape(Bq, Nq, C)
    check(out, ref, TOL16, "attention")


def test_attention_peaky():
    """online-softmax rescale path: a late key dominates an early maximum."""
    Bq, heads, Nq, Nk = 1, 1, 64, 200
    q = rnd(Bq, Nq, 64)
    k = rnd(Bq, Nk, 64)
    v = rnd(Bq, Nk, 64)
    k[0, 150] = q[0, 3] * 4.0
    k[0, 10] = q[0, 3] * 2.0
    out = torch.empty(Bq, Nq, 64, dtype=torch.float16, device=DEV)
    ops.attention(q, k, v, out, Bq=1, heads=1, Nq=Nq, Nk=Nk, ldq=64, ldk=64, ldv=64, ldo=64, kv_div=1, scale=0.125)
    ref = sdpa_ref(q[:, None], k[:, None], v[:, None], 0.125)[:, 0]
    check(out, ref, TOL16, "attention peaky")


@pytest.mark.parametrize("regime", ["low", "high", "late_peak", "mixed", "band_edge"])
def test_attention_reference_regimes(regime):
    """the long-key kernel keeps its softmax reference at 0 while every first-tile maximum of a wave lies inside [-6, 6] (log2 units)
    and takes the score MFMAs' accumulator input from the inline constant; outside the band, or once a later tile exceeds the
    reference by more than 8, it moves the reference as before.  Rows whose scores are all strongly negative / positive, a late
    dominant key, both kinds of row in one wavefront, and first-tile maxima right at the band's edge."""
    Bq, heads, Nq, Nk = 2, 2, 256, 512
    q, k, v = rnd(Bq, Nq, heads * 64), rnd(Bq, Nk, heads * 64), rnd(Bq, Nk, heads * 64)
    scale = 0.125
    l2e = 1.4426950408889634
    if regime in ("low", "high", "mixed", "band_edge"):
        # shift every score of selected query rows by a constant: add c * u to q and make every key carry a component along u
        u = torch.zeros(heads * 64, device=DEV)
        u[0::64] = 1.0                                            # channel 0 of every head
        k = k.clone(); k[..., 0::64] = 1.0
        target = {"low": -14.0, "high": 11.0, "mixed": -14.0, "band_edge": 6.0}[regime]      # log2 units
        shift = target / (scale * l2e)
        rows = slice(None) if regime != "mixed" else slice(0, Nq, 3)
        q = q.clone(); q[:, rows, 0::64] = shift
    if regime == "late_peak":
        k = k.clone(); k[:, 400] = q[:, 7] * 6.0                    # one key far above everything, six tiles in
        k[:, 30] = q[:, 7] * 1.5
    out = torch.empty(Bq, Nq, heads * 64, dtype=torch.float16, device=DEV)
    ops.attention(q, k, v, out, Bq=Bq, heads=heads, Nq=Nq, Nk=Nk, ldq=heads * 64, ldk=heads * 64, ldv=heads * 64, ldo=heads * 64, kv_div=1, scale=scale)
    sp = lambda t, n: t.float().view(Bq, n, heads, 64).permute(0, 2, 1, 3)
    ref = sdpa_ref(sp(q, Nq), sp(k, Nk), sp(v, Nk), scale).permute(0, 2, 1, 3).reshape(Bq, Nq, heads * 64)
    check(out, ref, TOL16, f"attention, reference regime {regime}")


@pytest.mark.parametrize("B,T,HW,heads", [(2, 16, 50, 5), (1, 8, 64, 8), (1, 16, 2560, 5), (3, 5, 7, 2)])
def test_temporal_attention(B, T, HW, heads):
    C = heads * 64
    qkv = rnd(B * T * HW, 3 * C)
    out = torch.empty(B * T * HW, C, dtype=torch.float16, device=DEV)
    ops.temporal_attention(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], out, B=B, T=T, HW=HW, heads=heads,
                           ld_qkv=3 * C, ldo=C, scale=0.125)
    x = qkv.view(B, T, HW, 3, heads, 64).permute(3, 0, 2, 4, 1, 5)   # [3,B,HW,heads,T,64]
    ref = sdpa_ref(x[0].reshape(-1, heads, T, 64), x[1].reshape(-1, heads, T, 64), x[2].reshape(-1, heads, T, 64), 0.125)
    ref = ref.view(B, HW, heads, T, 64).permute(0, 3, 1, 2, 4).reshape(B * T * HW, C)
    check(out, ref, TOL16, "temporal attention")


# ---------------------------------------------------------------- layout / embedding
def test_layout_roundtrip_and_concat():
    B, Cc, T, HW = 2, 4, 5, 30
    x = rnd(B, Cc, T, HW, dtype=torch.float32)
    y = torch.empty(B * T * HW, 8, dtype=torch.float16, device=DEV)
    ops.ncthw_to_nhwc(x, y, B=B, Cin=Cc, T=T, HW=HW, Cpad=8)
    ref = x.permute(0, 2, 3, 1).reshape(B * T * HW, Cc).half()
    assert torch.equal(y[:, :Cc], ref) and (y[:, Cc:] == 0).all()
    back = torch.empty(B, Cc, T, HW, dtype=torch.float32, device=DEV)
    ops.nhwc_to_ncthw(y, 8, back, B=B, Cout=Cc, T=T, HW=HW)
    assert torch.equal(back, x.half().float())
    a, b = rnd(100, 64), rnd(100, 40)
    o = torch.empty(100, 104, dtype=torch.float16, device=DEV)
    ops.concat_channels(a, b, o, rows=100, C1=64, C2=40)
    assert torch.equal(o, torch.cat([a, b], 1))


def test_timestep_embedding_and_silu_rows():
    t = torch.tensor([0, 1, 17, 500, 999, 250], dtype=torch.int64, device=DEV)
    out = torch.empty(6, 320, dtype=torch.float16, device=DEV)
    ops.timestep_embedding(t, out, n=6, dim=320)
    half = 160
    freqs = torch.exp(-math.log(10000) * torch.arange(half, dtype=torch.float32, device=DEV) / half)
    args = t[:, None].float() * freqs[None]
    ref = torch.cat([torch.cos(args), torch.sin(args)], -1)
    assert (out.float() - ref).abs().max().item() < 2e-3
    # the REAL reference's `timestep_embedding` (utils_diffusion.py:8-28) on its own timesteps: tests/golden/timestep_embedding.npz
    from helpers import golden
    g = golden("timestep_embedding")
    tg = torch.from_numpy(g["t"]).to(DEV)
    og = torch.empty(tg.shape[0], 320, dtype=torch.float16, device=DEV)
    ops.timestep_embedding(tg, og, n=tg.shape[0], dim=320)
    assert (og.float().cpu() - torch.from_numpy(g["y"])).abs().max().item() < 2e-3       # fp16 output of values in [-1, 1] (4.9e-4) + the fp32 argument error at t = 999
    a, b = rnd(6, 128), rnd(2, 128)
    o = torch.empty(6, 128, dtype=torch.float16, device=DEV)
    ops.silu_add_rows(a, 1, b, 3, o, rows=6, Cn=128, silu=True)
    check(o, F.silu(a.float() + b.float().repeat_interleave(3, 0)), TOL16, "silu_add_rows")


# ---------------------------------------------------------------- GEMM: randomized sweep over every big kernel
def _sweep_cases(n, seed):
    import random
    r = random.Random(seed)
    out = []
    for _ in range(n):
        mode = r.choice(["linear", "linear", "conv", "tconv"])
        N = r.choice([160, 320, 480, 640, 128, 256])
        splits = r.choice([1, 1, 1, 2, 3])
        if mode == "linear":
            M, K = r.randint(161, 1500), r.choice([64, 72, 128, 320, 328, 640, 1000, 1280, 2048])
            out.append((mode, M, N, K, splits, r.random() < 0.5, (0, 0, 0, 0)))
        elif mode == "conv":
            Fr, H, W, C = r.randint(1, 4), r.randint(5, 14), r.randint(5, 16), r.choice([8, 64, 128, 192, 320])
            out.append((mode, Fr * H * W, N, 9 * C, splits, r.random() < 0.5, (Fr, H, W, C)))
        else:
            B, T, HW, C = r.randint(1, 2), r.choice([4, 8, 16]), r.randint(6, 60), r.choice([64, 128, 320])
            out.append((mode, B * T * HW, N, 3 * C, splits, r.random() < 0.5, (B, T, HW, C)))
    return out


@pytest.mark.parametrize("kernel,env", [("w80", {L.MOCA_TUNE_GEMM_W80: 2}), ("glds", {L.MOCA_TUNE_GEMM_W80: 0, L.MOCA_TUNE_GEMM_G4: 0}),
                                        ("g4", {L.MOCA_TUNE_GEMM_W80: 0, L.MOCA_TUNE_GEMM_G4: 2})])
def test_gemm_random_sweep_forced_kernels(kernel, env, tune):
    """30 seeded random shapes per kernel family (forced through the MOCA_TUNE_* knobs): linear / conv3x3 / temporal conv,
    M tails, K % 64 != 0 (slow gather path), split-k, residual.  The kernels hand data over through counted vmcnt waits;
    this sweep is there to trip a mis-counted wait, which a handful of fixed shapes can miss."""
    for k, v in env.items():
        tune(k, v)
    for ci, (mode, M, N, K, splits, with_res, geo) in enumerate(_sweep_cases(30, {"w80": 1, "glds": 2, "g4": 3}[kernel])):
        res = rnd(M, N) if with_res else None
        b = rnd(N, dtype=torch.float32)
        if mode == "linear":
            a, w = rnd(M, K), rnd(N, K, scale=K ** -0.5)
            pw = ops.pack_linear(w, b)
            kw = {}
            ref = a.float() @ w.float().t() + b
        elif mode == "conv":
            Fr, H, W, C = geo
            x = rnd(Fr, C, H, W)
            w = rnd(N, C, 3, 3, scale=(9 * C) ** -0.5)
            pw = ops.pack_conv3x3(w, b)
            a = nhwc(x).reshape(M, C)
            kw = dict(mode=L.MOCA_A_CONV3X3, conv=(C, H, W, H, W, 1, 0))
            ref = F.conv2d(x.float(), w.float(), b, padding=1).permute(0, 2, 3, 1).reshape(M, N)
        else:
            B, T, HW, C = geo
            x = rnd(B, C, T, HW, 1)
            w = rnd(N, C, 3, 1, 1, scale=(3 * C) ** -0.5)
            pw = ops.pack_tconv3(w, b)
            a = x.permute(0, 2, 3, 4, 1).reshape(M, C).contiguous()
            kw = dict(mode=L.MOCA_A_TCONV3, tconv=(C, T, HW))
            ref = F.conv3d(x.float(), w.float(), b, padding=(1, 0, 0)).permute(0, 2, 3, 4, 1).reshape(M, N)
        if res is not None:
            ref = ref + res.float()
        out = torch.empty(M, pw.N, dtype=torch.float16, device=DEV)
        ws = torch.empty(splits * M * pw.N, dtype=torch.float32, device=DEV) if splits > 1 else None
        ops.gemm(a, pw, out, M=M, residual=res, splits=splits, splitk_ws=ws, **kw)
        check(out[:, :N], ref, TOL16, f"{kernel} case {ci}: {mode} M={M} N={N} K={K} splits={splits} res={with_res}")


# ---------------------------------------------------------------- Upsample as four 2x2 phase convs
@pytest.mark.parametrize("Fr,H,W,C,N", [(32, 10, 16, 1280, 1280),      # the 1280-channel Upsample of the B=2 forward (256-row kernel)
                                        (8, 20, 32, 640, 640),         # 640 channels
                                        (32, 20, 32, 640, 640),        # the 640-channel Upsample of the B=2 forward (320 x 160 tiles: statistics epilogue)
                                        (3, 7, 9, 64, 320), (5, 6, 11, 128, 256)])   # odd grids, M tails
def test_gemm_upconv_phases(Fr, H, W, C, N):
    """nearest x2 + conv3x3 (openaimodel3d.py:96-106) as four 2x2 convs on the low-resolution grid with pre-summed taps
    (ops.pack_upconv_phases, moca_gemm_params.up_phase): against F.interpolate + F.conv2d in fp32 AND against the 3x3 kernel with
    the `up` gather (same sums in another order: the weight pairs are rounded to fp16 after the addition instead of before)"""
    x = rnd(Fr, C, H, W)
    w, b = rnd(N, C, 3, 3, scale=(9 * C) ** -0.5), rnd(N, dtype=torch.float32)
    a = nhwc(x).reshape(Fr * H * W, C)
    M = Fr * H * W
    out = torch.full((4 * M, N), float("nan"), dtype=torch.float16, device=DEV)
    for ph, pw in enumerate(ops.pack_upconv_phases(w, b)):
        ops.gemm(a, pw, out, M=M, mode=L.MOCA_A_CONV3X3, conv=(C, H, W, H, W, 1, 0), up_phase=ph + 1)
    ref = F.conv2d(F.interpolate(x.float(), scale_factor=2, mode="nearest"), w.float(), b, padding=1).permute(0, 2, 3, 1).reshape(4 * M, N)
    assert torch.isfinite(out).all(), "a phase left output pixels unwritten"
    check(out, ref, TOL16, f"upsample phases {Fr}x{H}x{W} C={C} N={N}")
    one = torch.empty_like(out)
    ops.gemm(a, ops.pack_conv3x3(w, b), one, M=4 * M, mode=L.MOCA_A_CONV3X3, conv=(C, H, W, 2 * H, 2 * W, 1, 1))
    check(out, one.float(), TOL16, "phases vs the 3x3 conv with the upsampling gather")
    # the four launches accumulating the per-frame GroupNorm statistics of the upsampled map, as one source of a virtual concat
    # (groups of `gw` channels starting at channel `coff`): same outputs, statistics = those of the values stored
    rows = ops.gemm_colsum_rows(a, ops.pack_upconv_phases(w, b)[0], M=M, mode=L.MOCA_A_CONV3X3, conv=(C, H, W, H, W, 1, 0))
    if rows > 0 and (H * W) % rows == 0:
        gw, coff = (N + 320) // 32, 0
        gst = torch.zeros(Fr * 64, dtype=torch.int64, device=DEV)
        out2 = torch.full_like(out, float("nan"))
        for ph, pw in enumerate(ops.pack_upconv_phases(w, b)):
            ops.gemm(a, pw, out2, M=M, mode=L.MOCA_A_CONV3X3, conv=(C, H, W, H, W, 1, 0), up_phase=ph + 1, gstat=(gst, H * W, gw, coff))
        assert torch.equal(out2, out)
        acc = torch.zeros_like(gst)
        ops.gstat_accum(out, acc, F=Fr, HW=4 * H * W, Cn=N, frames_per_stat=1, cpg=gw, coff=coff)
        sc = torch.tensor([2.0 ** -20, 2.0 ** -12], dtype=torch.float64, device=DEV)
        assert relerr(gst.view(Fr, 32, 2).double() * sc, acc.view(Fr, 32, 2).double() * sc) < 2e-3
    else:
        assert (Fr, H, W) != (32, 20, 32), "the 640-channel Upsample of the B = 2 forward is expected to take the statistics epilogue"


# ---------------------------------------------------------------- CLIP text tower kernels
@pytest.mark.parametrize("B,heads,N", [(1, 16, 77), (2, 2, 77), (1, 4, 200)])
def test_attention_causal(B, heads, N):
    C = heads * 64
    qkv = rnd(B, N, 3 * C)
    q, k, v = qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:]
    out = torch.empty(B, N, C, dtype=torch.float16, device=DEV)
    ops.attention_causal(q, k, v, out, B=B, heads=heads, N=N, ldq=3 * C, ldk=3 * C, ldv=3 * C, ldo=C, scale=0.125)
    sp = lambda t: t.float().view(B, N, heads, 64).permute(0, 2, 1, 3)
    mask = torch.full((N, N), float("-inf"), device=DEV).triu_(1)
    s = torch.einsum("bhid,bhjd->bhij", sp(q), sp(k)) * 0.125 + mask
    ref = torch.einsum("bhij,bhjd->bhid", s.softmax(-1), sp(v)).permute(0, 2, 1, 3).reshape(B, N, C)
    check(out, ref, TOL16, "causal attention")


def test_gemm_gelu_epilogue_and_token_embedding():
    M, K, N = 77, 256, 512
    a, w, b = rnd(M, K), rnd(N, K, scale=K ** -0.5), rnd(N, dtype=torch.float32, scale=0.5)
    pw = ops.pack_linear(w, b)
    out = torch.empty(M, N, dtype=torch.float16, device=DEV)
    ops.gemm(a, pw, out, M=M, gelu=True)
    check(out, F.gelu(a.float() @ w.float().t() + b), TOL16, "gelu epilogue")
    big = rnd(300, K)                      # M > 128 is forced onto the 128-row kernel by the flag
    out2 = torch.empty(300, N, dtype=torch.float16, device=DEV)
    ops.gemm(big, pw, out2, M=300, gelu=True)
    check(out2, F.gelu(big.float() @ w.float().t() + b), TOL16, "gelu epilogue M=300")
    table, pos = rnd(50, 64, dtype=torch.float32), rnd(7, 64, dtype=torch.float32)
    tok = torch.tensor([3, 49, 0, 7, 7, 12, 1, 5, 48, 2, 9, 9, 30, 31], device=DEV)
    emb = torch.empty(14, 64, dtype=torch.float16, device=DEV)
    ops.embed_tokens(tok, table, pos, emb, n_tokens=14, L=7, Cn=64, vocab=50)
    check(emb, table[tok] + pos.repeat(2, 1), 1e-3, "token embedding")


# ---------------------------------------------------------------- GEMM: the 256 x 256 tiling of the wide projections
@pytest.mark.parametrize("M,K,N,geglu,with_res", [(5000, 640, 2560, True, False), (4500, 1280, 3072, False, True),
                                                  (5120, 384, 5120, True, False), (4100, 1280, 3072, False, False)])
def test_gemm_sq256(M, K, N, geglu, with_res, tune):
    """MOCA_TUNE_GEMM_SQ256 = 2 sends every wide linear (N >= 2560, >= 200 tiles) to the 256 x 256 staggered kernel: M tails, an odd
    number of 64-deep k-tiles, the GEGLU and the residual store loops."""
    tune(L.MOCA_TUNE_GEMM_SQ256, 2)
    a = rnd(M, K)
    w = rnd(N, K, scale=K ** -0.5)
    b = rnd(N, dtype=torch.float32, scale=0.1)
    y = a.float() @ w.float().t() + b
    if geglu:
        pw = ops.pack_geglu(w, b)
        out = torch.empty(M, N // 2, dtype=torch.float16, device=DEV)
        ops.gemm(a, pw, out, M=M)
        ref = y[:, :N // 2] * F.gelu(y[:, N // 2:])
    else:
        pw = ops.pack_linear(w, b)
        res = rnd(M, N) if with_res else None
        out = torch.empty(M, N, dtype=torch.float16, device=DEV)
        ops.gemm(a, pw, out, M=M, residual=res)
        ref = y + res.float() if with_res else y
    check(out, ref, TOL16, f"sq256 {M}x{N}x{K} geglu={geglu}")


# ---------------------------------------------------------------- GEMM: the persistent two-blocks-per-CU kernel (register epilogue)
@pytest.mark.parametrize("M,K,N,kind", [(8200, 320, 2048, "geglu"), (16384, 640, 1024, "geglu"), (8192, 640, 2048, "res"),
                                        (9000, 128, 2048, "rowadd+res"), (8192, 64, 2048, "plain"), (70000, 320, 256, "res")])
@pytest.mark.parametrize("kern", ["g4p", "g4q", "sqp"])
def test_gemm_g4p(M, K, N, kind, kern, tune):
    """MOCA_TUNE_GEMM_G4P = 2 sends every linear with >= 512 tiles of 256 x 128 to the persistent kernel: W rows fetched in permuted
    order, 16-byte stores straight from the accumulators (GEGLU / bias / row add / residual), M tails, a block's walk over several
    tiles with the DMA stream running across tile boundaries, K = 64 (one k-tile pair per tile)."""
    # g4p / g4q: 256 x 128 tiles, two blocks per CU, both MFMA shapes (16x16x32 / 32x32x16: different W-row permutations and store maps);
    # sqp: the persistent 256 x 256 staggered kernel (continuous DMA stream, statistics in the ring's free slot)
    tune(L.MOCA_TUNE_GEMM_SQP, 2 if kern == "sqp" else 0)
    tune(L.MOCA_TUNE_GEMM_G4P, 2)
    tune(L.MOCA_TUNE_GEMM_MF32, 1 if kern == "g4q" else 0)
    a = rnd(M, K)
    w = rnd(N, K, scale=K ** -0.5)
    b = rnd(N, dtype=torch.float32, scale=0.1)
    y = a.float() @ w.float().t() + b
    if kind == "geglu":
        pw = ops.pack_geglu(w, b)
        out = torch.full((M, N // 2), float("nan"), dtype=torch.float16, device=DEV)
        ops.gemm(a, pw, out, M=M)
        ref = y[:, :N // 2] * F.gelu(y[:, N // 2:])
    else:
        pw = ops.pack_linear(w, b)
        res = rnd(M, N) if "res" in kind else None
        div = 100
        ra = rnd(M // div, N) if "rowadd" in kind else None
        out = torch.full((M, N), float("nan"), dtype=torch.float16, device=DEV)
        ops.gemm(a, pw, out, M=M, residual=res, rowadd=ra, rowadd_div=div if ra is not None else 1)
        ref = y
        if ra is not None:
            ref = ref + ra.float().repeat_interleave(div, dim=0)
        if res is not None:
            ref = ref + res.float()
    check(out, ref, TOL16, f"g4p {M}x{N}x{K} {kind}")


def test_gemm_sqp_lnfold_repeatable(tune):
    """The persistent kernel fetches the NEXT tile's LayerNorm-fold statistics under the current tile's epilogue and reads them back
    behind a counted wait.  A read that is not ordered behind its load (the first form of this path: inline-asm VGPR loads, which the
    register allocator copies ahead of the asm wait that retires them) returns stale values whenever a load is slower than the epilogue.
    400 launches of the 320-channel GEGLU shape with cache-evicting fills in between, every one bit-identical to the first and to torch."""
    tune(L.MOCA_TUNE_GEMM_SQP, 1)
    M, C, n = 81920, 320, 1280
    x = rnd(M, C) * 3 + 1.5
    xf = x.float()
    part = torch.stack([xf.sum(1), (xf * xf).sum(1)], 1).contiguous()                 # one row partial (sum, sum of squares)
    g = rnd(C, dtype=torch.float32) * 0.3 + 1.0
    be = rnd(C, dtype=torch.float32) * 0.3
    wc, bc = rnd(2 * n, C, scale=C ** -0.5), rnd(2 * n, dtype=torch.float32, scale=0.1)
    wf, bf = ops.fold_layernorm(wc, bc, g, be)
    pwf = ops.finish_lnfold(ops.pack_geglu(wf, bf))
    y = F.layer_norm(xf, (C,), g, be, 1e-5) @ wc.float().t() + bc
    ref = y[:, :n] * F.gelu(y[:, n:])
    first = torch.empty(M, n, dtype=torch.float16, device=DEV)
    ops.gemm(x, pwf, first, M=M, lnfold=(part, 1, 1e-5))
    check(first, ref, TOL16, "sqp lnfold geglu")
    out = torch.empty_like(first)
    junk = torch.empty(64 << 20, dtype=torch.float16, device=DEV)
    bad = 0
    for i in range(400):
        if i % 8 == 0:
            junk.fill_(float(i))                                 # (evicts the statistics from the caches now and then)
        out.fill_(float("nan"))
        ops.gemm(x, pwf, out, M=M, lnfold=(part, 1, 1e-5))
        bad += int(not torch.equal(out, first))
    assert bad == 0, f"{bad} of 400 launches differ from the first"


# ---------------------------------------------------------------- LayerNorm folded into the consuming linear
@pytest.mark.parametrize("M,C,Kp,with_res,consumers", [
    (40000, 320, 320, True, [("lin", 960), ("geglu", 1280)]),            # 160 x 320 producer (1 partial); staggered / g4 consumers
    (20480, 640, 640, True, [("lin", 1920), ("geglu", 2560)]),           # 2 partials
    (5120, 1280, 1280, False, [("lin", 3840), ("lin", 1280), ("geglu", 5120)]),   # 256-row producer (10 partials); staggered, 256-row, sq256
    (33000, 320, 320, True, [("lin", 960)]),                             # M tail
    (33000, -320, 320, True, [("lin", 960)]),                            # (C < 0: MOCA_TUNE_GEMM_WIDE = 0) 320 x 160 producer, 2 partials
    (4000, 512, 320, False, [("lin", 1536), ("geglu", 2048)])])          # init_attn widths: 256-row producer (4 partials)
def test_gemm_rowsum_feeds_lnfold(M, C, Kp, with_res, consumers, tune):
    """MOCA_EP_ROWSUM + MOCA_EP_LNFOLD: the producer linear leaves per-(column tile, row) sums of what it stores; the consumer
    runs on x with W' = W diag(gamma), b' = b + W beta and finishes Linear(LayerNorm(x)) in its epilogue (ref: torch)."""
    if C < 0:
        C = -C
        tune(L.MOCA_TUNE_GEMM_WIDE, 0)
    a, w, b = rnd(M, Kp), rnd(C, Kp, scale=Kp ** -0.5), rnd(C, dtype=torch.float32)
    res = rnd(M, C) * 2 + 0.5 if with_res else None          # (a non-zero row mean: the fold subtracts mean * wsum)
    pw = ops.pack_linear(w, b)
    cols = ops.gemm_rowsum_cols(a, pw, M=M, residual=res, rowsum=True)
    assert cols > 0 and C % cols == 0
    nparts = C // cols
    x = torch.empty(M, C, dtype=torch.float16, device=DEV)
    part = torch.full((nparts * M, 2), float("nan"), dtype=torch.float32, device=DEV)
    ops.gemm(a, pw, x, M=M, residual=res, rowsum=part)
    xr = a.float() @ w.float().t() + b
    if with_res:
        xr = xr + res.float()
    check(x, xr, TOL16, "producer")
    pr = part.view(nparts, M, 2).sum(0)
    xf = x.float()
    assert relerr(pr[:, 0], xf.sum(1)) < 1e-4 and relerr(pr[:, 1], (xf * xf).sum(1)) < 1e-4
    g = rnd(C, dtype=torch.float32) * 0.3 + 1.0
    be = rnd(C, dtype=torch.float32) * 0.3
    ln = F.layer_norm(xf, (C,), g, be, 1e-5)
    for kind, n in consumers:
        if kind == "lin":
            wc, bc = rnd(n, C, scale=C ** -0.5), rnd(n, dtype=torch.float32)
            wf, bf = ops.fold_layernorm(wc, bc, g, be)
            pwf = ops.finish_lnfold(ops.pack_linear(wf, bf))
            ref = ln @ wc.float().t() + bc
            n_out = n
        else:
            wc, bc = rnd(2 * n, C, scale=C ** -0.5), rnd(2 * n, dtype=torch.float32, scale=0.1)
            wf, bf = ops.fold_layernorm(wc, bc, g, be)
            pwf = ops.finish_lnfold(ops.pack_geglu(wf, bf))
            y = ln @ wc.float().t() + bc
            ref = y[:, :n] * F.gelu(y[:, n:])
            n_out = n
        assert ops.gemm_lnfold_ok(x, pwf, M=M, lnfold=(None, nparts, 1e-5)), f"{kind} N={n}: expected a kernel with the fold epilogue"
        out = torch.empty(M, n_out, dtype=torch.float16, device=DEV)
        ops.gemm(x, pwf, out, M=M, lnfold=(part, nparts, 1e-5))
        check(out, ref, TOL16, f"lnfold consumer {kind} N={n} (C={C}, {nparts} partials)")


@pytest.mark.parametrize("offset,C", [(40.0, 320), (100.0, 640), (-60.0, 1280)])
def test_gemm_lnfold_offset_rows(offset, C):
    """The LayerNorm fold derives the variance single-pass from fp32 row partials (q / K - mean^2) and the output as
    rstd * (acc - mean * wsum): rows whose mean is far from zero (|mean| / std = 40 .. 100, as on outlier channels of the residual
    stream) lose digits to cancellation in both terms.  Bound: log2(100) ~ 7 of fp32's 24 mantissa bits -- the result must still
    agree with the two-pass torch LayerNorm of the same fp16 rows to the fp16 kernel tolerance."""
    M = 20480
    a, w, b = rnd(M, C), rnd(C, C, scale=C ** -0.5), rnd(C, dtype=torch.float32)
    res = (rnd(M, C).float() + offset).half()
    pw = ops.pack_linear(w, b)
    cols = ops.gemm_rowsum_cols(a, pw, M=M, residual=res, rowsum=True)
    assert cols > 0
    nparts = C // cols
    x = torch.empty(M, C, dtype=torch.float16, device=DEV)
    part = torch.empty(nparts * M, 2, dtype=torch.float32, device=DEV)
    ops.gemm(a, pw, x, M=M, residual=res, rowsum=part)
    xf = x.float()
    assert abs(float(xf.mean()) - offset) < 1.0 and 0.5 < float(xf.std(dim=1).mean()) < 3.0
    g = rnd(C, dtype=torch.float32) * 0.3 + 1.0
    be = rnd(C, dtype=torch.float32) * 0.3
    ln = F.layer_norm(xf, (C,), g, be, 1e-5)
    wc, bc = rnd(3 * C, C, scale=C ** -0.5), rnd(3 * C, dtype=torch.float32)
    wf, bf = ops.fold_layernorm(wc, bc, g, be)
    pwf = ops.finish_lnfold(ops.pack_linear(wf, bf))
    assert ops.gemm_lnfold_ok(x, pwf, M=M, lnfold=(None, nparts, 1e-5))
    out = torch.empty(M, 3 * C, dtype=torch.float16, device=DEV)
    ops.gemm(x, pwf, out, M=M, lnfold=(part, nparts, 1e-5))
    check(out, ln @ wc.float().t() + bc, TOL16, f"lnfold with row mean {offset}")


@pytest.mark.parametrize("Fr,HW,C1,C2", [(4, 100, 320, 320), (2, 2560, 640, 320), (3, 37, 1280, 1280), (2, 160, 1280, 640)])
def test_concat_with_groupnorm_statistics(Fr, HW, C1, C2):
    """torch.cat(dim=channels) that also accumulates the statistics of the GroupNorm that follows (openaimodel3d.py:571,149)"""
    a, b = rnd(Fr * HW, C1) * 1.3 + 0.2, rnd(Fr * HW, C2) * 0.7 - 0.4
    C = C1 + C2
    out = torch.empty(Fr * HW, C, dtype=torch.float16, device=DEV)
    gst = torch.zeros(Fr * 64, dtype=torch.int64, device=DEV)
    ops.concat_channels_gstat(a, b, out, gst, F=Fr, HW=HW, C1=C1, C2=C2, frames_per_stat=1)
    ref = torch.cat([a, b], dim=1)
    assert torch.equal(out, ref)
    xg = ref.float().view(Fr, HW, 32, C // 32)
    gs = gst.view(Fr, 32, 2).double() * torch.tensor([2.0 ** -20, 2.0 ** -12], dtype=torch.float64, device=DEV)   # fixed point
    assert relerr(gs[..., 0], xg.sum(dim=(1, 3))) < 1e-3 and relerr(gs[..., 1], (xg * xg).sum(dim=(1, 3))) < 1e-3
    g, be = rnd(C, dtype=torch.float32) * 0.2 + 1.0, rnd(C, dtype=torch.float32) * 0.2
    y = torch.empty_like(out)
    ops.groupnorm_gstat(out, y, g, be, gst, F=Fr, HW=HW, Cn=C, frames_per_stat=1, eps=1e-5, silu=True)
    gref = F.silu(F.group_norm(ref.float().view(Fr, HW, C).permute(0, 2, 1), 32, g, be, 1e-5)).permute(0, 2, 1).reshape(Fr * HW, C)
    check(y, gref, TOL16, "groupnorm of the concatenation")
    # non-finite activations must poison their statistics group (the fixed-point accumulators cannot hold NaN / Inf: a poison bit
    # does): one NaN and, separately, one Inf in frame 1 / channel group of column 5 -> that group's outputs are all NaN, every
    # other (frame, group) is untouched
    cpg = C // 32
    for bad in (float("nan"), float("inf"), float("-inf")):
        a2 = a.clone()
        a2[HW + 3, 5] = bad
        gst.zero_()
        ops.concat_channels_gstat(a2, b, out, gst, F=Fr, HW=HW, C1=C1, C2=C2, frames_per_stat=1)
        ops.groupnorm_gstat(out, y, g, be, gst, F=Fr, HW=HW, Cn=C, frames_per_stat=1, eps=1e-5, silu=True)
        yv = y.view(Fr, HW, 32, cpg)
        gi = 5 // cpg
        assert torch.isnan(yv[1, :, gi]).all(), f"{bad}: the poisoned group must come out NaN"
        keep = torch.ones(Fr, 32, dtype=torch.bool, device=DEV)
        keep[1, gi] = False
        assert torch.equal(yv.permute(0, 2, 1, 3)[keep], gref.view(Fr, HW, 32, cpg).permute(0, 2, 1, 3)[keep].to(y.dtype)) or \
            relerr(yv.permute(0, 2, 1, 3)[keep], gref.view(Fr, HW, 32, cpg).permute(0, 2, 1, 3)[keep]) < TOL16
    # LARGE finite activations: a group at rms 1e3 or 4e3 (the latents of bench.py's synthetic video) lies inside the
    # fixed-point range and must be right; at rms 3e4 a partial can leave the range -- that group then reads back NaN (poisoned),
    # never finite-but-wrong (out-of-range partials used to be clamped)
    for scale, must_match in ((1.0e3, True), (4.0e3, True), (3.0e4, False)):
        a2 = a.clone()
        a2[HW:2 * HW, :cpg] = (torch.randn(HW, cpg, device=DEV) * scale).clamp(-6.0e4, 6.0e4).half()
        gst.zero_()
        ops.concat_channels_gstat(a2, b, out, gst, F=Fr, HW=HW, C1=C1, C2=C2, frames_per_stat=1)
        ops.groupnorm_gstat(out, y, g, be, gst, F=Fr, HW=HW, Cn=C, frames_per_stat=1, eps=1e-5, silu=True)
        r2 = torch.cat([a2, b], dim=1).float()
        gref2 = F.silu(F.group_norm(r2.view(Fr, HW, C).permute(0, 2, 1), 32, g, be, 1e-5)).permute(0, 2, 1).reshape(Fr * HW, C)
        got, want = y.view(Fr, HW, 32, cpg)[1, :, 0], gref2.view(Fr, HW, 32, cpg)[1, :, 0]
        if must_match or not torch.isnan(got).all():
            assert relerr(got, want) < TOL16, f"rms {scale:g}: finite GroupNorm output that is not the fp32 reference's"
        keep = torch.ones(Fr, 32, dtype=torch.bool, device=DEV)
        keep[1, 0] = False
        assert relerr(y.view(Fr, HW, 32, cpg).permute(0, 2, 1, 3)[keep], gref2.view(Fr, HW, 32, cpg).permute(0, 2, 1, 3)[keep]) < TOL16


# ---------------------------------------------------------------- split-K reduce inside the consuming GroupNorm (the 5 x 8-latent level)
@pytest.mark.parametrize("Fr,HW,fps,mode,splits,res,radd", [(32, 40, 16, "tconv", 4, False, False), (32, 40, 16, "conv", 5, True, False),
                                                           (32, 40, 1, "conv", 5, False, True), (16, 40, 16, "tconv", 4, True, False),
                                                           (32, 40, 1, "lin", 4, True, False)])
def test_gemm_splitk_groupnorm(Fr, HW, fps, mode, splits, res, radd):
    """MOCA_EP_SLABS + moca_gemm_splitk_groupnorm_f16 against the three-launch path (GEMM, reduce, GroupNorm): x bit-identical when it is
    written, y to the fp16 tolerance (same fp16 x, statistics summed in another order)"""
    _splitk_groupnorm_case(Fr, HW, fps, mode, splits, res, radd)


def test_gemm_splitk_groupnorm_fewer_chunks_than_group_channels():
    """a statistics group of 128 channels over 4 rows is 64 eight-channel chunks: the block must still have >= 128 threads, which
    fill the per-channel scale / shift table (64 were launched and half the table was uninitialised LDS)"""
    _splitk_groupnorm_case(320, 4, 1, "lin", 4, True, False, C=1280, N=4096)


def _splitk_groupnorm_case(Fr, HW, fps, mode, splits, res, radd, C=1280, N=1280):
    M = Fr * HW
    H, W = (5, 8) if HW == 40 else (HW // 8, 8)
    if mode == "tconv":
        a, K, kw = rnd(M, C), 3 * C, dict(mode=L.MOCA_A_TCONV3, tconv=(C, 16, HW))
        pw = ops.pack_tconv3(rnd(N, C, 3, 1, 1, scale=K ** -0.5), rnd(N, dtype=torch.float32))
    elif mode == "conv":
        a, K, kw = rnd(M, C), 9 * C, dict(mode=L.MOCA_A_CONV3X3, conv=(C, H, W, H, W, 1, 0))
        pw = ops.pack_conv3x3(rnd(N, C, 3, 3, scale=K ** -0.5), rnd(N, dtype=torch.float32))
    else:
        a, K, kw = rnd(M, 2 * C), 2 * C, dict()
        pw = ops.pack_linear(rnd(N, K, scale=K ** -0.5), rnd(N, dtype=torch.float32))
    r = rnd(M, N) if res else None
    ra = rnd(Fr, N) if radd else None
    kw.update(M=M, splits=splits, residual=r, rowadd=ra, rowadd_div=HW if radd else 1)
    ws = torch.empty(splits * M * N, dtype=torch.float32, device=DEV)
    g, be = rnd(N, dtype=torch.float32) * 0.2 + 1.0, rnd(N, dtype=torch.float32) * 0.2
    x0 = torch.empty(M, N, dtype=torch.float16, device=DEV)
    ops.gemm(a, pw, x0, splitk_ws=ws, **kw)
    y0 = torch.empty_like(x0)
    wsg = torch.empty(ops.groupnorm_ws_floats(Fr, HW, N), dtype=torch.float32, device=DEV)
    ops.groupnorm(x0, y0, g, be, F=Fr, HW=HW, Cn=N, frames_per_stat=fps, eps=1e-5, silu=True, ws=wsg)
    assert ops.gemm_splitk_groupnorm_ok(a, pw, HW=HW, frames_per_stat=fps, splitk_ws=ws, **kw)
    for write_x in (True, False):
        x1 = torch.full_like(x0, float("nan"))
        y1 = torch.empty_like(x0)
        ws.fill_(float("nan"))
        ops.gemm(a, pw, x1, splitk_ws=ws, slabs=True, **kw)
        assert torch.isnan(x1).all(), "a MOCA_EP_SLABS launch must not write its output"
        ops.gemm_splitk_groupnorm(a, pw, x1, y1, g, be, HW=HW, frames_per_stat=fps, eps=1e-5, silu=True, write_x=write_x, splitk_ws=ws, **kw)
        if write_x:
            assert torch.equal(x1, x0)
        else:
            assert torch.isnan(x1).all()
        check(y1, y0.float(), 1e-3, f"split-K reduce + GroupNorm ({mode}, write_x={write_x})")
    # refused: no split-K, or another epilogue flag
    kw1 = dict(kw, splits=1)
    assert not ops.gemm_splitk_groupnorm_ok(a, pw, HW=HW, frames_per_stat=fps, **kw1)
    with pytest.raises(L.MocaHipError):
        ops.gemm(a, pw, x0, slabs=True, **kw1)


def test_groupnorm_gstat_rejects_misaligned_affine_parameters():
    """gamma / beta are fetched as 16-byte vectors by the statistics-fed apply kernels: a parameter pointer that is not 16-byte aligned is refused"""
    Fr, HW, C = 2, 64, 320
    x = rnd(Fr * HW, C)
    y = torch.empty_like(x)
    gst = torch.zeros(Fr * 64, dtype=torch.int64, device=DEV)
    ops.gstat_accum(x, gst, F=Fr, HW=HW, Cn=C, frames_per_stat=1, cpg=C // 32, coff=0)
    buf = rnd(2 * C + 8, dtype=torch.float32)
    ops.groupnorm_gstat(x, y, buf[:C], buf[C:2 * C], gst, F=Fr, HW=HW, Cn=C, frames_per_stat=1, eps=1e-5, silu=False)
    with pytest.raises(L.MocaHipError):
        ops.groupnorm_gstat(x, y, buf[1:C + 1], buf[C:2 * C], gst, F=Fr, HW=HW, Cn=C, frames_per_stat=1, eps=1e-5, silu=False)


# ---------------------------------------------------------------- the virtual torch.cat of the output blocks (openaimodel3d.py:571)
@pytest.mark.parametrize("M,C1,C2,N,res", [(81920, 640, 320, 320, False), (81920, 320, 320, 320, True), (20480, 1280, 640, 640, False),
                                           (20480, 640, 320, 640, True), (20000, 640, 640, 640, False), (40960, 512, 256, 320, False)])
def test_gemm_two_source_a(M, C1, C2, N, res):
    """moca_gemm_params.a2: the skip_connection 1x1 conv reads cat([h, skip], channels) from its two sources -- bit-identical to
    the same kernel on the materialised concat (same k order), and against torch"""
    h, sk = rnd(M, C1) * 1.2 + 0.1, rnd(M, C2) * 0.8 - 0.2
    K = C1 + C2
    w, b = rnd(N, K, scale=K ** -0.5), rnd(N, dtype=torch.float32)
    r = rnd(M, N) if res else None
    pw = ops.pack_linear(w, b)
    assert ops.gemm_cat_ok(h, pw, M=M, residual=r, a2=(sk, C1)), "expected a staggered-kernel launch"
    cat = torch.cat([h, sk], dim=1).contiguous()
    ref_k = torch.empty(M, N, dtype=torch.float16, device=DEV)
    ops.gemm(cat, pw, ref_k, M=M, residual=r)
    out = torch.empty(M, N, dtype=torch.float16, device=DEV)
    ops.gemm(h, pw, out, M=M, residual=r, a2=(sk, C1))
    assert torch.equal(out, ref_k), "two-source A differs from the materialised concat"
    ref = cat.float() @ w.float().t() + b
    if res:
        ref = ref.half().float() + r.float()
    check(out, ref, TOL16, "linear over the virtual concat")
    # sources with wider row strides (views of larger buffers)
    hb, sb = torch.zeros(M, C1 + 64, dtype=torch.float16, device=DEV), torch.zeros(M, C2 + 128, dtype=torch.float16, device=DEV)
    hb[:, :C1] = h
    sb[:, :C2] = sk
    out2 = torch.empty_like(out)
    ops.gemm(hb[:, :C1], pw, out2, M=M, residual=r, a2=(sb[:, :C2], C1))
    assert torch.equal(out2, ref_k)
    # refused: a split that is no multiple of 64 columns, split-K, a GEGLU / LayerNorm-fold epilogue
    assert not ops.gemm_cat_ok(h, pw, M=M, a2=(sk, C1 - 32))
    assert not ops.gemm_cat_ok(h, pw, M=M, splits=2, a2=(sk, C1))
    assert L.load().moca_gemm_f16 is not None
    with pytest.raises(L.MocaHipError):
        ops.gemm(h, pw, out2, M=M, a2=(sk, C1 - 32))


@pytest.mark.parametrize("Fr,HW,C1,C2,own", [(32, 2560, 640, 320, True), (32, 2560, 320, 320, True), (32, 640, 1280, 640, True),
                                             (32, 640, 640, 640, False), (32, 2560, 640, 320, False), (16, 2560, 320, 320, False)])
def test_groupnorm_virtual_cat(Fr, HW, C1, C2, own):
    """GroupNorm(+SiLU) of the never-materialised cat([h, skip]): h's producer accumulates its share of the concat's statistics in the
    concat's grouping (gstat_cpg), skip's share is either its own finished 32-group statistics (merged by the GroupNorm: `own`) or
    accumulated by its producer with a channel offset (gstat_coff); and the read-only statistics pass for producer-less sources"""
    M, C = Fr * HW, C1 + C2
    gw = C // 32
    pws, srcs, outs = [], [], []
    for Cn in (C1, C2):
        a, w, b = rnd(M, Cn), rnd(Cn, Cn, scale=Cn ** -0.5), rnd(Cn, dtype=torch.float32) * 0.5
        pws.append(ops.pack_linear(w, b)); srcs.append(a); outs.append(torch.empty(M, Cn, dtype=torch.float16, device=DEV))
    for (a, pw) in zip(srcs, pws):
        assert ops.gemm_colsum_rows(a, pw, M=M) > 0 and HW % ops.gemm_colsum_rows(a, pw, M=M) == 0
    gcat = torch.zeros(Fr * 64, dtype=torch.int64, device=DEV)
    gown = torch.zeros(Fr * 64, dtype=torch.int64, device=DEV)
    ops.gemm(srcs[0], pws[0], outs[0], M=M, gstat=(gcat, HW, gw, 0))
    if own:
        ops.gemm(srcs[1], pws[1], outs[1], M=M, gstat=(gown, HW))
    else:
        ops.gemm(srcs[1], pws[1], outs[1], M=M, gstat=(gcat, HW, gw, C1))
    h, sk = outs
    ref_cat = torch.cat([h, sk], dim=1)
    g, be = rnd(C, dtype=torch.float32) * 0.2 + 1.0, rnd(C, dtype=torch.float32) * 0.2
    y = torch.empty(M, C, dtype=torch.float16, device=DEV)
    ops.groupnorm_gstat_cat(h, sk, y, g, be, gcat, gown if own else None, F=Fr, HW=HW, C1=C1, C2=C2, frames_per_stat=1, eps=1e-5, silu=True)
    gref = F.silu(F.group_norm(ref_cat.float().view(Fr, HW, C).permute(0, 2, 1), 32, g, be, 1e-5)).permute(0, 2, 1).reshape(M, C)
    check(y, gref, TOL16, "groupnorm of the virtual concat (statistics from the producers)")
    # against the materialising path on the same tensors (statistics of the fp16 values instead of the producers' fp32 ones)
    out = torch.empty(M, C, dtype=torch.float16, device=DEV)
    gst = torch.zeros(Fr * 64, dtype=torch.int64, device=DEV)
    ops.concat_channels_gstat(h, sk, out, gst, F=Fr, HW=HW, C1=C1, C2=C2, frames_per_stat=1)
    y0 = torch.empty_like(y)
    ops.groupnorm_gstat(out, y0, g, be, gst, F=Fr, HW=HW, Cn=C, frames_per_stat=1, eps=1e-5, silu=True)
    assert relerr(y, y0) < 2e-3
    # the read-only statistics pass: both shares from moca_gstat_accum_f16 == the statistics the concat kernel leaves
    gacc = torch.zeros(Fr * 64, dtype=torch.int64, device=DEV)
    ops.gstat_accum(h, gacc, F=Fr, HW=HW, Cn=C1, frames_per_stat=1, cpg=gw, coff=0)
    ops.gstat_accum(sk, gacc, F=Fr, HW=HW, Cn=C2, frames_per_stat=1, cpg=gw, coff=C1)
    sc = torch.tensor([2.0 ** -20, 2.0 ** -12], dtype=torch.float64, device=DEV)
    assert relerr(gacc.view(Fr, 32, 2).double() * sc, gst.view(Fr, 32, 2).double() * sc) < 1e-6
    y1 = torch.empty_like(y)
    ops.groupnorm_gstat_cat(h, sk, y1, g, be, gacc, None, F=Fr, HW=HW, C1=C1, C2=C2, frames_per_stat=1, eps=1e-5, silu=True)
    assert relerr(y1, y0) < 1e-3
    # skip = two copies of Fr / 2 frames whose own statistics cover the distinct frames only (the skip connection out of the shared prefix)
    if gw % (C2 // 32) == 0 and (C1 % gw) % (C2 // 32) == 0:
        Fb = Fr // 2
        skr = torch.cat([sk[:Fb * HW], sk[:Fb * HW]], dim=0).contiguous()
        gb = torch.zeros(Fb * 64, dtype=torch.int64, device=DEV)
        ops.gstat_accum(skr, gb, F=Fb, HW=HW, Cn=C2, frames_per_stat=1, cpg=C2 // 32, coff=0)
        gh = torch.zeros(Fr * 64, dtype=torch.int64, device=DEV)
        ops.gstat_accum(h, gh, F=Fr, HW=HW, Cn=C1, frames_per_stat=1, cpg=gw, coff=0)
        y2 = torch.empty_like(y)
        ops.groupnorm_gstat_cat(h, skr, y2, g, be, gh, gb, F=Fr, HW=HW, C1=C1, C2=C2, frames_per_stat=1, eps=1e-5, silu=True, Fb=Fb)
        rref = torch.cat([h, skr], dim=1)
        gref2 = F.silu(F.group_norm(rref.float().view(Fr, HW, C).permute(0, 2, 1), 32, g, be, 1e-5)).permute(0, 2, 1).reshape(M, C)
        check(y2, gref2, TOL16, "virtual concat with a repeated skip source")
    # a poisoned source poisons exactly the concat groups it belongs to
    sk2 = sk.clone()
    sk2[HW + 5, 3] = float("nan")
    gacc.zero_()
    ops.gstat_accum(h, gacc, F=Fr, HW=HW, Cn=C1, frames_per_stat=1, cpg=gw, coff=0)
    ops.gstat_accum(sk2, gacc, F=Fr, HW=HW, Cn=C2, frames_per_stat=1, cpg=gw, coff=C1)
    ops.groupnorm_gstat_cat(h, sk2, y1, g, be, gacc, None, F=Fr, HW=HW, C1=C1, C2=C2, frames_per_stat=1, eps=1e-5, silu=True)
    yv = y1.view(Fr, HW, 32, gw)
    gi = (C1 + 3) // gw
    assert torch.isnan(yv[1, :, gi]).all() and torch.isfinite(yv[0]).all() and torch.isfinite(yv[1, :, :gi]).all()


# ---------------------------------------------------------------- q|k|v projection + temporal attention in one launch
@pytest.mark.parametrize("B,HW,heads,K,fold", [(2, 640, 5, 320, False), (1, 1280, 5, 320, True), (2, 100, 8, 512, True), (1, 40, 20, 1280, False)])
def test_gemm_temporal_attention_fused(B, HW, heads, K, fold):
    """MOCA_EP_TATTN: to_q|to_k|to_v (packed per head) on tiles of 16 frames x 20 pixels with the attention over the frame axis
    finished in the epilogue -- against the projection (fp16-rounded, as the two-launch path stores it) + temporal attention in
    torch; with MOCA_EP_LNFOLD the input is x and the projection is Linear(LayerNorm(x))."""
    T, C = 16, heads * 64
    M = B * T * HW
    x = rnd(M, K) * 1.5 + (0.3 if fold else 0.0)
    wq, wk, wv = (rnd(C, K, scale=K ** -0.5) for _ in range(3))
    scale = 0.125
    if fold:
        g, be = rnd(K, dtype=torch.float32) * 0.3 + 1.0, rnd(K, dtype=torch.float32) * 0.3
        wf, bf = ops.fold_layernorm(torch.cat([wq, wk, wv]), None, g, be)
        pw = ops.finish_lnfold(ops.pack_qkv_per_head(wf[:C], wf[C:2 * C], wf[2 * C:], heads, bias=bf))
        xf = x.float()
        part = torch.stack([xf.sum(1), (xf * xf).sum(1)], dim=1).contiguous()
        a_in = F.layer_norm(xf, (K,), g, be, 1e-5)
        kw = dict(lnfold=(part, 1, 1e-5))
    else:
        pw = ops.pack_qkv_per_head(wq, wk, wv, heads)
        a_in = x.float()
        kw = {}
    assert ops.gemm_tattn_ok(x, pw, M=M, tattn=(T, HW, scale), **({"lnfold": (None, 1, 1e-5)} if fold else {}))
    out = torch.full((M, C), float("nan"), dtype=torch.float16, device=DEV)
    ops.gemm(x, pw, out, M=M, tattn=(T, HW, scale), **kw)
    q, k, v = ((a_in @ w.float().t()).half().float().view(B, T, HW, heads, 64).permute(0, 2, 3, 1, 4) for w in (wq, wk, wv))   # [B,HW,h,T,64]
    att = torch.softmax(torch.einsum("bphid,bphjd->bphij", q, k) * scale, dim=-1)
    ref = torch.einsum("bphij,bphjd->bphid", att, v).permute(0, 3, 1, 2, 4).reshape(M, C)
    check(out, ref, TOL16, f"fused qkv + temporal attention (heads={heads}, K={K}, fold={fold})")
