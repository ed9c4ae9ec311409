/*
 * moca_hip.h -- C-ABI of libmoca_hip.so: the MI355X (gfx950) kernels behind the
 * MoCA-Video / VideoCrafter2 denoising hot path.
 *
 * Every entry point takes raw device pointers, sizes and a hipStream_t (as void*),
 * launches asynchronously on that stream and returns 0 on success or a negative
 * MOCA_E_* code.  No torch types appear here: the Python host
 * (moca_video_amd/lib.py) binds these with ctypes, exactly as a maintainer of the
 * reference would.
 *
 * The reference (ZhangT-tech/MoCA-Video) has no native code; each entry point
 * cites the reference Python op(s) it replaces (file:line under the reference
 * root).  Activations are channels-last fp16: a feature map of B videos x T
 * frames is stored as [B*T][H][W][C] ("NHWC, frame outermost").
 */
#ifndef MOCA_HIP_H
#define MOCA_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MOCA_OK            0
#define MOCA_E_BADARG     -1   /* shape/alignment contract violated              */
#define MOCA_E_LAUNCH     -2   /* hipLaunchKernel / runtime error                */
#define MOCA_E_NODEVICE   -3   /* no gfx950 device visible                       */
#define MOCA_E_GRAPH      -4   /* stream capture / graph replay failed           */

/* ---- implicit-GEMM: conv2d 3x3, temporal conv (3,1,1), linear ------------- */
/* A-operand gather modes */
#define MOCA_A_LINEAR   0  /* A[m][k] = a[m*lda + k]                                       */
#define MOCA_A_CONV3X3  1  /* k=(ky,kx,c): a[f][oy*s+ky-1][ox*s+kx-1][c], zero padded;      */
                           /* with up=1 the source is read through a nearest x2 upsample     */
#define MOCA_A_TCONV3   2  /* k=(kt,c): a[b][t+kt-1][pix][c], zero padded along t            */

/* epilogue flags */
#define MOCA_EP_GEGLU   1  /* out[m][j] = v[m][j] * gelu(g[m][j]); W rows are packed in     */
                           /* 64-row groups: 32 value rows then their 32 gate rows           */
#define MOCA_EP_OUT_F32 2  /* out is float32 instead of fp16                                 */
#define MOCA_FORCE_SMALL_TILE 4  /* tuning/testing: use the 128-row register-staged kernel   */
#define MOCA_EP_COLSUM 16  /* also write per-(row tile, column) sums and sums of squares of   */
                           /* the stored values to p.colsum (GroupNorm statistics of the     */
                           /* consumer: openaimodel3d.py:149,173,253-262; attention.py:238): */
                           /* only where moca_gemm_colsum_rows() > 0                         */
#define MOCA_EP_LN     32  /* also write ln_out = LayerNorm(out row) * ln_gamma + ln_beta    */
                           /* (attention.py:199-201,216-219: the norm1/2/3 that follows the  */
                           /* proj_in / to_out linears): only where moca_gemm_ln_ok() != 0   */
#define MOCA_EP_GELU    8  /* out = gelu(acc + bias) (exact erf GELU; 128-row kernel only:   */
                           /* M <= 128 or MOCA_FORCE_SMALL_TILE, splits = 1)                  */
#define MOCA_EP_ROWSUM 64  /* also write per-(column tile, row) sums and sums of squares of   */
                           /* the stored (fp16-rounded) values to p.rowsum: the LayerNorm     */
                           /* statistics of the consumer (attention.py:199-201); only where   */
                           /* moca_gemm_rowsum_cols() > 0                                     */
#define MOCA_EP_GSTAT  256 /* GroupNorm statistics of the consumer, finished: every block adds */
                           /* the sums / sums of squares of what it stores, per (statistics    */
                           /* group, channel group), to p.gstat (i64 fixed point; zero before  */
                           /* launch); moca_groupnorm_gstat_f16 then needs no finalize launch. */
                           /* Same kernels as MOCA_EP_COLSUM (moca_gemm_colsum_rows() > 0) and */
                           /* p.gstat_rows % that == 0                                         */
#define MOCA_EP_TATTN  512 /* the fused to_q|to_k|to_v projection of a TemporalTransformer's    */
                           /* self-attention FOLLOWED BY the attention over the frame axis      */
                           /* (attention.py:92-114 inside :331-352): W rows are packed per head  */
                           /* (64 q, 64 k, 64 v rows), an M tile = the 16 frames of 20 pixels,   */
                           /* the block finishes softmax(q k^T * tattn_scale) v per pixel and    */
                           /* writes only the attention output out[M][ldo >= N/3].  T == 16,     */
                           /* HW % 20 == 0 (moca_gemm_tattn_ok()); may carry MOCA_EP_LNFOLD      */
#define MOCA_EP_LNFOLD 128 /* the A operand is x, the linear wanted is Linear(LayerNorm(x)):  */
                           /* W is packed as W' = W * diag(ln weight), bias as b + W.ln_bias, */
                           /* p.lnf_wsum[n] = sum_k W'[n][k]; the epilogue computes           */
                           /* rstd[m] * (acc[m][n] - mean[m] * wsum[n]) + bias[n] with the    */
                           /* row statistics from p.lnf_part (what a MOCA_EP_ROWSUM producer  */
                           /* left behind): LayerNorm never touches memory                    */
                           /* (attention.py:216-220); only where moca_gemm_lnfold_ok() != 0   */

#define MOCA_EP_SLABS 1024 /* split-K only (splits > 1 after normalisation, no other flag): leave the fp32 partial  */
                           /* slabs in splitk_ws and launch NO reduce -- the caller finishes them, bias / row add /  */
                           /* residual included, with moca_gemm_splitk_groupnorm_f16 (same params)                  */

typedef struct moca_gemm_params {
    const void* a;         /* fp16 activations (gather source)                              */
    const void* w;         /* fp16 weights [N][ldw], K-contiguous, zero padded to ldw       */
    void*       out;       /* [M][ldo] fp16 (or f32 with MOCA_EP_OUT_F32)                   */
    const float* bias;     /* [N] f32 or NULL (packed like W for GEGLU)                     */
    const void* rowadd;    /* fp16 [M/rowadd_div][ld_rowadd] added per row group, or NULL   */
    const void* residual;  /* fp16 [M][ldr] added last, or NULL                             */
    float*      splitk_ws; /* f32 workspace [splits][M][N] when splits > 1                  */
    int32_t M, N, K;       /* logical sizes; N % 64 == 0, K % 8 == 0                        */
    int32_t lda, ldw, ldo, ldr, ld_rowadd;
    int32_t rowadd_div;    /* rows per rowadd group (H*W of the output map)                 */
    int32_t a_mode;
    int32_t C;             /* channels per tap (conv modes), C % 8 == 0                     */
    int32_t inH, inW;      /* stored source map (before the optional x2 upsample)           */
    int32_t outH, outW;    /* output map                                                    */
    int32_t stride;        /* 1 or 2                                                        */
    int32_t up;            /* 1: nearest x2 upsample fused into the gather                  */
    int32_t T;             /* frames per video (A_TCONV3)                                   */
    int32_t HW;            /* pixels per frame (A_TCONV3)                                   */
    int32_t flags;         /* MOCA_EP_*                                                     */
    int32_t splits;        /* split-K factor (>=1)                                          */
    int32_t nopad_lo;      /* A_CONV3X3, stride 2: 1 = no padding at the top/left, one row/column at the
                              bottom/right (F.pad(x,(0,1,0,1)) + conv pad 0: ae_modules.py Downsample.forward);
                              0 = the symmetric padding 1 of every other 3x3 conv                              */
    int32_t prefetch_kib;  /* size of `prefetch` in KiB (0: none) */
    float*      colsum;    /* MOCA_EP_COLSUM: f32 [ceil(M/rows)][N][2] = (sum, sum of squares) over the rows of each
                              row tile (rows = moca_gemm_colsum_rows()), of the values as stored (after bias / row add /
                              residual, before the fp16 rounding)                                                      */
    const float* ln_gamma; /* MOCA_EP_LN: LayerNorm weight / bias [N] f32                                              */
    const float* ln_beta;
    void*       ln_out;    /* MOCA_EP_LN: fp16 [M][ld_ln] second output                                                */
    int32_t     ld_ln;
    float       ln_eps;    /* MOCA_EP_LN / MOCA_EP_LNFOLD: LayerNorm eps                                               */
    float*      rowsum;    /* MOCA_EP_ROWSUM (out): f32 [N / cols][M][2] = (sum, sum of squares) over the columns of each
                              column tile (cols = moca_gemm_rowsum_cols()), of the values as stored                     */
    const float* lnf_part; /* MOCA_EP_LNFOLD (in): f32 [lnf_nparts][M][2] row partial sums of the A operand (K columns) */
    const float* lnf_wsum; /* MOCA_EP_LNFOLD: f32 [N], sum over k of the packed fp16 W'[n][k]                           */
    int32_t     lnf_nparts;
    int32_t     reserved2_;
    int64_t*    gstat;     /* MOCA_EP_GSTAT: i64 [M / gstat_rows][32][2] fixed-point accumulators (sum in 2^-20 units, sum of squares
                              in 2^-12 units: integer atomics, so the result does not depend on the arrival order of the blocks) per
                              (statistics group, GroupNorm channel group of N / 32 columns)                             */
    int32_t     gstat_rows;/* rows per statistics group (frames_per_stat * H*W of the consumer's GroupNorm)             */
    float       tattn_scale;/* MOCA_EP_TATTN: softmax scale (dim_head ** -0.5); frames / pixels per frame in T / HW     */
    const void* prefetch;  /* optional, no effect on results: [prefetch, prefetch + prefetch_kib KiB) -- the weights of the NEXT weight-heavy
                              launch -- is read once by a few EXTRA blocks of this launch's grid (the direct-to-LDS kernels), which land on
                              the CUs the launch leaves idle (200-250 tiles on 256 CUs at the 640- / 1280-channel levels) or on its tail, so
                              that the next launch finds its W in the 256 MB Infinity Cache instead of HBM (inside a replayed forward every
                              weight is a first-touch read: 2.83 GB per forward).  16-byte aligned.                               */
    int32_t     up_phase;  /* MOCA_A_CONV3X3 only.  0: the 3x3 conv.  1 + 2a + b (a, b in {0, 1}): phase (a, b) of `Upsample` =
                              F.interpolate(nearest, x2) + conv3x3 (openaimodel3d.py:96-106) as a 2 x 2 conv on the LOW-resolution grid:
                              out[f][2i+a][2j+b] = bias + sum_{r,s in {0,1}} Wp[r][s] . in[f][i+a-1+r][j+b-1+s], Wp = the 3x3 taps that
                              land on the same input pixel, summed (ops.pack_upconv_phases); K = 4 C, inH x inW = outH x outW = the
                              low-resolution grid, `out` = the [F][2H][2W][N] tensor (rows scattered by the kernel); 4/9 of the FLOPs */
    int32_t     reserved4_;
    const void* a2;        /* MOCA_A_LINEAR only, optional: the A operand is the VIRTUAL torch.cat([a, a2], dim=channels) of
                              openaimodel3d.py:571 -- columns [0, k1) of a row come from a (row stride lda >= k1), columns [k1, K) from a2
                              (fp16 [M][lda2], lda2 >= K - k1); k1 % 64 == 0, (K - k1) % 64 == 0, no split-K; only where
                              moca_gemm_cat_ok() != 0 (the staggered 320 x 160 / 160 x 320 kernels: the skip_connection 1x1 convs of the
                              320- / 640-channel output blocks).  NULL: the plain linear.                                       */
    int32_t     lda2;
    int32_t     k1;
    int32_t     gstat_cpg; /* MOCA_EP_GSTAT: columns per GroupNorm channel group (0 = N / 32) and the channel index of column 0 inside the   */
    int32_t     gstat_coff;/* consumer's tensor (0): a producer whose output is ONE SOURCE of a virtual concat accumulates the statistics of  */
                           /* the concat's groups, group (gstat_coff + n) / gstat_cpg (< 32), straight into the concat's accumulators        */
    int32_t     wgroup_rows;   /* MOCA_A_LINEAR only, 0 = off.  > 0: PER-ROW-GROUP weights -- rows [g R, (g + 1) R), R = wgroup_rows, are    */
    int32_t     wgroup_stride; /* multiplied by the matrix at w + g * wgroup_stride (fp16 elements, >= N * ldw) and take the bias at bias + */
                           /* g * N: `proj_in(GroupNorm(x))` (attention.py:238-242,262-268 / :297-302,333-341) with the GroupNorm -- whose    */
                           /* scale and shift are constant over a (frames_per_stat x H x W) row group -- folded into per-group weights by   */
                           /* moca_groupnorm_fold_weights_f16: the normalised tensor is never written or read.  Only where               */
                           /* moca_gemm_wgroup_ok() != 0 (the staggered kernels, row tiles inside one group)                            */
} moca_gemm_params;

/* Replaces F.conv2d 3x3 (openaimodel3d.py:152,177,66-70,96-106,376,531),
 * nn.Conv3d (3,1,1) (openaimodel3d.py:252-263) and every nn.Linear on the path
 * (attention.py:54-57,242,258,302,328,379,399; openaimodel3d.py:166-172,362-372),
 * with the bias / time-embedding add (openaimodel3d.py:217-226) / residual add
 * (openaimodel3d.py:228,276; attention.py:217-219,278,373) / GEGLU
 * (attention.py:381-383) folded into the epilogue.                              */
int moca_gemm_f16(const moca_gemm_params* p, void* stream);
/* Rows per row tile of the MOCA_EP_COLSUM output for this call (320 / 160 on the staggered kernel, 256 on the 256-row
 * direct-to-LDS kernel), or 0 when this call cannot produce column sums (split-K, GEGLU, fp32 output, small tiles ...).
 * The GroupNorm that consumes the sums (moca_groupnorm_colsum_f16) needs (frames_per_stat * H*W) % rows == 0.        */
int moca_gemm_colsum_rows(const moca_gemm_params* p);
/* 1 when this call can also produce the LayerNorm of its output rows (MOCA_EP_LN): a plain linear with N == 320 (a block of
 * the 160 x 320 tiling owns complete rows), fast gather, no split-K, enough rows to fill the chip; else 0.               */
int moca_gemm_ln_ok(const moca_gemm_params* p);
/* Columns per column tile of the MOCA_EP_ROWSUM output of this call (320 / 160 on the staggered kernel, 128 / 160 on the
 * 256-row kernel; N / cols partial sums per row), or 0 when this call cannot leave row sums behind.                     */
int moca_gemm_rowsum_cols(const moca_gemm_params* p);
/* 1 when this call's kernel has the MOCA_EP_LNFOLD epilogue (a plain or GEGLU linear on the staggered, 256-row or
 * two-blocks-per-CU kernel, fp16 output, no split-K); else 0 (the caller then runs moca_layernorm_f16 first).           */
int moca_gemm_lnfold_ok(const moca_gemm_params* p);
/* 1 when this call can run as MOCA_EP_TATTN (linear, K % 64 == 0, N % 192 == 0, T == 16, HW % 20 == 0, M % (16 HW) == 0,
 * no split-K / residual / row add); else 0 (the caller then runs the projection and moca_temporal_attention_f16).       */
int moca_gemm_tattn_ok(const moca_gemm_params* p);
/* 1 when this call (wgroup_rows / wgroup_stride set) can take per-row-group weights (see moca_gemm_params.wgroup_rows); else 0 (the
 * caller then runs the GroupNorm as a pass of its own).                                                                    */
int moca_gemm_wgroup_ok(const moca_gemm_params* p);
/* 1 when this call (a2 / lda2 / k1 set) can read its A operand from two sources (see moca_gemm_params.a2); else 0 (the caller
 * then materialises the concat with moca_concat_channels*_f16).                                                          */
int moca_gemm_cat_ok(const moca_gemm_params* p);
/* The split-K reduce of a MOCA_EP_SLABS call AND the GroupNorm(32)(+SiLU) that consumes its output (openaimodel3d.py:149-153,173-178,
 * 252-263 at the 5 x 8-latent level, where every conv runs split-K) in one launch: x = fp16(sum of slabs + bias + row add + residual)
 * exactly as moca_gemm_f16 would have stored it, y [M][N] = GroupNorm(x) over frames_per_stat frames of HW rows; x itself is written to
 * p->out only with write_x != 0.  `p` = the params of the MOCA_EP_SLABS call.  Only where moca_gemm_splitk_groupnorm_ok() != 0
 * (a (statistics group, channel group) slab of at most 4096 16-byte chunks, N / 32 a multiple of 8).                               */
int moca_gemm_splitk_groupnorm_ok(const moca_gemm_params* p, int32_t HW, int32_t frames_per_stat);
int moca_gemm_splitk_groupnorm_f16(const moca_gemm_params* p, void* y, const float* gamma, const float* beta, int32_t HW,
                                   int32_t frames_per_stat, float eps, int32_t silu, int32_t write_x, void* stream);
/* bytes of split-K workspace moca_gemm_f16 needs for (M,N,splits) */
int64_t moca_gemm_splitk_ws_bytes(int32_t M, int32_t N, int32_t splits);

/* ---- normalisation --------------------------------------------------------- */
/* GroupNorm(32 groups) over channels-last fp16 x[F][HW][C], statistics over
 * frames_per_stat consecutive frames (1: per-frame 4-D GroupNorm; T: the 5-D
 * GroupNorm of TemporalConvBlock / TemporalTransformer), fp32 statistics, optional
 * SiLU.  ws: f32 workspace of moca_groupnorm_ws_bytes().
 * Replaces GroupNormSpecific (basics.py:76-87), nn.GroupNorm(32,C,eps=1e-6)
 * (attention.py:238,297), nn.GroupNorm(32,C)+SiLU (openaimodel3d.py:253-262).   */
int moca_groupnorm_nhwc_f16(const void* x, void* y, const float* gamma, const float* beta,
                            int32_t F, int32_t HW, int32_t C, int32_t frames_per_stat,
                            float eps, int32_t silu, float* ws, void* stream);
int64_t moca_groupnorm_ws_bytes(int32_t F, int32_t HW, int32_t C);
/* The same GroupNorm when the producer of x was a moca_gemm_f16 call with MOCA_EP_COLSUM: the statistics pass over x
 * is replaced by a reduction of colsum [F*HW/tile_rows][C][2] (no row tile may straddle two statistics groups:
 * (frames_per_stat * HW) % tile_rows == 0); two launches (finalize, apply)
 * instead of three and x is read once.  ws as above.                                                                */
int moca_groupnorm_colsum_f16(const void* x, void* y, const float* gamma, const float* beta, const float* colsum,
                              int32_t tile_rows, int32_t F, int32_t HW, int32_t C, int32_t frames_per_stat,
                              float eps, int32_t silu, float* ws, void* stream);

/* GroupNorm(32, K, affine, no activation) folded into the Linear(K, N) that consumes it -- `x = self.norm(x); ...; x = self.proj_in(x)`
 * of SpatialTransformer / TemporalTransformer (attention.py:238-242,262-268 / :297-302,333-341) -- as PER-STATISTICS-GROUP weights:
 * with mean / rstd of group (sg, k / (K/32)) from the finished statistics gstat i64 [n_sg][32][2] (a MOCA_EP_GSTAT producer; `count` =
 * values per group = frames_per_stat * H*W * K/32), s[k] = gamma[k] rstd, t[k] = beta[k] - mean rstd gamma[k]:
 *     wg[sg][n][k] = fp16(w[n][k] * s[k])            (fp16 [n_sg][N][ldw], zero padding copied),
 *     bg[sg][n]    = bias[n] + sum_k t[k] * w[n][k]    (f32 [n_sg][N]; bias may be NULL),
 * so that Linear(GroupNorm(x)) = x . wg[sg]^T + bg[sg] for the rows of group sg: moca_gemm_f16 with wgroup_rows = rows per group,
 * wgroup_stride = N * ldw.  The normalised tensor is never stored (one fp16 rounding less than the reference's layout would need).
 * A poisoned / out-of-range statistics group yields NaN weights for that group.  K % 32 == 0, K % 8 == 0, ldw % 8 == 0.        */
int moca_groupnorm_fold_weights_f16(const void* w, const float* bias, const float* gamma, const float* beta, const int64_t* gstat,
                                    void* wg, float* bg, int32_t n_sg, int32_t N, int32_t K, int32_t ldw, int64_t count, float eps,
                                    void* stream);

/* The same GroupNorm when the producer of x was a moca_gemm_f16 call with MOCA_EP_GSTAT: gstat i64 [F / frames_per_stat][32][2]
 * holds the finished sums, so this is ONE launch (apply) and x is read once.                                          */
int moca_groupnorm_gstat_f16(const void* x, void* y, const float* gamma, const float* beta, const int64_t* gstat,
                             int32_t F, int32_t HW, int32_t C, int32_t frames_per_stat,
                             float eps, int32_t silu, void* stream);
/* torch.cat([a, b], dim=channels) (openaimodel3d.py:571) of a [F*HW][C1] and b [F*HW][C2] that also ADDS the GroupNorm
 * statistics of its output to gstat (as MOCA_EP_GSTAT; zero before the launch): the ResBlock.in_layers GroupNorm that
 * follows (openaimodel3d.py:149) is then one moca_groupnorm_gstat_f16 launch.                                        */
int moca_concat_channels_gstat_f16(const void* a, const void* b, void* out, int32_t F, int32_t HW, int32_t C1, int32_t C2,
                                   int32_t frames_per_stat, int64_t* gstat, void* stream);
/* The VIRTUAL torch.cat (openaimodel3d.py:571) in front of ResBlock.in_layers[0] (:149): GroupNorm(32)(+SiLU) of cat([a, b], channels),
 * a [F*HW][C1], b [F*HW][C2], y [F*HW][C1+C2] -- the concatenated tensor itself is never written.  Statistics (per-frame or per
 * frames_per_stat frames): gstat_cat i64 [F / frames_per_stat][32][2] in the CONCAT's grouping (groups of (C1+C2)/32 channels) holds
 * the contribution of a (a MOCA_EP_GSTAT producer with gstat_cpg = (C1+C2)/32, or moca_gstat_accum_f16) and, when gstat_b is NULL,
 * that of b as well; else gstat_b i64 [..][32][2] holds b's OWN finished statistics (groups of C2/32 channels: what b's producer
 * left for b's other consumer) and they are merged here: needs ((C1+C2)/32) % (C2/32) == 0 and (C1 % ((C1+C2)/32)) % (C2/32) == 0.
 * Fb (0 = F): gstat_b covers Fb frames and b is F / Fb copies of them (the skip connection out of the shared guidance prefix).   */
int moca_groupnorm_gstat_cat_f16(const void* a, const void* b, void* y, const float* gamma, const float* beta,
                                 const int64_t* gstat_cat, const int64_t* gstat_b, int32_t Fb, int32_t F, int32_t HW, int32_t C1,
                                 int32_t C2, int32_t frames_per_stat, float eps, int32_t silu, void* stream);
/* statistics only: adds the sums / sums of squares of x [F*HW][C] to gstat [F / frames_per_stat][32][2], channel c to group
 * (coff + c) / cpg (< 32) -- the share of one source of a virtual concat whose producer could not leave them (zero before the launch) */
int moca_gstat_accum_f16(const void* x, int32_t F, int32_t HW, int32_t C, int32_t frames_per_stat, int32_t cpg, int32_t coff,
                         int64_t* gstat, void* stream);
/* zero `bytes` bytes at the 16-byte aligned `ptr` on the stream (the MOCA_EP_GSTAT accumulators of a forward are zeroed by one call) */
int moca_memset_zero(void* ptr, int64_t bytes, void* stream);

/* LayerNorm over the last dim of fp16 x[M][C] (eps 1e-5): attention.py:199-201 */
int moca_layernorm_f16(const void* x, void* y, const float* gamma, const float* beta,
                       int32_t M, int32_t C, float eps, void* stream);

/* ---- attention --------------------------------------------------------------- */
/* softmax(Q K^T * scale) V, head dim 64, fp16 in/out, fp32 softmax.
 * q: [Bq][Nq][ldq] with head h at column h*64 of the q block; k, v likewise with
 * row strides ldk, ldv; kv batch index = q batch index / kv_div (cross-attention
 * shares one context per video: openaimodel3d.py:547).  out: [Bq][Nq][ldo].
 * Replaces CrossAttention.forward (attention.py:92-114) in its spatial self /
 * spatial cross roles.                                                           */
int moca_attention_f16(const void* q, const void* k, const void* v, void* out,
                       int32_t Bq, int32_t heads, int32_t Nq, int32_t Nk,
                       int32_t ldq, int32_t ldk, int32_t ldv, int32_t ldo,
                       int32_t kv_div, float scale, void* stream);

/* Causal self-attention, head dim 64: query i attends to keys 0..i (the text tower of the OpenCLIP encoder,
 * condition.py:205-212: `text_transformer_forward(x, attn_mask=self.model.attn_mask)`); q/k/v/out as above, N = Nq = Nk. */
int moca_attention_causal_f16(const void* q, const void* k, const void* v, void* out,
                              int32_t B, int32_t heads, int32_t N, int32_t ldq, int32_t ldk, int32_t ldv, int32_t ldo,
                              float scale, void* stream);

/* Temporal self-attention over the frame axis: for every (video b, pixel p, head h)
 * attend over the T (<=16) frames.  qkv rows are channels-last tokens
 * [(b*T+t)*HW + p][ld]; q/k/v point at their first column.  Replaces
 * CrossAttention.forward inside TemporalTransformer (attention.py:331-352) incl.
 * the (b t) c h w <-> (b h w) t c reshuffles (attention.py:335-338,367).          */
int moca_temporal_attention_f16(const void* q, const void* k, const void* v, void* out,
                                int32_t B, int32_t T, int32_t HW, int32_t heads,
                                int32_t ld_qkv, int32_t ldo, float scale, void* stream);

/* ---- layout / embedding helpers ----------------------------------------------- */
/* x [B][Cin][T][H][W] (f32 or f16) -> channels-last fp16 [B*T][H*W][Cpad], zero padded
 * (openaimodel3d.py:552-554) */
int moca_ncthw_to_nhwc_f16(const void* x, int32_t x_is_f32, void* y, int32_t B, int32_t Cin,
                           int32_t T, int32_t HW, int32_t Cpad, void* stream);
/* channels-last fp16 [B*T][H*W][ld] (first Cout columns) -> [B][Cout][T][H][W] f32/f16
 * (openaimodel3d.py:573-577) */
int moca_nhwc_to_ncthw(const void* y, int32_t ld, void* x, int32_t x_is_f32, int32_t B,
                       int32_t Cout, int32_t T, int32_t HW, void* stream);
/* out[r][0:C1] = a[r], out[r][C1:C1+C2] = b[r]  (torch.cat(dim=1), openaimodel3d.py:571) */
int moca_concat_channels_f16(const void* a, const void* b, void* out, int64_t rows,
                             int32_t C1, int32_t C2, void* stream);
/* dst = `reps` back-to-back copies of src[0:bytes] (bytes % 16 == 0).  The batch of a classifier-free-guidance forward repeats the
 * same latents with different contexts (ddim.py:298-299,366-369: two apply_model calls on the same x): everything before the
 * first cross-attention is computed once, this copy is where the branches start to differ. */
int moca_repeat_f16(const void* src, void* dst, int64_t bytes, int32_t reps, void* stream);
/* sinusoidal embedding [n][dim] fp16 = [cos(t f_k), sin(t f_k)] (utils_diffusion.py:8-28);
 * t is int64 on device */
int moca_timestep_embedding_f16(const int64_t* t, void* out, int32_t n, int32_t dim,
                                float max_period, void* stream);
/* y = silu(x) (+ broadcasting helper for the embedding MLPs): out[i][:] = silu(a[i/div_a] + b[i/div_b]) */
int moca_silu_add_rows_f16(const void* a, int32_t div_a, const void* b, int32_t div_b, void* out,
                           int32_t rows, int32_t C, int32_t apply_silu, void* stream);

/* out[i][:] = table[tokens[i]][:] + pos[i % L][:]  (fp32 tables, fp16 out; condition.py:206-207:
 * `token_embedding(text) + positional_embedding`) */
int moca_embed_tokens_f16(const int64_t* tokens, const float* table, const float* pos, void* out,
                          int32_t n_tokens, int32_t L, int32_t C, int32_t vocab, void* stream);

/* ---- VAE decoder helpers (next row N1: AutoencoderKL.decode, lvdm/models/autoencoder.py:104-107) ---- */
/* out[(b*T+t)*HW+p][co] = bias[co] + sum_ci w[co][ci] * z[b][ci][t][p] * inv_scale  (co < Cout; zero up to Cpad).
 * Replaces `z = 1/scale_factor * z` (ddpm3d.py:559) + `post_quant_conv` (autoencoder.py:105) + the NCHW -> channels-last
 * shuffle in one pass; Cin <= 8, w fp32 [Cout][Cin]. */
int moca_channel_mix_f16(const void* z, int32_t z_is_f32, const float* w, const float* bias, void* out,
                         int32_t B, int32_t Cin, int32_t T, int32_t HW, int32_t Cout, int32_t Cpad,
                         float inv_scale, void* stream);
/* out[n][z][hw] = scale * (mean + exp(0.5 * clamp(logvar, -30, 20)) * noise), moments [n][2z][hw] = (mean | logvar) fp32;
 * noise NULL = the mode.  Replaces DiagonalGaussianDistribution.sample/.mode (lvdm/distributions.py:24-40,66-67) and the
 * `scale_factor *` of get_first_stage_encoding (ddpm3d.py:458-465). */
int moca_gaussian_sample_f32(const float* moments, const float* noise, float* out, int32_t n, int32_t z, int32_t hw,
                             float scale, void* stream);
/* p[r][0:N] = softmax(scale * s[r][0:N]) for R rows, fp32 logits in, fp16 probabilities out; N, lds, ldp % 4 == 0.
 * Replaces `w_ = w_ * c**-0.5; softmax(w_, dim=2)` of AttnBlock.forward (ae_modules.py:66-68). */
int moca_softmax_rows_f16(const float* s, void* p, int64_t R, int32_t N, int64_t lds, int64_t ldp, float scale,
                          void* stream);

/* ---- sampler arithmetic (fp32) -------------------------------------------------- */
/* e = e_u + s (e_c - e_u): ddim.py:304,372 */
int moca_cfg_combine_f32(const float* e_c, const float* e_u, float* out, float scale,
                         int64_t n, void* stream);
/* base DDIM update with use_scale (ddim.py:328-357): per-element
 *   pred_x0 = (x - sqrt(1-a_t) e)/sqrt(a_t) [/ scale_t];
 *   x_prev = sqrt(a_prev) [scale_prev] pred_x0 + sqrt(1-a_prev-sigma^2) e + sigma*noise  */
int moca_ddim_update_f32(const float* x, const float* e, const float* noise, float* x_prev,
                         float* pred_x0, float a_t, float a_prev, float sigma_t,
                         float sqrt_one_minus_at, int32_t use_scale, float scale_t,
                         float scale_prev, int64_t n, void* stream);
/* MoCA FIFO step (ddim.py:405-430,556-609 arithmetic): per-frame coefficients, momentum
 * EMA along the frame axis, x_prev, mask injection into pred_x0, gamma blend.  All latent
 * tensors [B][C][F][HW] f32.  coef[F][6] = {sqrt(a_t), sqrt(a_prev), sigma_t, sqrt(1-a_t),
 * sqrt(1-a_prev-sigma_t^2), 2(1-ts/1000)} evaluated by the host in fp32 exactly as the
 * reference's 0-dim tensors are.  mask [B][1][Fm][HW] or NULL; cond [B][C][HW] or NULL;
 * mask_index[F] = mask frame consulted for frame i (-1: none; the reference's clobbered
 * loop variable makes this ceil(H/4)-1 for i>=1, ddim.py:477-567); enh[F] = factor k
 * multiplying cond (ddim.py:582).  ws: Fm floats of scratch (per-frame mask sums).   */
int moca_fifo_ddim_step_f32(const float* sample, const float* eps, const float* noise,
                            float* momentum, float* x_prev, float* pred_x0,
                            const float* coef, const float* mask, const float* cond,
                            const int32_t* mask_index, const float* enh, float* ws,
                            int32_t B, int32_t C, int32_t F, int32_t Fm, int32_t HW,
                            float beta, float one_minus_beta, float gamma, float one_minus_gamma,
                            void* stream);

/* ---- one outer MoCA-FIFO iteration as device-side work (scripts/evaluation/funcs.py:305-371) --------------------
 * The latent queue [C][Q][HW] f32 (batch 1: the reference's per-frame-timestep path needs B = 1, openaimodel3d.py:535)
 * is a RING: queue frame j lives in slot (head + j) mod Q.  head, the iteration counter and the RNG seed live in a
 * device-resident state block, so the whole iteration is a fixed launch sequence that is captured once into a hipGraph
 * (moca_graph_begin/end) and replayed with no host round trip.                                                        */
typedef struct moca_fifo_state {
    int32_t head;        /* ring slot of queue frame 0 */
    int32_t iter;        /* outer iterations completed (funcs.py:305 `i`) */
    uint32_t seed_lo, seed_hi;
    int32_t ext_noise;   /* 1: the host filled the noise buffer for this iteration (fixtures); cleared by the advance */
    int32_t reserved_[3];
} moca_fifo_state;       /* 32 bytes */

/* out[0:n] ~ N(0,1): Philox4x32-10 keyed by the state's seed, counter (i/4, iter), Box-Muller.  Replaces the
 * torch.randn draws of ddim.py:561 (`noise_like`, per frame) and funcs.py:92 (`new_noise`); no-op when ext_noise. */
int moca_fifo_randn_f32(const moca_fifo_state* state, float* out, int64_t n, void* stream);
/* x[(r nW + w)][c][j][p] = queue frame win_start[w] + j, r < reps (`latents[:,:,start:end].clone()`, funcs.py:315, repeated
 * for the unconditional branch of ddim.py:366-369); anchor[c][p] = queue frame 0 (funcs.py:88), may be NULL.  x == NULL (then
 * nW, win_start are ignored): the anchor only -- the call placed BEHIND the step kernel when the windows rewrite frame 0 (no lookahead,
 * funcs.py:353-354: the reference reads latents[:,:,0] after the write-backs). */
int moca_fifo_gather_windows_f32(const moca_fifo_state* state, const float* queue, float* x, float* anchor,
                                 const int32_t* win_start, int32_t nW, int32_t reps, int32_t C, int32_t Q, int32_t f,
                                 int32_t HW, void* stream);
typedef struct moca_fifo_step_params {
    const moca_fifo_state* state;
    const float* x;            /* [nW][C][f][HW] the windows as gathered BEFORE the iteration */
    const float* eps_c;        /* [nW][C][f][HW] conditional noise prediction */
    const float* eps_u;        /* unconditional one, or NULL (no guidance) */
    const float* noise;        /* [nW][C][f][HW] standard normal (ddim.py:561) */
    float* momentum;           /* [nW][C][f][HW]; frame 0 is read, never written (ddim.py:424) */
    float* queue;              /* ring: x_prev frames wb_from..f-1 of window w -> queue frames win_start[w] + j (funcs.py:351-354); or NULL */
    float* x_prev;             /* [nW][C][f][HW] or NULL */
    float* pred_x0;            /* [nW][C][f][HW] or NULL (the loop discards it, funcs.py:320) */
    const float* coef;         /* [nW][f][6] as moca_fifo_ddim_step_f32 */
    const int32_t* win_start;  /* [nW] first queue frame of each window (funcs.py:307) */
    const float* mask;         /* ring [Q][HW] (DAVIS masks, one channel) or NULL */
    const float* mask_sums;    /* [Q] per-slot sums (ddim.py:585 `mask.sum() != 0`) */
    const int32_t* mask_frame; /* [nW][f] queue frame consulted for window frame j (-1: none; ddim.py:565-567 incl. the clobbered index) */
    const float* enh;          /* [nW][f] factor on cond (ddim.py:582) */
    const float* cond;         /* [C][HW] conditioning image or NULL (zeros, ddim.py:573-574) */
    const float* sam_eff;      /* [nW][f][HW] effective segmentation mask per window frame (moca_sam_select_masks_f32) or NULL; excludes `mask` */
    const int32_t* sam_idx;    /* [nW][f] >= 0: inject cond * 2 where sam_eff > 0.5 (ddim.py:592-606,847,897-901); -1: leave pred_x0 alone */
    float cfg_scale, beta, one_minus_beta, gamma, one_minus_gamma;
    int32_t nW, C, Q, f, HW, wb_from;
} moca_fifo_step_params;
/* e = e_u + s (e_c - e_u) (ddim.py:372) + the MoCA ddim_step arithmetic of moca_fifo_ddim_step_f32 for all windows of an
 * iteration (they are independent: funcs.py:305-355 walks ranks in reverse so every window reads pre-iteration frames) +
 * the write-back of their second halves into the ring. */
int moca_fifo_step_windows_f32(const moca_fifo_step_params* p, void* stream);
/* The mask bookkeeping of `DDIMSampler._apply_segmentation` (ddim.py:739-903) for all windows of an iteration, on candidate masks
 * computed beforehand (the Grounded-SAM-2 producer, ddim.py:745-801, is outside the path).  cand: pool of [HW] f32 masks; window frame
 * (w, i) has ncand[w f + i] candidates starting at pool index cand_off[w f + i] (0 = no box detected, ddim.py:788).  Per window -- one
 * `ddim_step` call, `pre_masks = None` at its start (:391) -- and frame in order: only t_rows[w f + i] <= 300 (:592); no detection ->
 * previous masks, or nothing when there are none (:788-793); mean IoU over zip(new, previous) of the masks binarised at 0.5 (both
 * empty -> 1) < 0.5 -> previous masks (:804-807,905-943); masks applied in order, one whose sum exceeds 0.8 HW resets the frame
 * (:820-822).  eff [nW][f][HW] = the resulting 0/1 mask, eff_idx [nW][f] = i or -1 (no injection); consumed by
 * moca_fifo_step_windows_f32 (sam_eff / sam_idx). */
int moca_sam_select_masks_f32(const float* cand, const int32_t* cand_off, const int32_t* ncand, const int64_t* t_rows, float* eff,
                              int32_t* eff_idx, int32_t nW, int32_t f, int32_t HW, void* stream);
/* funcs.py:357-371 minus the decode: emitted[iter mod n_slots][C][HW] = queue frame emit_frame (may be NULL); the slot of
 * the dequeued frame receives `newframe` [C][HW] (the FreeInit mix, funcs.py:97) and becomes the tail; the mask ring keeps
 * its last frame (funcs.py:113-116); then head = head + 1 mod Q, iter += 1, ext_noise = 0. */
int moca_fifo_advance_f32(moca_fifo_state* state, float* queue, const float* newframe, float* emitted, int32_t n_slots,
                          int32_t emit_frame, float* mask, float* mask_sums, int32_t C, int32_t Q, int32_t HW, void* stream);
/* prepare_latents (funcs.py:53-79): queue[bc][j][p] = coef_z[j] * z[bc][frame_idx[j]][p] + coef_noise[j] * noise[bc][j][p] for the Q queue
 * frames (lookahead copies first); z [BC][Tz][HW], noise / queue [BC][Q][HW] f32; coef_z[j] = alpha_j ** 0.5 and coef_noise[j] =
 * (1 - alpha_j) ** 0.5 evaluated by the host as the reference's fp32 tensors are. */
int moca_fifo_prepare_queue_f32(const float* z, const float* noise, float* queue, const float* coef_z, const float* coef_noise,
                                const int32_t* frame_idx, int32_t BC, int32_t Tz, int32_t Q, int32_t HW, void* stream);
/* ---- one step of base sampling (`ddim_sampling`, ddim.py:226-252) as device-side work: step i = state->iter mod S uses schedule
 * index S - 1 - i (:238), so the captured launch sequence [timestep rows, UNet, this step] replays for every i.
 * moca_base_set_timestep: rows[0:n] = table[S - 1 - i] (`ts = torch.full((b,), step)`, :239; table = ddim_timesteps as int64).
 * moca_base_ddim_step_f32: guidance e_u + s (e_c - e_u) (:304; eps_u NULL: none) + the tail of p_sample_ddim (:328-357) with
 *   coef[idx][8] = {sqrt(a_t), sqrt(a_prev), sigma_t, sqrt(1-a_t), sqrt(1-a_prev-sigma_t^2), scale_t, scale_prev, -} (host,
 *   fp32 like the reference's tensors); x [n] is updated IN PLACE to x_prev, pred_x0 optional; then iter += 1, ext_noise = 0. */
int moca_base_set_timestep(const moca_fifo_state* state, const int64_t* table, int32_t S, int64_t* rows, int32_t n, void* stream);
int moca_base_ddim_step_f32(moca_fifo_state* state, float* x, const float* eps_c, const float* eps_u, const float* noise,
                            float* pred_x0, const float* coef, int32_t S, float cfg_scale, int32_t use_scale, int64_t n, void* stream);
/* sums[fr] = sum of mask frame fr, mask [frames][HW] (ddim.py:585) */
int moca_mask_frame_sums_f32(const float* mask, float* sums, int32_t frames, int32_t HW, void* stream);

/* ---- FreeInit spectral mix (freeinit_utils.py:7-47) ------------------------------- */
/* out = Re ifftn( fftshift^-1( fftshift(fftn x) * LPF + fftshift(fftn n) * (1-LPF) ) )
 * over the last three dims of x,n [C][T][H][W] f32; lpf [T][H][W] f32 (shifted layout,
 * as the reference builds it).  ws: f32 workspace of moca_freq_mix_ws_bytes().        */
int moca_freq_mix_3d_f32(const float* x, const float* noise, const float* lpf, float* out,
                         int32_t C, int32_t T, int32_t H, int32_t W, float* ws, void* stream);
int64_t moca_freq_mix_ws_bytes(int32_t C, int32_t T, int32_t H, int32_t W);
/* closed-form filters (freeinit_utils.py:73-156): type 0 gaussian, 1 butterworth, 2 ideal, 3 box */
int moca_freq_filter_f32(float* lpf, int32_t T, int32_t H, int32_t W, int32_t type, int32_t n,
                         double d_s, double d_t, void* stream);

/* ---- runtime: capture a sequence of launches into a hipGraph and replay it --------- */
int moca_graph_begin(void* stream);
int moca_graph_end(void* stream, void** graph_exec_out);
int moca_graph_launch(void* graph_exec, void* stream);
int moca_graph_destroy(void* graph_exec);
/* own non-blocking stream (so capture never touches the caller's default stream) */
int moca_stream_create(void** stream_out);
int moca_stream_destroy(void* stream);
int moca_stream_sync(void* stream);
/* HIP events on OUR stream (bench.py: torch.cuda.Event only sees torch's stream) */
int moca_event_create(void** ev_out);
int moca_event_record(void* ev, void* stream);
int moca_event_elapsed_ms(void* ev_start, void* ev_stop, float* ms_out);
int moca_event_destroy(void* ev);

/* process-wide KERNEL-CHOICE knobs for tests and A/B runs (they select which kernel runs a shape, never what it computes);
 * value 0..2, 1 = the default rule.  Returns the previous value or MOCA_E_BADARG. */
#define MOCA_TUNE_GEMM_W80   0   /* 320 x 160 staggered kernel: 0 never, 1 where its tiles fill the chip, 2 wherever it applies */
#define MOCA_TUNE_GEMM_G4    1   /* two-blocks-per-CU 256 x 128 kernel: 0 never, 1 GEGLU with K <= 640, 2 wherever it applies   */
#define MOCA_TUNE_GEMM_SQ256 2   /* 256 x 256 staggered kernel: 0 never, 1 wide linears g4 does not take, 2 every wide linear   */
#define MOCA_TUNE_GEMM_WIDE  3   /* 160 x 320 tiling: 0 only with MOCA_EP_LN, 1 linears with N % 320 == 0, 2 convs too          */
#define MOCA_TUNE_GN_SLAB    4   /* single-launch GroupNorm: 0 never, 1 small tensors, 2 always                                 */
#define MOCA_TUNE_GEMM_G4P   5   /* persistent 256 x 128 kernel (two blocks per CU, register epilogue): 0 never, 1 GEGLU linears, 2 every linear it can run */
#define MOCA_TUNE_GEMM_MF32  6   /* MFMA shape of the persistent kernel: 0 v_mfma_f32_16x16x32_f16, 1 v_mfma_f32_32x32x16_f16                          */
#define MOCA_TUNE_GEMM_SQP   7   /* persistent 256 x 256 kernel (register epilogue): 0 never, 1 GEGLU linears, 2 every linear it can run               */
#define MOCA_TUNE_SQP_WALK   8   /* tile walk of the persistent 256 x 256 kernel: 0 strided over the XCD's blocks, 1 a contiguous range per block           */
#define MOCA_TUNE_SLAB_F16   9   /* split-K partial slabs of the 256-row kernel: 0 fp32, 1 fp16 (A/B of VERDICT r4 #5's candidate; changes results within the fp16 tolerance) */
#define MOCA_TUNE_GEMM_WS    10  /* weight-stationary streaming kernel of the 320 -> 320 linears (gemm_ws.hip): 0 never, 1 where it applies                */
#define MOCA_TUNE_COUNT      11
int moca_set_tuning(int32_t knob, int32_t value);
/* Measurement aid, no counterpart in the reference (tools/clock_in_step.py): `blocks` one-wave blocks each record `nsamples` pairs
 * (shader-cycle counter, 100 MHz real-time counter) ~8 us apart into buf = u64 [blocks][nsamples][2] (device memory, zeroed by the
 * caller) while other streams' launches run: the shader clock the chip holds under THAT load, per interval.  Stops early when *stop
 * (device-visible, may be NULL) becomes non-zero.                                                                              */
int moca_debug_clock_sampler(void* buf, int32_t blocks, int32_t nsamples, const int32_t* stop, void* stream);

/* device query: returns 0 and fills name[len] / cu count, or MOCA_E_NODEVICE */
int moca_device_info(char* name, int32_t len, int32_t* cu_count);
const char* moca_version(void);

#ifdef __cplusplus
}
#endif
#endif /* MOCA_HIP_H */
