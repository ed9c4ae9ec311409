// GroupNorm(32) (+SiLU) and LayerNorm on channels-last fp16 activations, fp32/fp64
// statistics.  HBM-bound kernels: every access is a 16-byte (8 x fp16) coalesced
// load/store along the channel axis; partial statistics are deterministic (no atomics).
#include "common.h"

namespace {

constexpr int GN_GROUPS = 32;

__host__ __device__ inline int gn_nchunk(int F, int HW) {
    int n = 2048 / (F > 0 ? F : 1);
    if (n < 1) n = 1;
    const int maxc = (HW + 7) / 8;
    if (n > maxc) n = maxc;
    if (n < 1) n = 1;
    return n;
}

// ---- K1: per (frame, pixel-chunk) partial sums per group ---------------------
// blockDim = (C/8, ppb): thread (cx, py) owns channels [8cx, 8cx+8) of pixels py, py+ppb, ...
__global__ void gn_partial_kernel(const half_t* __restrict__ x, float* __restrict__ partial,
                                  int HW, int C, int nchunk) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* s_sum = reinterpret_cast<float*>(smem_raw);   // [ppb][C]
    float* s_sq = s_sum + blockDim.y * C;                 // [ppb][C]
    const int f = blockIdx.x, chunk = blockIdx.y;
    const int cx = threadIdx.x, py = threadIdx.y, ppb = blockDim.y;
    const int pc = (HW + nchunk - 1) / nchunk;
    const int p_begin = chunk * pc, p_end = min(p_begin + pc, HW);
    float s[8], q[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { s[j] = 0.f; q[j] = 0.f; }
    const half_t* base = x + ((int64_t)f * HW) * C + cx * 8;
    // four pixels per trip: the four 16-byte loads of a thread are in flight together (a one-load-per-trip loop leaves the
    // memory pipe of this streaming pass mostly empty)
    int pp = p_begin + py;
    for (; pp + 3 * ppb < p_end; pp += 4 * ppb) {
        half8v v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const half8v*>(base + (int64_t)(pp + u * ppb) * C);
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float a = (float)v[u][j]; s[j] += a; q[j] += a * a; }
    }
    for (; pp < p_end; pp += ppb) {
        const half8v v = *reinterpret_cast<const half8v*>(base + (int64_t)pp * C);
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float a = (float)v[j]; s[j] += a; q[j] += a * a; }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) { s_sum[py * C + cx * 8 + j] = s[j]; s_sq[py * C + cx * 8 + j] = q[j]; }
    __syncthreads();
    const int tid = py * blockDim.x + cx;
    if (tid < GN_GROUPS) {
        const int cpg = C / GN_GROUPS;
        float a = 0.f, b = 0.f;
        for (int y = 0; y < ppb; ++y)
            for (int c = tid * cpg; c < (tid + 1) * cpg; ++c) { a += s_sum[y * C + c]; b += s_sq[y * C + c]; }
        float* o = partial + (((int64_t)f * nchunk + chunk) * GN_GROUPS + tid) * 2;
        o[0] = a; o[1] = b;
    }
}

// ---- K2: finalize mean / rstd per (stat group, channel group) in fp64 --------
// one block of FIN_T threads per (stat group, channel group): grid (stat groups, 32).  Up to 1280 partial pairs per block at
// the 320-channel level with 16 frames per statistics group: every thread issues all its loads before the first add (a
// one-wavefront loop of dependent L2 round trips took ~10 us there), fp64 accumulation, fixed combine order (deterministic).
constexpr int FIN_T = 256;

__device__ __forceinline__ void gn_finalize_store(double a, double b, float* __restrict__ meanrstd, int64_t slot, double inv_count, float eps) {
    __shared__ double s_a[FIN_T / 64], s_b[FIN_T / 64];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
    if ((threadIdx.x & 63) == 0) { s_a[threadIdx.x >> 6] = a; s_b[threadIdx.x >> 6] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double sa = 0.0, sb = 0.0;
#pragma unroll
        for (int w = 0; w < FIN_T / 64; ++w) { sa += s_a[w]; sb += s_b[w]; }
        const double mean = sa * inv_count;
        double var = sb * inv_count - mean * mean;
        if (var < 0.0) var = 0.0;
        meanrstd[slot * 2] = (float)mean;
        meanrstd[slot * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
    }
}

__global__ __launch_bounds__(FIN_T) void gn_finalize_kernel(const float* __restrict__ partial, float* __restrict__ meanrstd,
                                                            int frames_per_stat, int nchunk, double inv_count, float eps) {
    const int sg = blockIdx.x, g = blockIdx.y, l = threadIdx.x;
    const int n = frames_per_stat * nchunk;
    const float* base = partial + (int64_t)sg * n * GN_GROUPS * 2;
    double a = 0.0, b = 0.0;
    for (int i0 = l; i0 < n; i0 += 4 * FIN_T) {
        float2 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * FIN_T;
            v[u] = i < n ? *reinterpret_cast<const float2*>(base + ((int64_t)i * GN_GROUPS + g) * 2) : float2{0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) { a += (double)v[u].x; b += (double)v[u].y; }
    }
    gn_finalize_store(a, b, meanrstd, (int64_t)sg * GN_GROUPS + g, inv_count, eps);
}

// ---- K2': the same from the column sums a MOCA_EP_COLSUM GEMM left behind: colsum[row tile][C][2], a statistics group
// (sg, g) = tiles [sg*tps, (sg+1)*tps) x channels [g*cpg, (g+1)*cpg) ----
__global__ __launch_bounds__(FIN_T) void gn_finalize_colsum_kernel(const float* __restrict__ colsum, float* __restrict__ meanrstd,
                                                                   int tps, int C, int cpg, double inv_count, float eps) {
    const int sg = blockIdx.x, g = blockIdx.y, l = threadIdx.x;
    const int n = tps * cpg;
    double a = 0.0, b = 0.0;
    for (int i0 = l; i0 < n; i0 += 4 * FIN_T) {
        float2 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * FIN_T;
            const int t = i / cpg, c = g * cpg + (i - t * cpg);
            v[u] = i < n ? *reinterpret_cast<const float2*>(colsum + (((int64_t)sg * tps + t) * C + c) * 2) : float2{0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) { a += (double)v[u].x; b += (double)v[u].y; }
    }
    gn_finalize_store(a, b, meanrstd, (int64_t)sg * GN_GROUPS + g, inv_count, eps);
}

// ---- K3: apply (x - mean) * rstd * gamma + beta, optional SiLU ----------------
// Two batches of four 16-byte loads per thread in flight: the first batch is issued BEFORE the statistics / affine parameters
// are fetched (their latency used to precede the first x load of every block), the next batch before the current one is
// normalised and stored.  Rows past the end of the chunk are fetched from the last valid row (cache hits) and not stored.
__global__ void gn_apply_kernel(const half_t* __restrict__ x, half_t* __restrict__ y,
                                const float* __restrict__ gamma, const float* __restrict__ beta,
                                const float* __restrict__ meanrstd, int HW, int C, int nchunk,
                                int frames_per_stat, int silu) {
    const int f = blockIdx.x, chunk = blockIdx.y;
    const int cx = threadIdx.x, py = threadIdx.y, ppb = blockDim.y;
    const int pc = (HW + nchunk - 1) / nchunk;
    const int p_begin = chunk * pc, p_end = min(p_begin + pc, HW);
    if (p_begin >= p_end) return;
    const int cpg = C / GN_GROUPS;
    const int sg = f / frames_per_stat;
    const int64_t off = ((int64_t)f * HW) * C + cx * 8;
    auto load4 = [&](half8v (&v)[4], int pp) {
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const half8v*>(x + off + (int64_t)min(pp + u * ppb, p_end - 1) * C);
    };
    int pp = p_begin + py;
    half8v cur[4], nxt[4];
    load4(cur, pp);
    float sc[8], sh[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = cx * 8 + j;
        const int g = c / cpg;
        const float mean = meanrstd[((int64_t)sg * GN_GROUPS + g) * 2];
        const float rstd = meanrstd[((int64_t)sg * GN_GROUPS + g) * 2 + 1];
        sc[j] = rstd * gamma[c];
        sh[j] = beta[c] - mean * sc[j];
    }
    auto norm8 = [&](const half8v& v) {
        half8v o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float a = (float)v[j] * sc[j] + sh[j];
            if (silu) a = moca_silu(a);
            o[j] = (half_t)a;
        }
        return o;
    };
    for (; pp < p_end; pp += 4 * ppb) {
        const bool more = pp + 4 * ppb < p_end;
        if (more) load4(nxt, pp + 4 * ppb);
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (pp + u * ppb < p_end) *reinterpret_cast<half8v*>(y + off + (int64_t)(pp + u * ppb) * C) = norm8(cur[u]);
        if (more) {
#pragma unroll
            for (int u = 0; u < 4; ++u) cur[u] = nxt[u];
        }
    }
}

// ---- K3': apply with the statistics a MOCA_EP_GSTAT producer accumulated: gstat i64 fixed point [statistics group][32][2] = (sum, sum of
// squares), finished.  The first 32 threads of a block turn its statistics group's 32 pairs into (mean, rstd) in fp64 while the
// block's first batch of x loads is in flight; no finalize launch exists.
// STREAM: the output is at least half the Infinity Cache (B = 16 forwards at the 320- / 640-channel levels) and leaves with non-temporal
// stores, as the GEMM outputs of that size do (gemm.hip, out_streams)
template <bool STREAM>
__device__ __forceinline__ void gn_st8(half_t* ptr, const half8v v) {
    if constexpr (STREAM) __builtin_nontemporal_store(v, reinterpret_cast<half8v*>(ptr));
    else *reinterpret_cast<half8v*>(ptr) = v;
}
template <bool STREAM>
__global__ void gn_apply_gstat_kernel(const half_t* __restrict__ x, half_t* __restrict__ y,
                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                      const int64_t* __restrict__ gstat, int HW, int C, int nchunk,
                                      int frames_per_stat, double inv_count, float eps, int silu) {
    __shared__ float s_mr[2 * GN_GROUPS];
    const int f = blockIdx.x, chunk = blockIdx.y;
    const int cx = threadIdx.x, py = threadIdx.y, ppb = blockDim.y;
    const int tid = py * blockDim.x + cx;
    const int pc = (HW + nchunk - 1) / nchunk;
    const int p_begin = chunk * pc, p_end = min(p_begin + pc, HW);
    if (p_begin >= p_end) return;
    const int cpg = C / GN_GROUPS;
    const int sg = f / frames_per_stat;
    const int64_t off = ((int64_t)f * HW) * C + cx * 8;
    auto load4 = [&](half8v (&v)[4], int pp) {
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const half8v*>(x + off + (int64_t)min(pp + u * ppb, p_end - 1) * C);
    };
    int pp = p_begin + py;
    half8v cur[4], nxt[4];
    load4(cur, pp);
    // (the affine parameters are fetched BEFORE the statistics hand-off: behind the barrier they were a third dependent round trip of
    //  a block whose whole life is one batch of x -- profiles/r05_ab_gn_apply_prologue.txt)
    const f32x4 gm0 = *reinterpret_cast<const f32x4*>(gamma + cx * 8), gm1 = *reinterpret_cast<const f32x4*>(gamma + cx * 8 + 4);
    const f32x4 bt0 = *reinterpret_cast<const f32x4*>(beta + cx * 8), bt1 = *reinterpret_cast<const f32x4*>(beta + cx * 8 + 4);
    if (tid < GN_GROUPS) {
        const double a = moca_gstat_get(gstat + ((int64_t)sg * GN_GROUPS + tid) * 2, 0), b = moca_gstat_get(gstat + ((int64_t)sg * GN_GROUPS + tid) * 2 + 1, 1);
        const double mean = a * inv_count;
        double var = b * inv_count - mean * mean;
        if (var < 0.0) var = 0.0;
        s_mr[2 * tid] = (float)mean;
        s_mr[2 * tid + 1] = (float)(1.0 / sqrt(var + (double)eps));
    }
    __syncthreads();
    float sc[8], sh[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = cx * 8 + j;
        const int g = c / cpg;
        sc[j] = s_mr[2 * g + 1] * (j < 4 ? gm0[j & 3] : gm1[j & 3]);
        sh[j] = (j < 4 ? bt0[j & 3] : bt1[j & 3]) - s_mr[2 * g] * sc[j];
    }
    auto norm8 = [&](const half8v& v) {
        half8v o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float a = (float)v[j] * sc[j] + sh[j];
            if (silu) a = moca_silu(a);
            o[j] = (half_t)a;
        }
        return o;
    };
    for (; pp < p_end; pp += 4 * ppb) {
        const bool more = pp + 4 * ppb < p_end;
        if (more) load4(nxt, pp + 4 * ppb);
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (pp + u * ppb < p_end) gn_st8<STREAM>(y + off + (int64_t)(pp + u * ppb) * C, norm8(cur[u]));
        if (more) {
#pragma unroll
            for (int u = 0; u < 4; ++u) cur[u] = nxt[u];
        }
    }
}

// ---- torch.cat(dim=channels) that also leaves the GroupNorm statistics of its output behind (openaimodel3d.py:571 followed by
// ResBlock.in_layers[0], :149): the copy is a streaming pass over both inputs anyway; every thread owns 8 fixed channels of
// the concatenated row, accumulates their sums / sums of squares over its pixels, the block combines them per channel group
// through LDS and adds them to gstat (fixed-point atomics, as MOCA_EP_GSTAT) -- the consumer GroupNorm is then one apply launch instead
// of partial + finalize + apply.  blockDim = (C/8, ppb), grid (F, nchunk) as the GroupNorm passes. ----
template <bool STREAM>
__global__ void concat_gstat_kernel(const half_t* __restrict__ a, const half_t* __restrict__ b, half_t* __restrict__ out,
                                    int64_t* __restrict__ gstat, int HW, int C1, int C2, int nchunk, int frames_per_stat) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int C = C1 + C2;
    float* s_sum = reinterpret_cast<float*>(smem_raw);   // [ppb][C]
    float* s_sq = s_sum + blockDim.y * C;                 // [ppb][C]
    const int f = blockIdx.x, chunk = blockIdx.y;
    const int cx = threadIdx.x, py = threadIdx.y, ppb = blockDim.y;
    const int pc = (HW + nchunk - 1) / nchunk;
    const int p_begin = chunk * pc, p_end = min(p_begin + pc, HW);
    const bool from_a = cx * 8 < C1;
    const half_t* src = from_a ? a + ((int64_t)f * HW) * C1 + cx * 8 : b + ((int64_t)f * HW) * C2 + (cx * 8 - C1);
    const int ld = from_a ? C1 : C2;
    half_t* dst = out + ((int64_t)f * HW) * C + cx * 8;
    float s[8], q[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { s[j] = 0.f; q[j] = 0.f; }
    int pp = p_begin + py;
    for (; pp + 3 * ppb < p_end; pp += 4 * ppb) {
        half8v v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const half8v*>(src + (int64_t)(pp + u * ppb) * ld);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            gn_st8<STREAM>(dst + (int64_t)(pp + u * ppb) * C, v[u]);
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float t = (float)v[u][j]; s[j] += t; q[j] += t * t; }
        }
    }
    for (; pp < p_end; pp += ppb) {
        const half8v v = *reinterpret_cast<const half8v*>(src + (int64_t)pp * ld);
        gn_st8<STREAM>(dst + (int64_t)pp * C, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float t = (float)v[j]; s[j] += t; q[j] += t * t; }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) { s_sum[py * C + cx * 8 + j] = s[j]; s_sq[py * C + cx * 8 + j] = q[j]; }
    __syncthreads();
    const int tid = py * blockDim.x + cx;
    if (tid < 2 * GN_GROUPS) {
        const int g = tid >> 1, comp = tid & 1;
        const int cpg = C / GN_GROUPS;
        const float* base = comp ? s_sq : s_sum;
        float t = 0.f;
        for (int y = 0; y < ppb; ++y)
            for (int c = g * cpg; c < (g + 1) * cpg; ++c) t += base[y * C + c];
        moca_gstat_add(gstat + ((int64_t)(f / frames_per_stat) * GN_GROUPS + g) * 2 + comp, comp, t);
    }
}

// ---- the VIRTUAL torch.cat (openaimodel3d.py:571) in front of ResBlock.in_layers[0]: GroupNorm(+SiLU) of cat([a, b], channels) reading a
// and b where their producers left them; only the normalised concat is written.  Thread cx owns 8 channels of the concatenated row (C1 % 8
// == 0: from one source), as in concat_gstat_kernel.  Statistics: gstat_cat (the concat's grouping) holds a's share -- accumulated by a's
// producer (MOCA_EP_GSTAT with gstat_cpg = C / 32) or by gstat_accum_kernel -- and either b's share too (gstat_b == NULL) or b's OWN
// finished statistics are merged here (groups of cpg2 = C2 / 32 channels: cpg % cpg2 == 0 and (C1 % cpg) % cpg2 == 0, host-checked, so
// every group of b lies inside one group of the concat). ----
template <bool STREAM>
__global__ void gn_apply_gstat_cat_kernel(const half_t* __restrict__ a, const half_t* __restrict__ b, half_t* __restrict__ y,
                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                          const int64_t* __restrict__ gstat_cat, const int64_t* __restrict__ gstat_b,
                                          int HW, int C1, int C2, int nchunk, int frames_per_stat, int sgb_mod, double inv_count, float eps, int silu) {
    __shared__ float s_mr[2 * GN_GROUPS];
    const int C = C1 + C2;
    const int f = blockIdx.x, chunk = blockIdx.y;
    const int cx = threadIdx.x, py = threadIdx.y, ppb = blockDim.y;
    const int tid = py * blockDim.x + cx;
    const int pc = (HW + nchunk - 1) / nchunk;
    const int p_begin = chunk * pc, p_end = min(p_begin + pc, HW);
    if (p_begin >= p_end) return;
    const int cpg = C / GN_GROUPS;
    const int sg = f / frames_per_stat;
    const bool from_a = cx * 8 < C1;
    const half_t* src = from_a ? a + ((int64_t)f * HW) * C1 + cx * 8 : b + ((int64_t)f * HW) * C2 + (cx * 8 - C1);
    const int ld = from_a ? C1 : C2;
    const int64_t off = ((int64_t)f * HW) * C + cx * 8;
    auto load4 = [&](half8v (&v)[4], int pp) {
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const half8v*>(src + (int64_t)min(pp + u * ppb, p_end - 1) * ld);
    };
    int pp = p_begin + py;
    half8v cur[4], nxt[4];
    load4(cur, pp);
    const f32x4 gm0 = *reinterpret_cast<const f32x4*>(gamma + cx * 8), gm1 = *reinterpret_cast<const f32x4*>(gamma + cx * 8 + 4);
    const f32x4 bt0 = *reinterpret_cast<const f32x4*>(beta + cx * 8), bt1 = *reinterpret_cast<const f32x4*>(beta + cx * 8 + 4);
    if (tid < GN_GROUPS) {
        const int64_t* ga = gstat_cat + ((int64_t)sg * GN_GROUPS + tid) * 2;
        double sa = moca_gstat_get(ga, 0), sq = moca_gstat_get(ga + 1, 1);
        if (gstat_b) {
            const int cpg2 = C2 / GN_GROUPS;
            const int lo = max(tid * cpg - C1, 0), hi = min((tid + 1) * cpg - C1, C2);     // b's channels inside concat group tid
            for (int j = lo / cpg2; j * cpg2 < hi; ++j) {
                const int64_t* gb = gstat_b + ((int64_t)(sg % sgb_mod) * GN_GROUPS + j) * 2;   // (b = `reps` copies of Fb frames: the shared prefix)
                sa += moca_gstat_get(gb, 0); sq += moca_gstat_get(gb + 1, 1);
            }
        }
        const double mean = sa * inv_count;
        double var = sq * inv_count - mean * mean;
        if (var < 0.0) var = 0.0;
        s_mr[2 * tid] = (float)mean;
        s_mr[2 * tid + 1] = (float)(1.0 / sqrt(var + (double)eps));
    }
    __syncthreads();
    float sc[8], sh[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = cx * 8 + j;
        const int g = c / cpg;
        sc[j] = s_mr[2 * g + 1] * (j < 4 ? gm0[j & 3] : gm1[j & 3]);
        sh[j] = (j < 4 ? bt0[j & 3] : bt1[j & 3]) - s_mr[2 * g] * sc[j];
    }
    auto norm8 = [&](const half8v& v) {
        half8v o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float t = (float)v[j] * sc[j] + sh[j];
            if (silu) t = moca_silu(t);
            o[j] = (half_t)t;
        }
        return o;
    };
    for (; pp < p_end; pp += 4 * ppb) {
        const bool more = pp + 4 * ppb < p_end;
        if (more) load4(nxt, pp + 4 * ppb);
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (pp + u * ppb < p_end) gn_st8<STREAM>(y + off + (int64_t)(pp + u * ppb) * C, norm8(cur[u]));
        if (more) {
#pragma unroll
            for (int u = 0; u < 4; ++u) cur[u] = nxt[u];
        }
    }
}

// ---- statistics only: the share of ONE source x [F][HW][C] of a virtual concat in the concat's GroupNorm statistics, for a source whose
// producer could not leave them (an `Upsample`, the repeat of the shared prefix): channel c counts for group (coff + c) / cpg of
// gstat [statistics group][32][2].  A read-only pass (half the traffic of the copy it replaces); accumulation as concat_gstat_kernel. ----
__global__ void gstat_accum_kernel(const half_t* __restrict__ x, int64_t* __restrict__ gstat, int HW, int C, int nchunk,
                                   int frames_per_stat, int cpg, int coff) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* s_sum = reinterpret_cast<float*>(smem_raw);   // [ppb][C]
    float* s_sq = s_sum + blockDim.y * C;                 // [ppb][C]
    const int f = blockIdx.x, chunk = blockIdx.y;
    const int cx = threadIdx.x, py = threadIdx.y, ppb = blockDim.y;
    const int pc = (HW + nchunk - 1) / nchunk;
    const int p_begin = chunk * pc, p_end = min(p_begin + pc, HW);
    const half_t* src = x + ((int64_t)f * HW) * C + cx * 8;
    float s[8], q[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { s[j] = 0.f; q[j] = 0.f; }
    int pp = p_begin + py;
    for (; pp + 3 * ppb < p_end; pp += 4 * ppb) {
        half8v v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const half8v*>(src + (int64_t)(pp + u * ppb) * C);
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float t = (float)v[u][j]; s[j] += t; q[j] += t * t; }
    }
    for (; pp < p_end; pp += ppb) {
        const half8v v = *reinterpret_cast<const half8v*>(src + (int64_t)pp * C);
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float t = (float)v[j]; s[j] += t; q[j] += t * t; }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) { s_sum[py * C + cx * 8 + j] = s[j]; s_sq[py * C + cx * 8 + j] = q[j]; }
    __syncthreads();
    const int tid = py * blockDim.x + cx;
    const int g0 = coff / cpg, g1 = (coff + C - 1) / cpg;
    if (tid < 2 * (g1 - g0 + 1)) {
        const int g = g0 + (tid >> 1), comp = tid & 1;
        const int c0 = max(g * cpg - coff, 0), c1 = min((g + 1) * cpg - coff, C);
        const float* base = comp ? s_sq : s_sum;
        float t = 0.f;
        for (int yy = 0; yy < ppb; ++yy)
            for (int c = c0; c < c1; ++c) t += base[yy * C + c];
        moca_gstat_add(gstat + ((int64_t)(f / frames_per_stat) * GN_GROUPS + g) * 2 + comp, comp, t);
    }
}

// ---- single-launch GroupNorm for small tensors ------------------------------------
// A statistics slab = (statistics group sg, channel group g): R = frames_per_stat*HW consecutive rows x cpg channels.
// S blocks share one slab: each of them reduces the WHOLE slab (redundantly -- a slab is at most a few hundred KB and
// its S blocks are placed on the same XCD, so the repeats are L2 hits) and then normalises its own 1/S of the rows.
// No inter-block hand-off, so the three launches of the streaming path become one and x is fetched from HBM once.
// Used when the tensor is small enough that the streaming path is bound by its launches rather than by HBM.
template <int VEC>
__global__ __launch_bounds__(256) void gn_slab_kernel(const half_t* __restrict__ x, half_t* __restrict__ y,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      int R, int C, int cpg, int S, int xcd_map, double inv_count,
                                                      float eps, int silu) {
    typedef _Float16 vec_t __attribute__((ext_vector_type(VEC)));
    __shared__ float s_red[2][4];
    __shared__ float s_sc[128], s_sh[128];               // cpg <= 128
    const int tid = threadIdx.x, L = blockIdx.x;
    int slab, split;
    if (xcd_map) { const int w = L >> 3; slab = (L & 7) + 8 * (w / S); split = w % S; }   // consecutive ids go round-robin over the 8 XCDs
    else { slab = L / S; split = L % S; }
    const int sg = slab / GN_GROUPS, g = slab % GN_GROUPS;
    const int vpr = cpg / VEC;                            // vectors per row
    const int dr = 256 / vpr, dv = 256 % vpr;             // advancing a flat index by 256 = dr rows + dv vectors
    const int64_t base = (int64_t)sg * R * C + g * cpg;
    const half_t* xs = x + base;
    const float gm_t = tid < cpg ? gamma[g * cpg + tid] : 0.f, bt_t = tid < cpg ? beta[g * cpg + tid] : 0.f;   // (in flight under the statistics pass)
    // ---- pass 1: sum / sum of squares of the whole slab ----
    float s = 0.f, q = 0.f;
    {
        int row = tid / vpr, v = tid % vpr;
#pragma unroll 4
        for (; row < R;) {
            const vec_t d = *reinterpret_cast<const vec_t*>(xs + (int64_t)row * C + v * VEC);
#pragma unroll
            for (int j = 0; j < VEC; ++j) { const float a = (float)d[j]; s += a; q += a * a; }
            row += dr; v += dv;
            if (v >= vpr) { v -= vpr; ++row; }
        }
    }
    s = wave_sum(s); q = wave_sum(q);
    if ((tid & 63) == 0) { s_red[0][tid >> 6] = s; s_red[1][tid >> 6] = q; }
    __syncthreads();
    if (tid < cpg) {
        const double a = (double)s_red[0][0] + (double)s_red[0][1] + (double)s_red[0][2] + (double)s_red[0][3];
        const double b = (double)s_red[1][0] + (double)s_red[1][1] + (double)s_red[1][2] + (double)s_red[1][3];
        const double mean = a * inv_count;
        double var = b * inv_count - mean * mean;
        if (var < 0.0) var = 0.0;
        const float rstd = (float)(1.0 / sqrt(var + (double)eps));
        const float sc = rstd * gm_t;
        s_sc[tid] = sc;
        s_sh[tid] = bt_t - (float)mean * sc;
    }
    __syncthreads();
    // ---- pass 2: normalise rows [r0, r1) of the slab ----
    const int rps = (R + S - 1) / S;
    const int r0 = split * rps, r1 = min(R, r0 + rps);
    half_t* ys = y + base;
    {
        int row = r0 + tid / vpr, v = tid % vpr;
#pragma unroll 2
        for (; row < r1;) {
            const int64_t o = (int64_t)row * C + v * VEC;
            const vec_t d = *reinterpret_cast<const vec_t*>(xs + o);
            vec_t r;
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                float a = (float)d[j] * s_sc[v * VEC + j] + s_sh[v * VEC + j];
                if (silu) a = moca_silu(a);
                r[j] = (half_t)a;
            }
            *reinterpret_cast<vec_t*>(ys + o) = r;
            row += dr; v += dv;
            if (v >= vpr) { v -= vpr; ++row; }
        }
    }
}

// ---- single-launch GroupNorm with the slab in REGISTERS: one block per (statistics group, channel group) slab of at most 4096
// 16-byte chunks (64 KiB: the 1280-channel levels -- 640 rows x 80 B with 16-frame statistics at 5 x 8 latents, 160 rows x 80 B per
// frame at 10 x 16), every thread keeps its <= 4 chunks, the block reduces (wave shuffles + LDS), normalises from registers and
// stores: x is read ONCE and nothing is reduced redundantly (gn_slab_kernel: S = 16 blocks each re-reduce the whole slab).
template <int CPT>
__global__ __launch_bounds__(1024) void gn_slab_reg_kernel(const half_t* __restrict__ x, half_t* __restrict__ y,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            int R, int C, int cpg, double inv_count, float eps, int silu) {
    __shared__ float s_red[2][16];
    __shared__ float s_sc[128], s_sh[128];
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int slab = blockIdx.x, sg = slab / GN_GROUPS, g = slab % GN_GROUPS;
    const int vpr = cpg / 8, nchunks = R * vpr;
    const int64_t base = (int64_t)sg * R * C + g * cpg;
    const float gm_t = tid < cpg ? gamma[g * cpg + tid] : 0.f, bt_t = tid < cpg ? beta[g * cpg + tid] : 0.f;   // (in flight under the slab's loads)
    half8v v[CPT];
    int off[CPT];
    float s = 0.f, q = 0.f;
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
        const int idx = tid + i * nthr;
        off[i] = -1;
        if (idx < nchunks) {
            const int row = idx / vpr, c = idx - row * vpr;
            off[i] = row * C + c * 8;
            v[i] = *reinterpret_cast<const half8v*>(x + base + off[i]);
        }
    }
#pragma unroll
    for (int i = 0; i < CPT; ++i)
        if (off[i] >= 0) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float a = (float)v[i][j]; s += a; q += a * a; }
        }
    s = wave_sum(s); q = wave_sum(q);
    if ((tid & 63) == 0) { s_red[0][tid >> 6] = s; s_red[1][tid >> 6] = q; }
    __syncthreads();
    if (tid < cpg) {
        double a = 0.0, b = 0.0;
        for (int w = 0; w < (nthr >> 6); ++w) { a += (double)s_red[0][w]; b += (double)s_red[1][w]; }
        const double mean = a * inv_count;
        double var = b * inv_count - mean * mean;
        if (var < 0.0) var = 0.0;
        const float rstd = (float)(1.0 / sqrt(var + (double)eps));
        const float sc = rstd * gm_t;
        s_sc[tid] = sc;
        s_sh[tid] = bt_t - (float)mean * sc;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < CPT; ++i)
        if (off[i] >= 0) {
            const int c0 = (off[i] % C);                 // = c * 8 (row * C is a multiple of C)
            half8v r;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float a = (float)v[i][j] * s_sc[c0 + j] + s_sh[c0 + j];
                if (silu) a = moca_silu(a);
                r[j] = (half_t)a;
            }
            *reinterpret_cast<half8v*>(y + base + off[i]) = r;
        }
}

// ---- LayerNorm: one wavefront per row, row kept in registers; every wave keeps LN_ROWS rows in flight ------
// (one row per wave leaves a CU with ~20 KB of loads in flight -- 3.6 TB/s chip-wide at C = 320; with four rows per
//  wave the loads of all four are issued before the first reduction)
template <int MAXCH, int LN_ROWS>  // max 16-byte chunks per lane; rows per wave (4 when there are enough rows to fill the chip)
__global__ __launch_bounds__(256) void layernorm_kernel(const half_t* __restrict__ x, half_t* __restrict__ y,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        int M, int C, float eps) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row0 = (blockIdx.x * 4 + wave) * LN_ROWS;
    if (row0 >= M) return;
    const int nch = C / 8;
    half8v v[LN_ROWS][MAXCH];
#pragma unroll
    for (int r = 0; r < LN_ROWS; ++r) {
        const int row = min(row0 + r, M - 1);            // (rows past the end repeat the last row; their store is skipped)
        const half_t* xr = x + (int64_t)row * C;
#pragma unroll
        for (int i = 0; i < MAXCH; ++i) {
            const int ch = lane + 64 * i;
            if (ch < nch) v[r][i] = *reinterpret_cast<const half8v*>(xr + ch * 8);
        }
    }
    float mean[LN_ROWS], rstd[LN_ROWS];
#pragma unroll
    for (int r = 0; r < LN_ROWS; ++r) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < MAXCH; ++i) {
            const int ch = lane + 64 * i;
            if (ch < nch) {
#pragma unroll
                for (int j = 0; j < 8; ++j) s += (float)v[r][i][j];
            }
        }
        s = wave_sum(s);
        mean[r] = s / (float)C;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < MAXCH; ++i) {
            const int ch = lane + 64 * i;
            if (ch < nch) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { const float d = (float)v[r][i][j] - mean[r]; q += d * d; }
            }
        }
        q = wave_sum(q);
        rstd[r] = rsqrtf(q / (float)C + eps);
    }
#pragma unroll
    for (int i = 0; i < MAXCH; ++i) {
        const int ch = lane + 64 * i;
        if (ch < nch) {
            const f32x4 g0 = *reinterpret_cast<const f32x4*>(gamma + ch * 8), g1 = *reinterpret_cast<const f32x4*>(gamma + ch * 8 + 4);
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(beta + ch * 8), b1 = *reinterpret_cast<const f32x4*>(beta + ch * 8 + 4);
#pragma unroll
            for (int r = 0; r < LN_ROWS; ++r) {
                if (row0 + r < M) {
                    half8v o;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        o[j] = (half_t)(((float)v[r][i][j] - mean[r]) * rstd[r] * g0[j] + b0[j]);
                        o[4 + j] = (half_t)(((float)v[r][i][4 + j] - mean[r]) * rstd[r] * g1[j] + b1[j]);
                    }
                    *reinterpret_cast<half8v*>(y + (int64_t)(row0 + r) * C + ch * 8) = o;
                }
            }
        }
    }
}

}  // namespace

extern "C" int64_t moca_groupnorm_ws_bytes(int32_t F, int32_t HW, int32_t C) {
    (void)C;
    const int nchunk = gn_nchunk(F, HW);
    return ((int64_t)F * nchunk * GN_GROUPS * 2 + (int64_t)F * GN_GROUPS * 2) * 4;
}

extern "C" int moca_groupnorm_nhwc_f16(const void* x, void* y, const float* gamma, const float* beta,
                                       int32_t F, int32_t HW, int32_t C, int32_t frames_per_stat,
                                       float eps, int32_t silu, float* ws, void* stream) {
    if (!x || !y || !gamma || !beta || !ws) return MOCA_E_BADARG;
    if (F <= 0 || HW <= 0 || C <= 0 || C % (8 * 1) || C % GN_GROUPS || frames_per_stat <= 0 || F % frames_per_stat) return MOCA_E_BADARG;
    const int nch8 = C / 8;
    if (nch8 > 1024) return MOCA_E_BADARG;
    int ppb = 256 / nch8;
    if (ppb < 1) ppb = 1;
    const int nchunk = gn_nchunk(F, HW);
    float* partial = ws;
    float* meanrstd = ws + (int64_t)F * nchunk * GN_GROUPS * 2;
    hipStream_t st = moca_stream(stream);
    const double inv_count = 1.0 / ((double)frames_per_stat * HW * (C / GN_GROUPS));
    {   // small tensors: one launch (gn_slab_kernel)
        const int slab_mode = moca_tuning_get(MOCA_TUNE_GN_SLAB);   // 0: always the streaming path, 2: always the slab path (tests)
        const int cpg = C / GN_GROUPS;
        const int64_t bytes = (int64_t)F * HW * C * 2;
        const int64_t R64 = (int64_t)frames_per_stat * HW;
        if (slab_mode && cpg % 2 == 0 && cpg <= 128 && R64 < (1 << 24) &&
            (slab_mode == 2 || bytes <= (8 << 20) || (frames_per_stat == 1 && HW <= 160 && bytes <= (28 << 20)))) {
            // (measured on the bench workload, profiles/r01_plan_profile_*.txt: the slab kernel's 20..160-byte row segments
            //  lose to the streaming path's full-row reads as soon as the tensor is more than a few MB)
            const int R = (int)R64;
            const int n_slabs = (F / frames_per_stat) * GN_GROUPS;
            {   // slab in registers: <= 4096 chunks of 16 B, rows of >= 160 B ... or many rows
                const int nchunks = R * (cpg / 8);
                if (cpg % 8 == 0 && nchunks <= 4096 && nchunks >= 256) {
                    const int cpt = (nchunks + 1023) / 1024;
                    const int thr = ((nchunks + cpt - 1) / cpt + 63) / 64 * 64;
                    const half_t* xi = reinterpret_cast<const half_t*>(x);
                    half_t* yo = reinterpret_cast<half_t*>(y);
#define MOCA_SLABREG(CPT) hipLaunchKernelGGL(gn_slab_reg_kernel<CPT>, dim3(n_slabs), dim3(thr), 0, st, xi, yo, gamma, beta, R, C, cpg, inv_count, eps, silu)
                    if (cpt == 1) MOCA_SLABREG(1); else if (cpt == 2) MOCA_SLABREG(2); else if (cpt == 3) MOCA_SLABREG(3); else MOCA_SLABREG(4);
#undef MOCA_SLABREG
                    MOCA_CHECK_LAUNCH();
                    return MOCA_OK;
                }
            }
            int S = 1024 / n_slabs;
            if (S > 16) S = 16;
            if (S > R) S = R;
            if (S < 1) S = 1;
            const int xcd_map = (n_slabs % 8 == 0) ? 1 : 0;
            const dim3 g1(n_slabs * S), b1(256);
            const half_t* xi = reinterpret_cast<const half_t*>(x);
            half_t* yo = reinterpret_cast<half_t*>(y);
            if (cpg % 8 == 0)
                hipLaunchKernelGGL(gn_slab_kernel<8>, g1, b1, 0, st, xi, yo, gamma, beta, R, C, cpg, S, xcd_map, inv_count, eps, silu);
            else if (cpg % 4 == 0)
                hipLaunchKernelGGL(gn_slab_kernel<4>, g1, b1, 0, st, xi, yo, gamma, beta, R, C, cpg, S, xcd_map, inv_count, eps, silu);
            else
                hipLaunchKernelGGL(gn_slab_kernel<2>, g1, b1, 0, st, xi, yo, gamma, beta, R, C, cpg, S, xcd_map, inv_count, eps, silu);
            MOCA_CHECK_LAUNCH();
            return MOCA_OK;
        }
    }
    const dim3 grid(F, nchunk), block(nch8, ppb);
    const size_t lds = (size_t)ppb * C * 2 * sizeof(float);
    hipLaunchKernelGGL(gn_partial_kernel, grid, block, lds, st, reinterpret_cast<const half_t*>(x), partial, HW, C, nchunk);
    MOCA_CHECK_LAUNCH();
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(F / frames_per_stat, GN_GROUPS), dim3(FIN_T), 0, st, partial, meanrstd,
                       frames_per_stat, nchunk, inv_count, eps);
    MOCA_CHECK_LAUNCH();
    hipLaunchKernelGGL(gn_apply_kernel, grid, block, 0, st, reinterpret_cast<const half_t*>(x), reinterpret_cast<half_t*>(y),
                       gamma, beta, meanrstd, HW, C, nchunk, frames_per_stat, silu);
    MOCA_CHECK_LAUNCH();
    return MOCA_OK;
}

extern "C" int moca_groupnorm_colsum_f16(const void* x, void* y, const float* gamma, const float* beta, const float* colsum,
                                         int32_t tile_rows, int32_t F, int32_t HW, int32_t C, int32_t frames_per_stat,
                                         float eps, int32_t silu, float* ws, void* stream) {
    if (!x || !y || !gamma || !beta || !ws || !colsum) return MOCA_E_BADARG;
    if (F <= 0 || HW <= 0 || C <= 0 || C % 8 || C % GN_GROUPS || frames_per_stat <= 0 || F % frames_per_stat) return MOCA_E_BADARG;
    if (tile_rows <= 0 || ((int64_t)frames_per_stat * HW) % tile_rows) return MOCA_E_BADARG;   // a row tile must not straddle two statistics groups
    const int nch8 = C / 8;
    if (nch8 > 1024) return MOCA_E_BADARG;
    int ppb = 256 / nch8;
    if (ppb < 1) ppb = 1;
    const int nchunk = gn_nchunk(F, HW);
    float* meanrstd = ws + (int64_t)F * nchunk * GN_GROUPS * 2;         // same workspace layout as the three-launch path
    hipStream_t st = moca_stream(stream);
    const double inv_count = 1.0 / ((double)frames_per_stat * HW * (C / GN_GROUPS));
    const int tps = (int)(((int64_t)frames_per_stat * HW) / tile_rows);   // row tiles per statistics group
    hipLaunchKernelGGL(gn_finalize_colsum_kernel, dim3(F / frames_per_stat, GN_GROUPS), dim3(FIN_T), 0, st, colsum, meanrstd,
                       tps, C, C / GN_GROUPS, inv_count, eps);
    MOCA_CHECK_LAUNCH();
    const dim3 grid(F, nchunk), block(nch8, ppb);
    hipLaunchKernelGGL(gn_apply_kernel, grid, block, 0, st, reinterpret_cast<const half_t*>(x), reinterpret_cast<half_t*>(y),
                       gamma, beta, meanrstd, HW, C, nchunk, frames_per_stat, silu);
    MOCA_CHECK_LAUNCH();
    return MOCA_OK;
}

extern "C" int moca_groupnorm_gstat_f16(const void* x, void* y, const float* gamma, const float* beta, const int64_t* gstat,
                                        int32_t F, int32_t HW, int32_t C, int32_t frames_per_stat,
                                        float eps, int32_t silu, void* stream) {
    if (!x || !y || !gamma || !beta || !gstat) return MOCA_E_BADARG;
    if ((reinterpret_cast<uintptr_t>(gamma) | reinterpret_cast<uintptr_t>(beta)) & 15) return MOCA_E_BADARG;     // (fetched as 16-byte vectors)
    if (F <= 0 || HW <= 0 || C <= 0 || C % 8 || C % GN_GROUPS || frames_per_stat <= 0 || F % frames_per_stat) return MOCA_E_BADARG;
    const int nch8 = C / 8;
    if (nch8 > 1024) return MOCA_E_BADARG;
    int ppb = 256 / nch8;
    if (ppb < 1) ppb = 1;
    if (nch8 * ppb < GN_GROUPS) return MOCA_E_BADARG;                     // (the first 32 threads of a block finish the statistics)
    const int nchunk = gn_nchunk(F, HW);
    const double inv_count = 1.0 / ((double)frames_per_stat * HW * (C / GN_GROUPS));
    const dim3 grid(F, nchunk), block(nch8, ppb);
    const bool streams = (int64_t)F * HW * C * 2 >= (128ll << 20);
    hipLaunchKernelGGL(streams ? gn_apply_gstat_kernel<true> : gn_apply_gstat_kernel<false>, grid, block, 0, moca_stream(stream),
                       reinterpret_cast<const half_t*>(x), reinterpret_cast<half_t*>(y), gamma, beta, gstat, HW, C, nchunk, frames_per_stat,
                       inv_count, eps, silu);
    MOCA_CHECK_LAUNCH();
    return MOCA_OK;
}

// ---- GroupNorm folded into the consuming linear as per-statistics-group weights (moca_groupnorm_fold_weights_f16) ----
// block (sg, row block of 16 W rows): the 2 x K scale / shift values of the group in LDS, then thread = one 8-element chunk of a W row:
// scaled copy + the dot product with the shift (reduced over the row's chunks through LDS in a fixed order: deterministic)
__global__ __launch_bounds__(256) void gn_fold_weights_kernel(const half_t* __restrict__ w, const float* __restrict__ bias,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              const int64_t* __restrict__ gstat, half_t* __restrict__ wg, float* __restrict__ bg,
                                                              int N, int K, int ldw, double inv_count, float eps) {
    extern __shared__ float s_st[];                       // [2][K] scale, shift; then [rows][chunks] partial dots
    const int sg = blockIdx.x, tid = threadIdx.x;
    const int cpg = K / GN_GROUPS;
    __shared__ float s_mr[2 * GN_GROUPS];
    if (tid < GN_GROUPS) {
        const double a = moca_gstat_get(gstat + ((int64_t)sg * GN_GROUPS + tid) * 2, 0), b = moca_gstat_get(gstat + ((int64_t)sg * GN_GROUPS + tid) * 2 + 1, 1);
        const double mean = a * inv_count;
        double var = b * inv_count - mean * mean;
        if (var < 0.0) var = 0.0;
        s_mr[2 * tid] = (float)mean;
        s_mr[2 * tid + 1] = (float)(1.0 / sqrt(var + (double)eps));
    }
    __syncthreads();
    for (int k = tid; k < K; k += 256) {
        const int g = k / cpg;
        const float sc = s_mr[2 * g + 1] * gamma[k];
        s_st[k] = sc;
        s_st[K + k] = beta[k] - s_mr[2 * g] * sc;
    }
    __syncthreads();
    const int cpr = ldw / 8;                              // 16-byte chunks per W row (padding included: copied as zeros x scale 0)
    float* s_dot = s_st + 2 * K;
    const int rows_pb = 16;
    const int n0 = blockIdx.y * rows_pb;
    for (int idx = tid; idx < rows_pb * cpr; idx += 256) {
        const int r = idx / cpr, ch = idx - r * cpr, n = n0 + r;
        float dot = 0.f;
        if (n < N) {
            const half8v v = *reinterpret_cast<const half8v*>(w + (int64_t)n * ldw + ch * 8);
            half8v o;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = ch * 8 + j;
                const float x = (float)v[j];
                const float sc = k < K ? s_st[k] : 0.f, sh = k < K ? s_st[K + k] : 0.f;
                o[j] = (half_t)(x * sc);
                dot += x * sh;
            }
            *reinterpret_cast<half8v*>(wg + ((int64_t)sg * N + n) * ldw + ch * 8) = o;
        }
        s_dot[idx] = dot;
    }
    __syncthreads();
    if (tid < rows_pb && n0 + tid < N) {
        float a = bias ? bias[n0 + tid] : 0.f;
        for (int ch = 0; ch < cpr; ++ch) a += s_dot[tid * cpr + ch];
        bg[(int64_t)sg * N + n0 + tid] = a;
    }
}

extern "C" int moca_groupnorm_fold_weights_f16(const void* w, const float* bias, const float* gamma, const float* beta, const int64_t* gstat,
                                               void* wg, float* bg, int32_t n_sg, int32_t N, int32_t K, int32_t ldw, int64_t count, float eps,
                                               void* stream) {
    if (!w || !gamma || !beta || !gstat || !wg || !bg) return MOCA_E_BADARG;
    if (n_sg <= 0 || N <= 0 || K <= 0 || K % GN_GROUPS || K % 8 || ldw % 8 || ldw < K || count <= 0) return MOCA_E_BADARG;
    if ((reinterpret_cast<uintptr_t>(w) | reinterpret_cast<uintptr_t>(wg)) & 15) return MOCA_E_BADARG;
    const size_t lds = (size_t)(2 * K + 16 * (ldw / 8)) * sizeof(float);
    if (lds > 60 * 1024) return MOCA_E_BADARG;
    hipLaunchKernelGGL(gn_fold_weights_kernel, dim3(n_sg, (N + 15) / 16), dim3(256), lds, moca_stream(stream), reinterpret_cast<const half_t*>(w), bias,
                       gamma, beta, gstat, reinterpret_cast<half_t*>(wg), bg, N, K, ldw, 1.0 / (double)count, eps);
    MOCA_CHECK_LAUNCH();
    return MOCA_OK;
}

extern "C" int moca_concat_channels_gstat_f16(const void* a, const void* b, void* out, int32_t F, int32_t HW, int32_t C1, int32_t C2,
                                              int32_t frames_per_stat, int64_t* gstat, void* stream) {
    if (!a || !b || !out || !gstat || F <= 0 || HW <= 0 || C1 <= 0 || C2 <= 0 || C1 % 8 || C2 % 8) return MOCA_E_BADARG;
    const int C = C1 + C2;
    if (C % GN_GROUPS || frames_per_stat <= 0 || F % frames_per_stat) return MOCA_E_BADARG;
    const int nch8 = C / 8;
    if (nch8 > 1024) return MOCA_E_BADARG;
    int ppb = 256 / nch8;
    if (ppb < 1) ppb = 1;
    if (nch8 * ppb < 2 * GN_GROUPS) return MOCA_E_BADARG;
    const int nchunk = gn_nchunk(F, HW);
    const size_t lds = (size_t)ppb * C * 2 * sizeof(float);
    if (lds > 64 * 1024) return MOCA_E_BADARG;
    const bool streams = (int64_t)F * HW * C * 2 >= (128ll << 20);
    hipLaunchKernelGGL(streams ? concat_gstat_kernel<true> : concat_gstat_kernel<false>, dim3(F, nchunk), dim3(nch8, ppb), lds, moca_stream(stream),
                       reinterpret_cast<const half_t*>(a), reinterpret_cast<const half_t*>(b), reinterpret_cast<half_t*>(out), gstat, HW, C1, C2,
                       nchunk, frames_per_stat);
    MOCA_CHECK_LAUNCH();
    return MOCA_OK;
}

extern "C" int moca_groupnorm_gstat_cat_f16(const void* a, const void* b, void* y, const float* gamma, const float* beta,
                                            const int64_t* gstat_cat, const int64_t* gstat_b, int32_t Fb, int32_t F, int32_t HW, int32_t C1,
                                            int32_t C2, int32_t frames_per_stat, float eps, int32_t silu, void* stream) {
    if (!a || !b || !y || !gamma || !beta || !gstat_cat) return MOCA_E_BADARG;
    if ((reinterpret_cast<uintptr_t>(gamma) | reinterpret_cast<uintptr_t>(beta)) & 15) return MOCA_E_BADARG;     // (fetched as 16-byte vectors)
    if (Fb <= 0) Fb = F;
    if (Fb > F || F % Fb || Fb % frames_per_stat) return MOCA_E_BADARG;
    if (F <= 0 || HW <= 0 || C1 <= 0 || C2 <= 0 || C1 % 8 || C2 % 8 || frames_per_stat <= 0 || F % frames_per_stat) return MOCA_E_BADARG;
    const int C = C1 + C2;
    if (C % GN_GROUPS) return MOCA_E_BADARG;
    if (gstat_b) {                                                       // b's own groups must nest in the concat's
        const int cpg = C / GN_GROUPS, cpg2 = C2 / GN_GROUPS;
        if (C2 % GN_GROUPS || cpg % cpg2 || (C1 % cpg) % cpg2) return MOCA_E_BADARG;
    }
    const int nch8 = C / 8;
    if (nch8 > 1024) return MOCA_E_BADARG;
    int ppb = 256 / nch8;
    if (ppb < 1) ppb = 1;
    if (nch8 * ppb < GN_GROUPS) return MOCA_E_BADARG;
    const int nchunk = gn_nchunk(F, HW);
    const double inv_count = 1.0 / ((double)frames_per_stat * HW * (C / GN_GROUPS));
    const dim3 grid(F, nchunk), block(nch8, ppb);
    const bool streams = (int64_t)F * HW * C * 2 >= (128ll << 20);
    hipLaunchKernelGGL(streams ? gn_apply_gstat_cat_kernel<true> : gn_apply_gstat_cat_kernel<false>, grid, block, 0, moca_stream(stream),
                       reinterpret_cast<const half_t*>(a), reinterpret_cast<const half_t*>(b), reinterpret_cast<half_t*>(y), gamma, beta,
                       gstat_cat, gstat_b, HW, C1, C2, nchunk, frames_per_stat, Fb / frames_per_stat, inv_count, eps, silu);
    MOCA_CHECK_LAUNCH();
    return MOCA_OK;
}

extern "C" int moca_gstat_accum_f16(const void* x, int32_t F, int32_t HW, int32_t C, int32_t frames_per_stat, int32_t cpg, int32_t coff,
                                    int64_t* gstat, void* stream) {
    if (!x || !gstat || F <= 0 || HW <= 0 || C <= 0 || C % 8 || frames_per_stat <= 0 || F % frames_per_stat) return MOCA_E_BADARG;
    if (cpg <= 0 || coff < 0 || (coff + C - 1) / cpg >= GN_GROUPS) return MOCA_E_BADARG;
    const int nch8 = C / 8;
    if (nch8 > 1024) return MOCA_E_BADARG;
    int ppb = 256 / nch8;
    if (ppb < 1) ppb = 1;
    if (nch8 * ppb < 2 * ((coff + C - 1) / cpg - coff / cpg + 1)) return MOCA_E_BADARG;
    const int nchunk = gn_nchunk(F, HW);
    const size_t lds = (size_t)ppb * C * 2 * sizeof(float);
    if (lds > 64 * 1024) return MOCA_E_BADARG;
    hipLaunchKernelGGL(gstat_accum_kernel, dim3(F, nchunk), dim3(nch8, ppb), lds, moca_stream(stream),
                       reinterpret_cast<const half_t*>(x), gstat, HW, C, nchunk, frames_per_stat, cpg, coff);
    MOCA_CHECK_LAUNCH();
    return MOCA_OK;
}

extern "C" int moca_layernorm_f16(const void* x, void* y, const float* gamma, const float* beta,
                                  int32_t M, int32_t C, float eps, void* stream) {
    if (!x || !y || !gamma || !beta || M <= 0 || C <= 0 || C % 8) return MOCA_E_BADARG;
    const int nch = C / 8;
    hipStream_t st = moca_stream(stream);
    const half_t* xi = reinterpret_cast<const half_t*>(x);
    half_t* yo = reinterpret_cast<half_t*>(y);
    const dim3 block(256);
#define MOCA_LN(MAXCH, ROWS) hipLaunchKernelGGL((layernorm_kernel<MAXCH, ROWS>), dim3((M + 4 * ROWS - 1) / (4 * ROWS)), block, 0, st, xi, yo, gamma, beta, M, C, eps)
    if (M >= 16384) {
        if (nch <= 64) MOCA_LN(1, 4);
        else if (nch <= 128) MOCA_LN(2, 4);
        else if (nch <= 192) MOCA_LN(3, 4);
        else if (nch <= 320) MOCA_LN(5, 4);
        else return MOCA_E_BADARG;
    } else {
        if (nch <= 64) MOCA_LN(1, 1);
        else if (nch <= 128) MOCA_LN(2, 1);
        else if (nch <= 192) MOCA_LN(3, 1);
        else if (nch <= 320) MOCA_LN(5, 1);
        else return MOCA_E_BADARG;
    }
#undef MOCA_LN
    MOCA_CHECK_LAUNCH();
    return MOCA_OK;
}
