// Attention kernels for gfx950, head dim 64, fp16 I/O, fp32 softmax.
//
// moca_attention_f16: flash-style softmax(QK^T*scale)V for the spatial self / cross
//   attention (N = H*W up to 2560 queries; 77/154 context keys).  256 threads = 4
//   wavefronts x 32 query rows; K/V tiles of 64 keys staged in LDS (K row-major with
//   an XOR swizzle, V transposed + key-permuted so that the P.V operand is one
//   ds_read_b128).  The score tile is computed "swapped" (S^T = K.Q^T with
//   v_mfma_f32_32x32x16_f16) so every lane owns ONE query column: the online-softmax
//   row reduction is 31 in-lane max/adds plus one lane<->lane+32 exchange, and the
//   S^T accumulator registers are directly the B operand of O^T += V^T.P^T.
//
// moca_temporal_attention_f16: attention over the frame axis (T <= 16) per
//   (pixel, head): one wavefront per problem with v_mfma_f32_16x16x32_f16 (QK^T) and
//   v_mfma_f32_16x16x16_f16 (PV); gathers the T frames of a pixel straight from the
//   channels-last token matrix, so the reference's (b t) c h w <-> (b h w) t c
//   reshuffles (attention.py:335-338,367) never touch memory.
#include "common.h"

namespace {

constexpr int D = 64;        // head dim
constexpr int KT = 64;       // keys per LDS tile
constexpr int QB = 128;      // queries per block
constexpr int ROWB = 128;    // bytes per LDS row (64 halves)

// position of key kk (0..63) inside a transposed V row so that the 8 keys one lane
// needs for k-step (sub, s) and lane-half h are contiguous (see header comment)
__device__ __forceinline__ int vpos(int key) {
    const int sub = key >> 5, kk = key & 31;
    const int s = kk >> 4, jhi = (kk >> 3) & 1, h = (kk >> 2) & 1, jlo = kk & 3;
    return sub * 32 + (s * 2 + h) * 8 + jhi * 4 + jlo;
}

__global__ __launch_bounds__(256, 2) void attention_kernel(
    const half_t* __restrict__ q, const half_t* __restrict__ k, const half_t* __restrict__ v, half_t* __restrict__ out,
    int heads, int Nq, int Nk, int ldq, int ldk, int ldv, int ldo, int kv_div, float scale_log2e) {
    __shared__ __attribute__((aligned(16))) char sK[KT * ROWB];
    __shared__ __attribute__((aligned(16))) char sVt[D * ROWB];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int bq = blockIdx.y / heads, head = blockIdx.y % heads;
    const int bkv = bq / kv_div;
    const int q0 = blockIdx.x * QB + wave * 32;
    const int fr = lane & 31, fh = lane >> 5;

    const half_t* qb = q + (int64_t)bq * Nq * ldq + head * D;
    const half_t* kb = k + (int64_t)bkv * Nk * ldk + head * D;
    const half_t* vb = v + (int64_t)bkv * Nk * ldv + head * D;

    // Q fragments (B operand of S^T = K.Q^T): lane (q=fr, h=fh) holds Q[q][16ks + 8h + j]
    half8v qf[4];
    const int qrow = q0 + fr;
    const bool q_ok = qrow < Nq;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        half8v t = {0, 0, 0, 0, 0, 0, 0, 0};
        if (q_ok) t = *reinterpret_cast<const half8v*>(qb + (int64_t)qrow * ldq + ks * 16 + fh * 8);
        qf[ks] = t;
    }

    f32x16 o[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    // staging coordinates: thread -> rows (tid>>3) + 32 i, chunk tid&7
    const int cc = tid & 7, r0 = tid >> 3;
    half8v rk[2], rv[2];
    auto load_kv = [&](int kt) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int key = kt * KT + r0 + 32 * i;
            half8v a = {0, 0, 0, 0, 0, 0, 0, 0}, b = {0, 0, 0, 0, 0, 0, 0, 0};
            if (key < Nk) {
                a = *reinterpret_cast<const half8v*>(kb + (int64_t)key * ldk + cc * 8);
                b = *reinterpret_cast<const half8v*>(vb + (int64_t)key * ldv + cc * 8);
            }
            rk[i] = a; rv[i] = b;
        }
    };
    auto store_kv = [&]() {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = r0 + 32 * i;
            *reinterpret_cast<half8v*>(sK + row * ROWB + ((cc ^ ((row >> 1) & 7)) << 4)) = rk[i];
            const int pos = vpos(row);
            const int pch = pos >> 3, pin = pos & 7;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int d = cc * 8 + j;
                *reinterpret_cast<half_t*>(sVt + d * ROWB + ((pch ^ ((d >> 1) & 7)) << 4) + pin * 2) = rv[i][j];
            }
        }
    };

    const int nkt = (Nk + KT - 1) / KT;
    load_kv(0);
    for (int kt = 0; kt < nkt; ++kt) {
        __syncthreads();   // previous tile's LDS reads are done
        store_kv();
        __syncthreads();
        if (kt + 1 < nkt) load_kv(kt + 1);

        // ---- S^T = K . Q^T ----
        f32x16 s[2];
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[sub][r] = 0.f;
            const int row = sub * 32 + fr;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int ch = ks * 2 + fh;
                const half8v kf = *reinterpret_cast<const half8v*>(sK + row * ROWB + ((ch ^ ((row >> 1) & 7)) << 4));
                s[sub] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[ks], s[sub], 0, 0, 0);
            }
        }
        // mask keys beyond Nk (last tile only)
        if ((kt + 1) * KT > Nk) {
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kt * KT + sub * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                    if (key >= Nk) s[sub][r] = -INFINITY;
                }
        }
        // ---- online softmax (this lane = one query column; partner lane^32 has the other keys) ----
        float tmax = s[0][0];
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, s[sub][r]);
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float m_new = fmaxf(m_run, tmax);
        const float alpha = exp2f((m_run - m_new) * scale_log2e);
        const float mb = m_new * scale_log2e;
        float psum = 0.f;
        half8v pf[2][2];
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pv = exp2f(s[sub][r] * scale_log2e - mb);
                psum += pv;
                pf[sub][r >> 3][r & 7] = (half_t)pv;
            }
        l_run = l_run * alpha + psum;
        m_run = m_new;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
        // ---- O^T += V^T . P^T ----
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
            const int drow = dt * 32 + fr;
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int ss = 0; ss < 2; ++ss) {
                    const int ch = sub * 4 + ss * 2 + fh;
                    const half8v vf = *reinterpret_cast<const half8v*>(sVt + drow * ROWB + ((ch ^ ((drow >> 1) & 7)) << 4));
                    o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[sub][ss], o[dt], 0, 0, 0);
                }
        }
    }

    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    if (q_ok) {
        half_t* ob = out + ((int64_t)bq * Nq + qrow) * ldo + head * D;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                half4v h4;
#pragma unroll
                for (int j = 0; j < 4; ++j) h4[j] = (half_t)(o[dt][g * 4 + j] * inv);
                *reinterpret_cast<half4v*>(ob + dt * 32 + 8 * g + 4 * fh) = h4;
            }
    }
}

// ---- temporal attention: one wavefront per (video, pixel, head), T <= 16 -------
constexpr int TV_ROWB = 144;  // 128 B + 16 B pad per V row in LDS

__global__ __launch_bounds__(256) void temporal_attention_kernel(
    const half_t* __restrict__ q, const half_t* __restrict__ k, const half_t* __restrict__ v, half_t* __restrict__ out,
    int B, int T, int HW, int heads, int ld, int ldo, float scale_log2e) {
    __shared__ __attribute__((aligned(16))) char sV[4][16 * TV_ROWB];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t total = (int64_t)B * HW * heads;
    int64_t prob = (int64_t)blockIdx.x * 4 + wave;
    const bool active = prob < total;
    if (!active) prob = total - 1;
    const int head = (int)(prob % heads);
    const int64_t bp = prob / heads;
    const int pix = (int)(bp % HW), b = (int)(bp / HW);
    // token row of frame t: (b*T + t)*HW + pix
    const int64_t row0 = (int64_t)b * T * HW + pix;
    const int64_t rstride = (int64_t)HW * ld;
    const int64_t base = row0 * ld + head * D;

    const int fr = lane & 15, fg = lane >> 4;
    // K (A operand) and Q (B operand) fragments of S^T = K.Q^T: row fr, d = 32 ks + 8 fg + j
    half8v kf[2], qf[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        half8v a = {0, 0, 0, 0, 0, 0, 0, 0}, c = {0, 0, 0, 0, 0, 0, 0, 0};
        if (fr < T) {
            a = *reinterpret_cast<const half8v*>(k + base + fr * rstride + ks * 32 + fg * 8);
            c = *reinterpret_cast<const half8v*>(q + base + fr * rstride + ks * 32 + fg * 8);
        }
        kf[ks] = a; qf[ks] = c;
    }
    // V -> LDS row-major [key][d]
    char* sv = sV[wave];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = (lane >> 3) + 8 * i, ch = lane & 7;
        half8v a = {0, 0, 0, 0, 0, 0, 0, 0};
        if (row < T) a = *reinterpret_cast<const half8v*>(v + base + row * rstride + ch * 8);
        *reinterpret_cast<half8v*>(sv + row * TV_ROWB + ch * 16) = a;
    }
    __syncthreads();

    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    s = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[0], qf[0], s, 0, 0, 0);
    s = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[1], qf[1], s, 0, 0, 0);
    // lane holds S^T[key = 4 fg + r][q = fr]
    float mx = -INFINITY;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        if (4 * fg + r >= T) s[r] = -INFINITY;
        mx = fmaxf(mx, s[r]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
    half4v pf;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float pv = exp2f((s[r] - mx) * scale_log2e);
        sum += pv;
        pf[r] = (half_t)pv;
    }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.0f / sum;

    // O^T[d][q] = V^T[d][key] . P^T[key][q]; A: lane (d = fr, k = 4 fg + j) = V[4fg+j][16dt + fr]
    const bool st_ok = active && fr < T;
    half_t* ob = out + (row0 + (int64_t)fr * HW) * ldo + head * D;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
        half4v vf;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            vf[j] = *reinterpret_cast<const half_t*>(sv + (4 * fg + j) * TV_ROWB + (dt * 16 + fr) * 2);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        acc = __builtin_amdgcn_mfma_f32_16x16x16f16(vf, pf, acc, 0, 0, 0);
        // lane holds O^T[d = 16dt + 4fg + r][q = fr]
        if (st_ok) {
            half4v h4;
#pragma unroll
            for (int r = 0; r < 4; ++r) h4[r] = (half_t)(acc[r] * inv);
            *reinterpret_cast<half4v*>(ob + dt * 16 + 4 * fg) = h4;
        }
    }
}

}  // namespace

extern "C" int moca_attention_f16(const void* q, const void* k, const void* v, void* out,
                                  int32_t Bq, int32_t heads, int32_t Nq, int32_t Nk,
                                  int32_t ldq, int32_t ldk, int32_t ldv, int32_t ldo,
                                  int32_t kv_div, float scale, void* stream) {
    if (!q || !k || !v || !out) return MOCA_E_BADARG;
    if (Bq <= 0 || heads <= 0 || Nq <= 0 || Nk <= 0 || kv_div <= 0 || Bq % kv_div) return MOCA_E_BADARG;
    if (ldq % 8 || ldk % 8 || ldv % 8 || ldo % 4) return MOCA_E_BADARG;
    if (ldq < heads * D || ldk < heads * D || ldv < heads * D || ldo < heads * D) return MOCA_E_BADARG;
    if ((int64_t)Bq * heads > 65535) return MOCA_E_BADARG;
    const dim3 grid((Nq + QB - 1) / QB, Bq * heads), block(256);
    hipLaunchKernelGGL(attention_kernel, grid, block, 0, moca_stream(stream),
                       reinterpret_cast<const half_t*>(q), reinterpret_cast<const half_t*>(k),
                       reinterpret_cast<const half_t*>(v), reinterpret_cast<half_t*>(out),
                       heads, Nq, Nk, ldq, ldk, ldv, ldo, kv_div, scale * 1.4426950408889634f);
    MOCA_CHECK_LAUNCH();
    return MOCA_OK;
}

extern "C" int moca_temporal_attention_f16(const void* q, const void* k, const void* v, void* out,
                                           int32_t B, int32_t T, int32_t HW, int32_t heads,
                                           int32_t ld_qkv, int32_t ldo, float scale, void* stream) {
    if (!q || !k || !v || !out) return MOCA_E_BADARG;
    if (B <= 0 || T <= 0 || T > 16 || HW <= 0 || heads <= 0) return MOCA_E_BADARG;
    if (ld_qkv % 8 || ldo % 4 || ld_qkv < heads * D || ldo < heads * D) return MOCA_E_BADARG;
    const int64_t total = (int64_t)B * HW * heads;
    const dim3 grid((unsigned)((total + 3) / 4)), block(256);
    hipLaunchKernelGGL(temporal_attention_kernel, grid, block, 0, moca_stream(stream),
                       reinterpret_cast<const half_t*>(q), reinterpret_cast<const half_t*>(k),
                       reinterpret_cast<const half_t*>(v), reinterpret_cast<half_t*>(out),
                       B, T, HW, heads, ld_qkv, ldo, scale * 1.4426950408889634f);
    MOCA_CHECK_LAUNCH();
    return MOCA_OK;
}
