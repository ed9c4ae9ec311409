// Layout conversion, channel concat, sinusoidal embedding and the small row-broadcast
// SiLU/add used by the time/fps embedding MLPs.  All HBM-bound, 16-byte accesses where
// the layout allows it.
#include "common.h"

namespace {

// x [B][Cin][T][HW] -> y [(b*T+t)*HW + p][Cpad] fp16, zero padded channels
template <typename TIN>
__global__ __launch_bounds__(256) void ncthw_to_nhwc_kernel(const TIN* __restrict__ x, half_t* __restrict__ y,
                                                            int B, int Cin, int T, int HW, int Cpad) {
    const int64_t total = (int64_t)B * T * HW;
    for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < total; r += (int64_t)gridDim.x * 256) {
        const int p = (int)(r % HW);
        const int64_t bt = r / HW;
        const int t = (int)(bt % T), b = (int)(bt / T);
        for (int c = 0; c < Cpad; ++c) {
            float v = 0.f;
            if (c < Cin) v = (float)x[(((int64_t)b * Cin + c) * T + t) * HW + p];
            y[r * Cpad + c] = (half_t)v;
        }
    }
}

template <typename TOUT>
__global__ __launch_bounds__(256) void nhwc_to_ncthw_kernel(const half_t* __restrict__ y, int ld, TOUT* __restrict__ x,
                                                            int B, int Cout, int T, int HW) {
    const int64_t total = (int64_t)B * Cout * T * HW;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int p = (int)(i % HW);
        int64_t r = i / HW;
        const int t = (int)(r % T); r /= T;
        const int c = (int)(r % Cout);
        const int b = (int)(r / Cout);
        x[i] = (TOUT)(float)y[(((int64_t)b * T + t) * HW + p) * ld + c];
    }
}

__global__ __launch_bounds__(256) void concat_kernel(const half_t* __restrict__ a, const half_t* __restrict__ b,
                                                     half_t* __restrict__ out, int64_t rows, int C1, int C2) {
    const int n1 = C1 / 8, n2 = C2 / 8, n = n1 + n2;
    const int64_t total = rows * n;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / n;
        const int ch = (int)(i - r * n);
        half8v v;
        if (ch < n1) v = *reinterpret_cast<const half8v*>(a + r * C1 + ch * 8);
        else v = *reinterpret_cast<const half8v*>(b + r * C2 + (ch - n1) * 8);
        *reinterpret_cast<half8v*>(out + r * (C1 + C2) + ch * 8) = v;
    }
}

// dst = [src; src; ...] (reps copies of n16 16-byte chunks): the point where a forward whose batch repeats the same latents with
// different contexts (classifier-free guidance) stops sharing its activations
__global__ __launch_bounds__(256) void repeat_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, int64_t n16, int reps) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (int64_t)gridDim.x * 256) {
        const uint4 v = src[i];
        for (int r = 0; r < reps; ++r) dst[(int64_t)r * n16 + i] = v;
    }
}

// utils_diffusion.py:8-28: [cos(t f_k), sin(t f_k)], f_k = exp(-ln(P) k / half)
__global__ __launch_bounds__(256) void timestep_embedding_kernel(const int64_t* __restrict__ t, half_t* __restrict__ out,
                                                                 int n, int dim, float neg_log_period) {
    const int half_dim = dim / 2;
    const int total = n * dim;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int row = i / dim, c = i - row * dim;
        float v = 0.f;
        if (c < 2 * half_dim) {
            const int kf = c < half_dim ? c : c - half_dim;
            const float freq = expf(neg_log_period * (float)kf / (float)half_dim);
            const float arg = (float)t[row] * freq;
            v = c < half_dim ? cosf(arg) : sinf(arg);
        }
        out[i] = (half_t)v;
    }
}

__global__ __launch_bounds__(256) void silu_add_rows_kernel(const half_t* __restrict__ a, int div_a, const half_t* __restrict__ b,
                                                            int div_b, half_t* __restrict__ out, int rows, int C, int apply_silu) {
    const int total = rows * C;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int r = i / C, c = i - r * C;
        float v = (float)a[(int64_t)(r / div_a) * C + c];
        if (b) v += (float)b[(int64_t)(r / div_b) * C + c];
        if (apply_silu) v = moca_silu(v);
        out[i] = (half_t)v;
    }
}

inline int grid_for(int64_t total) {
    int64_t g = (total + 255) / 256;
    if (g > 4096) g = 4096;
    if (g < 1) g = 1;
    return (int)g;
}

// ---- VAE: z / scale_factor -> post_quant_conv (1x1, <= 8 channels) -> channels-last fp16 ---------------
// out[(b*T+t)*HW + p][co] = bias[co] + sum_ci w[co][ci] * (z[b][ci][t][p] * inv_scale), co < Cout; 0 for Cout <= co < Cpad
template <typename TIN>
__global__ void channel_mix_kernel(const TIN* __restrict__ z, const float* __restrict__ w, const float* __restrict__ bias,
                                   half_t* __restrict__ out, int B, int Cin, int T, int HW, int Cout, int Cpad, float inv_scale) {
    const int64_t total = (int64_t)B * T * HW;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int p = (int)(i % HW);
        const int64_t bt = i / HW;
        const int t = (int)(bt % T), b = (int)(bt / T);
        float v[8];
        for (int ci = 0; ci < Cin; ++ci) v[ci] = (float)z[(((int64_t)b * Cin + ci) * T + t) * HW + p] * inv_scale;
        for (int co = 0; co < Cpad; ++co) {
            float a = 0.f;
            if (co < Cout) {
                a = bias ? bias[co] : 0.f;
                for (int ci = 0; ci < Cin; ++ci) a += w[co * Cin + ci] * v[ci];
            }
            out[i * Cpad + co] = (half_t)a;
        }
    }
}

// ---- row softmax: p[r][:] = softmax(scale * s[r][:]) , fp32 in, fp16 out; one wavefront per row -------
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ s, half_t* __restrict__ pr, int64_t R,
                                                           int N, int64_t lds_, int64_t ldp, float scale_log2e) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= R) return;
    const float* x = s + row * lds_;
    float m = -INFINITY;
    for (int j = lane * 4; j < N; j += 256) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + j);
        m = fmaxf(fmaxf(m, fmaxf(v[0], v[1])), fmaxf(v[2], v[3]));
    }
    m = wave_max(m);
    float sum = 0.f;
    for (int j = lane * 4; j < N; j += 256) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + j);
#pragma unroll
        for (int q = 0; q < 4; ++q) sum += exp2f((v[q] - m) * scale_log2e);
    }
    sum = wave_sum(sum);
    const float inv = 1.f / sum;
    half_t* o = pr + row * ldp;
    for (int j = lane * 4; j < N; j += 256) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + j);
        half4v h;
#pragma unroll
        for (int q = 0; q < 4; ++q) h[q] = (half_t)(exp2f((v[q] - m) * scale_log2e) * inv);
        *reinterpret_cast<half4v*>(o + j) = h;
    }
}

// ---- token + positional embedding of the CLIP text tower ----
__global__ void embed_tokens_kernel(const int64_t* __restrict__ tokens, const float* __restrict__ table, const float* __restrict__ pos,
                                    half_t* __restrict__ out, int n_tokens, int L, int C, int vocab) {
    const int64_t total = (int64_t)n_tokens * C;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int t = (int)(i / C), c = (int)(i - (int64_t)t * C);
        int64_t tok = tokens[t];
        tok = tok < 0 ? 0 : (tok >= vocab ? vocab - 1 : tok);
        out[i] = (half_t)(table[tok * C + c] + pos[(int64_t)(t % L) * C + c]);
    }
}

}  // namespace

extern "C" int moca_ncthw_to_nhwc_f16(const void* x, int32_t x_is_f32, void* y, int32_t B, int32_t Cin,
                                      int32_t T, int32_t HW, int32_t Cpad, void* stream) {
    if (!x || !y || B <= 0 || Cin <= 0 || T <= 0 || HW <= 0 || Cpad < Cin) return MOCA_E_BADARG;
    const int g = grid_for((int64_t)B * T * HW);
    if (x_is_f32)
        hipLaunchKernelGGL(ncthw_to_nhwc_kernel<float>, dim3(g), dim3(256), 0, moca_stream(stream),
                           reinterpret_cast<const float*>(x), reinterpret_cast<half_t*>(y), B, Cin, T, HW, Cpad);
    else
        hipLaunchKernelGGL(ncthw_to_nhwc_kernel<half_t>, dim3(g), dim3(256), 0, moca_stream(stream),
                           reinterpret_cast<const half_t*>(x), reinterpret_cast<half_t*>(y), B, Cin, T, HW, Cpad);
    MOCA_CHECK_LAUNCH();
    return MOCA_OK;
}

extern "C" int moca_nhwc_to_ncthw(const void* y, int32_t ld, void* x, int32_t x_is_f32, int32_t B,
                                  int32_t Cout, int32_t T, int32_t HW, void* stream) {
    if (!x || !y || B <= 0 || Cout <= 0 || T <= 0 || HW <= 0 || ld < Cout) return MOCA_E_BADARG;
    const int g = grid_for((int64_t)B * Cout * T * HW);
    if (x_is_f32)
        hipLaunchKernelGGL(nhwc_to_ncthw_kernel<float>, dim3(g), dim3(256), 0, moca_stream(stream),
                           reinterpret_cast<const half_t*>(y), ld, reinterpret_cast<float*>(x), B, Cout, T, HW);
    else
        hipLaunchKernelGGL(nhwc_to_ncthw_kernel<half_t>, dim3(g), dim3(256), 0, moca_stream(stream),
                           reinterpret_cast<const half_t*>(y), ld, reinterpret_cast<half_t*>(x), B, Cout, T, HW);
    MOCA_CHECK_LAUNCH();
    return MOCA_OK;
}

extern "C" int moca_concat_channels_f16(const void* a, const void* b, void* out, int64_t rows,
                                        int32_t C1, int32_t C2, void* stream) {
    if (!a || !b || !out || rows <= 0 || C1 <= 0 || C2 <= 0 || C1 % 8 || C2 % 8) return MOCA_E_BADARG;
    const int g = grid_for(rows * ((C1 + C2) / 8));
    hipLaunchKernelGGL(concat_kernel, dim3(g), dim3(256), 0, moca_stream(stream), reinterpret_cast<const half_t*>(a),
                       reinterpret_cast<const half_t*>(b), reinterpret_cast<half_t*>(out), rows, C1, C2);
    MOCA_CHECK_LAUNCH();
    return MOCA_OK;
}

extern "C" int moca_repeat_f16(const void* src, void* dst, int64_t bytes, int32_t reps, void* stream) {
    if (!src || !dst || bytes <= 0 || bytes % 16 || reps < 1 || ((uintptr_t)src & 15) || ((uintptr_t)dst & 15)) return MOCA_E_BADARG;
    hipLaunchKernelGGL(repeat_kernel, dim3(grid_for(bytes / 16)), dim3(256), 0, moca_stream(stream),
                       reinterpret_cast<const uint4*>(src), reinterpret_cast<uint4*>(dst), bytes / 16, reps);
    MOCA_CHECK_LAUNCH();
    return MOCA_OK;
}

extern "C" int moca_timestep_embedding_f16(const int64_t* t, void* out, int32_t n, int32_t dim,
                                           float max_period, void* stream) {
    if (!t || !out || n <= 0 || dim <= 1 || max_period <= 0.f) return MOCA_E_BADARG;
    const int g = grid_for((int64_t)n * dim);
    hipLaunchKernelGGL(timestep_embedding_kernel, dim3(g), dim3(256), 0, moca_stream(stream), t,
                       reinterpret_cast<half_t*>(out), n, dim, -logf(max_period));
    MOCA_CHECK_LAUNCH();
    return MOCA_OK;
}

extern "C" int moca_silu_add_rows_f16(const void* a, int32_t div_a, const void* b, int32_t div_b, void* out,
                                      int32_t rows, int32_t C, int32_t apply_silu, void* stream) {
    if (!a || !out || rows <= 0 || C <= 0 || div_a <= 0 || (b && div_b <= 0)) return MOCA_E_BADARG;
    const int g = grid_for((int64_t)rows * C);
    hipLaunchKernelGGL(silu_add_rows_kernel, dim3(g), dim3(256), 0, moca_stream(stream), reinterpret_cast<const half_t*>(a),
                       div_a, reinterpret_cast<const half_t*>(b), div_b > 0 ? div_b : 1, reinterpret_cast<half_t*>(out),
                       rows, C, apply_silu);
    MOCA_CHECK_LAUNCH();
    return MOCA_OK;
}

extern "C" int moca_channel_mix_f16(const void* z, int32_t z_is_f32, const float* w, const float* bias, void* out,
                                    int32_t B, int32_t Cin, int32_t T, int32_t HW, int32_t Cout, int32_t Cpad,
                                    float inv_scale, void* stream) {
    if (!z || !w || !out || B <= 0 || T <= 0 || HW <= 0 || Cin <= 0 || Cin > 8 || Cout <= 0 || Cout > Cpad) return MOCA_E_BADARG;
    const int g = grid_for((int64_t)B * T * HW);
    if (z_is_f32)
        hipLaunchKernelGGL(channel_mix_kernel<float>, dim3(g), dim3(256), 0, moca_stream(stream), reinterpret_cast<const float*>(z),
                           w, bias, reinterpret_cast<half_t*>(out), B, Cin, T, HW, Cout, Cpad, inv_scale);
    else
        hipLaunchKernelGGL(channel_mix_kernel<half_t>, dim3(g), dim3(256), 0, moca_stream(stream), reinterpret_cast<const half_t*>(z),
                           w, bias, reinterpret_cast<half_t*>(out), B, Cin, T, HW, Cout, Cpad, inv_scale);
    MOCA_CHECK_LAUNCH();
    return MOCA_OK;
}

extern "C" int moca_softmax_rows_f16(const float* s, void* p, int64_t R, int32_t N, int64_t lds, int64_t ldp, float scale,
                                     void* stream) {
    if (!s || !p || R <= 0 || N <= 0 || N % 4 || lds % 4 || ldp % 4 || lds < N || ldp < N) return MOCA_E_BADARG;
    const int64_t blocks = (R + 3) / 4;
    if (blocks > 0x7fffffff) return MOCA_E_BADARG;
    hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, moca_stream(stream), s,
                       reinterpret_cast<half_t*>(p), R, N, lds, ldp, scale * 1.4426950408889634f);
    MOCA_CHECK_LAUNCH();
    return MOCA_OK;
}

extern "C" int moca_embed_tokens_f16(const int64_t* tokens, const float* table, const float* pos, void* out,
                                     int32_t n_tokens, int32_t L, int32_t C, int32_t vocab, void* stream) {
    if (!tokens || !table || !pos || !out || n_tokens <= 0 || L <= 0 || C <= 0 || vocab <= 0) return MOCA_E_BADARG;
    const int g = grid_for((int64_t)n_tokens * C);
    hipLaunchKernelGGL(embed_tokens_kernel, dim3(g), dim3(256), 0, moca_stream(stream), tokens, table, pos,
                       reinterpret_cast<half_t*>(out), n_tokens, L, C, vocab);
    MOCA_CHECK_LAUNCH();
    return MOCA_OK;
}
