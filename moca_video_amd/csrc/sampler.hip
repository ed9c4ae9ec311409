// fp32 sampler arithmetic of the DDIM / FIFO / MoCA step (lvdm/models/samplers/ddim.py).
// Compiled with -ffp-contract=off: the reference evaluates these expressions as separate
// fp32 torch ops (mul, sub, div ...), so no FMA contraction here keeps results within
// an ulp of the CPU path.  HBM-bound, one thread per latent element.
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void cfg_combine_kernel(const float* __restrict__ ec, const float* __restrict__ eu,
                                                          float* __restrict__ out, float scale, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float u = eu[i];
        out[i] = u + scale * (ec[i] - u);   // ddim.py:304,372
    }
}

__global__ __launch_bounds__(256) void ddim_update_kernel(const float* __restrict__ x, const float* __restrict__ e,
                                                          const float* __restrict__ noise, float* __restrict__ x_prev,
                                                          float* __restrict__ pred_x0, float sqrt_at, float sqrt_aprev,
                                                          float sigma_t, float sqrt_one_minus_at, float dir_coef,
                                                          int use_scale, float scale_t, float scale_prev, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float et = e[i];
        float p0 = (x[i] - sqrt_one_minus_at * et) / sqrt_at;   // ddim.py:339
        const float dir = dir_coef * et;                        // :343
        const float nz = sigma_t * noise[i];                    // :345
        float xp;
        if (use_scale) {
            p0 = p0 / scale_t;                                  // :353
            xp = sqrt_aprev * scale_prev * p0 + dir + nz;       // :354
        } else {
            xp = sqrt_aprev * p0 + dir + nz;                    // :356
        }
        x_prev[i] = xp;
        pred_x0[i] = p0;
    }
}

// sum of each mask frame over (batch, pixels): ddim.py:585 `if mask.sum() != 0`
__global__ __launch_bounds__(256) void mask_sum_kernel(const float* __restrict__ mask, float* __restrict__ sums,
                                                       int B, int Fm, int HW) {
    __shared__ float red[4];
    const int fm = blockIdx.x;
    float s = 0.f;
    for (int i = threadIdx.x; i < B * HW; i += 256) {
        const int b = i / HW, p = i - b * HW;
        s += mask[((int64_t)b * Fm + fm) * HW + p];
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) sums[fm] = red[0] + red[1] + red[2] + red[3];
}

// MoCA FIFO step, ddim.py:405-430,556-609.  coef[i] = {sqrt(a_t), sqrt(a_prev), sigma_t,
// sqrt(1-a_t), sqrt(1-a_prev-sigma^2), 2(1 - ts/1000)}.  One thread per (b, c, pixel);
// the frame axis is walked sequentially (momentum EMA + previous-frame dependence).
__global__ __launch_bounds__(256) void fifo_step_kernel(
    const float* __restrict__ sample, const float* __restrict__ eps, const float* __restrict__ noise,
    float* __restrict__ momentum, float* __restrict__ x_prev, float* __restrict__ pred_x0,
    const float* __restrict__ coef, const float* __restrict__ mask, const float* __restrict__ mask_sums,
    const float* __restrict__ cond, const int32_t* __restrict__ mask_index, const float* __restrict__ enh,
    int B, int C, int F, int Fm, int HW, float beta, float one_minus_beta, float gamma, float one_minus_gamma) {
    const int64_t total = (int64_t)B * C * HW;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int p = (int)(i % HW);
        const int64_t bc = i / HW;
        const int b = (int)(bc / C);
        const int64_t base = bc * F * HW + p;   // element (b,c,frame 0,p); frame stride HW
        float prev = 0.f, mom_prev = momentum[base];   // momentum[:, :, 0] is never written (stays at its initial value)
        const float cnd = cond ? cond[bc * HW + p] : 0.f;
        for (int f = 0; f < F; ++f) {
            const float* cf = coef + f * 6;
            const int64_t o = base + (int64_t)f * HW;
            const float et = eps[o];
            float p0 = (sample[o] - cf[3] * et) / cf[0];        // :415
            const float dir = cf[4] * et;                       // :418
            if (f >= 1) {
                float g = p0 - prev;                            // :422
                g = g + 1.5f * dir;                             // :423
                const float m = beta * mom_prev + one_minus_beta * g;   // :424-427
                momentum[o] = m;
                mom_prev = m;
                p0 = p0 + cf[5] * m;                            // :428-430,557
            }
            prev = p0;                                          // :559
            const float nz = cf[2] * noise[o];                  // :561
            x_prev[o] = cf[1] * p0 + dir + nz;                  // :562
            const int mi = mask_index ? mask_index[f] : -1;
            if (mi >= 0 && mask && mask_sums[mi] != 0.f) {      // :565-590
                const float mk = mask[((int64_t)b * Fm + mi) * HW + p];
                if (mk > 0.5f) p0 = cnd * enh[f];
            }
            pred_x0[o] = one_minus_gamma * p0 + gamma * nz;     // :609
        }
    }
}

inline int grid_for(int64_t total) {
    int64_t g = (total + 255) / 256;
    if (g > 4096) g = 4096;
    if (g < 1) g = 1;
    return (int)g;
}

// ---- DiagonalGaussianDistribution.sample / .mode (lvdm/distributions.py:24-40) + get_first_stage_encoding's scale_factor
//      (ddpm3d.py:458-465): moments [n][2z][hw] = (mean | logvar) -> out [n][z][hw]
__global__ void gaussian_sample_kernel(const float* __restrict__ mom, const float* __restrict__ noise, float* __restrict__ out,
                                       int n, int z, int hw, float scale) {
    const int64_t total = (int64_t)n * z * hw;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = i / ((int64_t)z * hw), r = i - b * z * hw;
        const float mean = mom[b * 2 * z * hw + r];
        float v = mean;
        if (noise) {
            float lv = mom[b * 2 * z * hw + (int64_t)z * hw + r];
            lv = fminf(fmaxf(lv, -30.0f), 20.0f);
            const float sd = expf(0.5f * lv);
            v = mean + sd * noise[i];
        }
        out[i] = scale * v;
    }
}

}  // namespace

extern "C" int moca_cfg_combine_f32(const float* e_c, const float* e_u, float* out, float scale,
                                    int64_t n, void* stream) {
    if (!e_c || !e_u || !out || n <= 0) return MOCA_E_BADARG;
    hipLaunchKernelGGL(cfg_combine_kernel, dim3(grid_for(n)), dim3(256), 0, moca_stream(stream), e_c, e_u, out, scale, n);
    MOCA_CHECK_LAUNCH();
    return MOCA_OK;
}

extern "C" int moca_ddim_update_f32(const float* x, const float* e, const float* noise, float* x_prev,
                                    float* pred_x0, float a_t, float a_prev, float sigma_t,
                                    float sqrt_one_minus_at, int32_t use_scale, float scale_t,
                                    float scale_prev, int64_t n, void* stream) {
    if (!x || !e || !noise || !x_prev || !pred_x0 || n <= 0 || a_t <= 0.f) return MOCA_E_BADARG;
    // same fp32 scalar ops as the torch.full(...).sqrt() tensors of ddim.py:331-343
    const float sqrt_at = sqrtf(a_t), sqrt_aprev = sqrtf(a_prev);
    const float dir_coef = sqrtf(1.0f - a_prev - sigma_t * sigma_t);
    hipLaunchKernelGGL(ddim_update_kernel, dim3(grid_for(n)), dim3(256), 0, moca_stream(stream), x, e, noise, x_prev, pred_x0,
                       sqrt_at, sqrt_aprev, sigma_t, sqrt_one_minus_at, dir_coef, use_scale, scale_t, scale_prev, n);
    MOCA_CHECK_LAUNCH();
    return MOCA_OK;
}

extern "C" int moca_fifo_ddim_step_f32(const float* sample, const float* eps, const float* noise,
                                       float* momentum, float* x_prev, float* pred_x0,
                                       const float* coef, const float* mask, const float* cond,
                                       const int32_t* mask_index, const float* enh, float* ws,
                                       int32_t B, int32_t C, int32_t F, int32_t Fm, int32_t HW,
                                       float beta, float one_minus_beta, float gamma, float one_minus_gamma,
                                       void* stream) {
    if (!sample || !eps || !noise || !momentum || !x_prev || !pred_x0 || !coef) return MOCA_E_BADARG;
    if (B <= 0 || C <= 0 || F <= 0 || HW <= 0) return MOCA_E_BADARG;
    if (mask && (Fm <= 0 || !mask_index || !enh || !ws)) return MOCA_E_BADARG;
    hipStream_t st = moca_stream(stream);
    if (mask) {
        hipLaunchKernelGGL(mask_sum_kernel, dim3(Fm), dim3(256), 0, st, mask, ws, B, Fm, HW);
        MOCA_CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(fifo_step_kernel, dim3(grid_for((int64_t)B * C * HW)), dim3(256), 0, st, sample, eps, noise, momentum,
                       x_prev, pred_x0, coef, mask, ws, cond, mask_index, enh, B, C, F, Fm, HW, beta, one_minus_beta,
                       gamma, one_minus_gamma);
    MOCA_CHECK_LAUNCH();
    return MOCA_OK;
}

extern "C" int moca_gaussian_sample_f32(const float* moments, const float* noise, float* out, int32_t n, int32_t z, int32_t hw,
                                        float scale, void* stream) {
    if (!moments || !out || n <= 0 || z <= 0 || hw <= 0) return MOCA_E_BADARG;
    const int64_t total = (int64_t)n * z * hw;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(gaussian_sample_kernel, dim3(blocks), dim3(256), 0, moca_stream(stream), moments, noise, out, n, z, hw, scale);
    MOCA_CHECK_LAUNCH();
    return MOCA_OK;
}
