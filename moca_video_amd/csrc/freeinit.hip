// FreeInit spectral mixing (utils/freeinit_utils.py:7-47) and its frequency filters
// (:73-156) in fp32.  The latent volumes are tiny (T x 40 x 64 per channel, T = 1 in
// the driver's shift_latents call), so each axis transform is a direct DFT by
// twiddle summation: one thread per output bin, launch-bound rather than bandwidth- or
// flop-bound.  fftshift / ifftshift never move data: they become an index rotation of
// the filter, L'[i] = LPF[(i + n/2) mod n] per axis.
#include "common.h"

namespace {

struct cplx { float re, im; };

// out[outer][k][inner] = sum_j in[outer][j][inner] * exp(sign * 2 pi i j k / n)
template <bool IN_REAL>
__global__ __launch_bounds__(256) void dft_axis_kernel(const float* __restrict__ in, cplx* __restrict__ out,
                                                       int64_t total, int n, int inner, float sign) {
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int in_i = (int)(idx % inner);
        const int64_t r = idx / inner;
        const int kk = (int)(r % n);
        const int64_t outer = r / n;
        const int64_t base = outer * n * inner + in_i;
        float ar = 0.f, ai = 0.f;
        for (int j = 0; j < n; ++j) {
            const int ph = (int)(((int64_t)j * kk) % n);
            float sn, cs;
            sincospif(2.0f * (float)ph / (float)n, &sn, &cs);
            sn *= sign;
            float xr, xi;
            if (IN_REAL) { xr = in[base + (int64_t)j * inner]; xi = 0.f; }
            else { const cplx v = reinterpret_cast<const cplx*>(in)[base + (int64_t)j * inner]; xr = v.re; xi = v.im; }
            ar += xr * cs - xi * sn;
            ai += xr * sn + xi * cs;
        }
        out[idx] = cplx{ar, ai};
    }
}

// mixed = X * L' + N * (1 - L'),  X = spec[0:C], N = spec[C:2C]  (freeinit_utils.py:34-38)
__global__ __launch_bounds__(256) void mix_kernel(const cplx* __restrict__ spec, const float* __restrict__ lpf,
                                                  cplx* __restrict__ mixed, int C, int T, int H, int W) {
    const int vol = T * H * W;
    const int64_t total = (int64_t)C * vol;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int e = (int)(idx % vol);
        const int w = e % W, h = (e / W) % H, t = e / (W * H);
        const int ts = (t + T / 2) % T, hs = (h + H / 2) % H, ws = (w + W / 2) % W;
        const float l = lpf[(ts * H + hs) * W + ws];
        const float hp = 1.0f - l;
        const cplx x = spec[idx], nz = spec[idx + total];
        mixed[idx] = cplx{x.re * l + nz.re * hp, x.im * l + nz.im * hp};
    }
}

// final inverse axis: real part only, scaled by 1/(T H W)
__global__ __launch_bounds__(256) void idft_last_real_kernel(const cplx* __restrict__ in, float* __restrict__ out,
                                                             int64_t total, int n, float scale) {
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int kk = (int)(idx % n);
        const int64_t base = (idx / n) * n;
        float ar = 0.f;
        for (int j = 0; j < n; ++j) {
            const int ph = (int)(((int64_t)j * kk) % n);
            float sn, cs;
            sincospif(2.0f * (float)ph / (float)n, &sn, &cs);
            const cplx v = in[base + j];
            ar += v.re * cs - v.im * sn;
        }
        out[idx] = ar * scale;
    }
}

__global__ __launch_bounds__(256) void filter_kernel(float* __restrict__ lpf, int T, int H, int W, int type, int n,
                                                     double d_s, double d_t, int tt, int th) {
    const int total = T * H * W;
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < total; idx += gridDim.x * 256) {
        const int w = idx % W, h = (idx / W) % H, t = idx / (W * H);
        double v = 0.0;
        if (d_s != 0.0 && d_t != 0.0) {
            if (type == 3) {   // box, freeinit_utils.py:137-156
                const int ct = T / 2, cr = H / 2, cw = W / 2;
                v = (t >= ct - tt && t < ct + tt && h >= cr - th && h < cr + th && w >= cw - th && w < cw + th) ? 1.0 : 0.0;
            } else {
                const double a = (d_s / d_t) * (2.0 * t / T - 1.0);
                const double b = 2.0 * h / H - 1.0, c = 2.0 * w / W - 1.0;
                const double d2 = a * a + b * b + c * c;
                if (type == 0) v = exp(-1.0 / (2.0 * d_s * d_s) * d2);                    // gaussian :73-91
                else if (type == 1) v = 1.0 / (1.0 + pow(d2 / (d_s * d_s), (double)n));   // butterworth :94-113
                else v = d2 <= d_s * 2.0 ? 1.0 : 0.0;                                      // ideal :116-134
            }
        }
        lpf[idx] = (float)v;
    }
}

inline int grid_for(int64_t total) {
    int64_t g = (total + 255) / 256;
    if (g > 4096) g = 4096;
    if (g < 1) g = 1;
    return (int)g;
}

}  // namespace

extern "C" int64_t moca_freq_mix_ws_bytes(int32_t C, int32_t T, int32_t H, int32_t W) {
    // two complex ping-pong buffers of 2C volumes
    return (int64_t)2 * 2 * C * T * H * W * 8;
}

extern "C" int moca_freq_mix_3d_f32(const float* x, const float* noise, const float* lpf, float* out,
                                    int32_t C, int32_t T, int32_t H, int32_t W, float* ws, void* stream) {
    if (!x || !noise || !lpf || !out || !ws || C <= 0 || T <= 0 || H <= 0 || W <= 0) return MOCA_E_BADARG;
    hipStream_t st = moca_stream(stream);
    const int64_t vol = (int64_t)T * H * W;
    const int64_t n1 = (int64_t)C * vol;       // complex elements per tensor
    cplx* bufA = reinterpret_cast<cplx*>(ws);
    cplx* bufB = bufA + 2 * n1;
    // forward along W (real input) for x -> bufA[0:n1], noise -> bufA[n1:2n1]
    hipLaunchKernelGGL(dft_axis_kernel<true>, dim3(grid_for(n1)), dim3(256), 0, st, x, bufA, n1, W, 1, -1.0f);
    MOCA_CHECK_LAUNCH();
    hipLaunchKernelGGL(dft_axis_kernel<true>, dim3(grid_for(n1)), dim3(256), 0, st, noise, bufA + n1, n1, W, 1, -1.0f);
    MOCA_CHECK_LAUNCH();
    // forward along H
    hipLaunchKernelGGL(dft_axis_kernel<false>, dim3(grid_for(2 * n1)), dim3(256), 0, st, reinterpret_cast<const float*>(bufA), bufB,
                       2 * n1, H, W, -1.0f);
    MOCA_CHECK_LAUNCH();
    cplx* spec = bufB;
    cplx* other = bufA;
    if (T > 1) {
        hipLaunchKernelGGL(dft_axis_kernel<false>, dim3(grid_for(2 * n1)), dim3(256), 0, st, reinterpret_cast<const float*>(bufB), bufA,
                           2 * n1, T, H * W, -1.0f);
        MOCA_CHECK_LAUNCH();
        spec = bufA; other = bufB;
    }
    // mix -> other[0:n1]
    hipLaunchKernelGGL(mix_kernel, dim3(grid_for(n1)), dim3(256), 0, st, spec, lpf, other, C, T, H, W);
    MOCA_CHECK_LAUNCH();
    cplx* cur = other;      // n1 elements valid
    cplx* nxt = spec;
    if (T > 1) {
        hipLaunchKernelGGL(dft_axis_kernel<false>, dim3(grid_for(n1)), dim3(256), 0, st, reinterpret_cast<const float*>(cur), nxt,
                           n1, T, H * W, 1.0f);
        MOCA_CHECK_LAUNCH();
        cplx* t = cur; cur = nxt; nxt = t;
    }
    hipLaunchKernelGGL(dft_axis_kernel<false>, dim3(grid_for(n1)), dim3(256), 0, st, reinterpret_cast<const float*>(cur), nxt,
                       n1, H, W, 1.0f);
    MOCA_CHECK_LAUNCH();
    hipLaunchKernelGGL(idft_last_real_kernel, dim3(grid_for(n1)), dim3(256), 0, st, nxt, out, n1, W, 1.0f / (float)vol);
    MOCA_CHECK_LAUNCH();
    return MOCA_OK;
}

extern "C" int moca_freq_filter_f32(float* lpf, int32_t T, int32_t H, int32_t W, int32_t type, int32_t n,
                                    double d_s, double d_t, void* stream) {
    if (!lpf || T <= 0 || H <= 0 || W <= 0 || type < 0 || type > 3) return MOCA_E_BADARG;
    // box thresholds: Python round() is round-half-to-even == rint() in the default mode
    const int th = (int)rint((double)(H / 2) * d_s);
    const int tt = (int)rint((double)(T / 2) * d_t);
    hipLaunchKernelGGL(filter_kernel, dim3(grid_for((int64_t)T * H * W)), dim3(256), 0, moca_stream(stream), lpf, T, H, W,
                       type, n, d_s, d_t, tt, th);
    MOCA_CHECK_LAUNCH();
    return MOCA_OK;
}
