// Shared device/host helpers for libmoca_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/moca_hip.h"

typedef _Float16 half_t;
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef _Float16 half4v __attribute__((ext_vector_type(4)));
typedef _Float16 half8v __attribute__((ext_vector_type(8)));
typedef float    f32x4  __attribute__((ext_vector_type(4)));
typedef float    f32x16 __attribute__((ext_vector_type(16)));

#define MOCA_WAVE 64

#define MOCA_CHECK_LAUNCH()                                  \
    do {                                                     \
        hipError_t e__ = hipGetLastError();                  \
        if (e__ != hipSuccess) return MOCA_E_LAUNCH;         \
    } while (0)

static inline hipStream_t moca_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

__device__ __forceinline__ float moca_silu(float x) { return x / (1.0f + __expf(-x)); }
// erf GELU, as F.gelu default (attention.py:383).  erf by Abramowitz-Stegun 7.1.26
// (|abs err| <= 1.5e-7, far below the fp16 output resolution): 1 rcp + 1 exp + 6 FMA instead
// of libm erff's ~40-instruction branchy polynomial -- the GEGLU epilogue runs it M*4C times.
__device__ __forceinline__ float moca_erf(float x) {
    const float ax = fabsf(x);
    const float t = __frcp_rn(1.0f + 0.3275911f * ax);
    float y = 1.061405429f;
    y = y * t - 1.453152027f;
    y = y * t + 1.421413741f;
    y = y * t - 0.284496736f;
    y = y * t + 0.254829592f;
    y = 1.0f - y * t * __expf(-ax * ax);
    return copysignf(y, x);
}
__device__ __forceinline__ float moca_gelu(float x) { return 0.5f * x * (1.0f + moca_erf(x * 0.70710678118654752f)); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
