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

// x * sigmoid(x) with v_exp + v_rcp (1 ulp each, far below the fp16 output resolution) instead of an IEEE division: the
// GroupNorm apply pass runs it on every activation and the division's ~10 instructions were a third of its VALU work
__device__ __forceinline__ float moca_silu(float x) {
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.4426950408889634f));
}
// erf GELU, as F.gelu default (attention.py:383).  erf by Abramowitz-Stegun 7.1.26
// (|abs err| <= 1.5e-7 + 1 ulp of v_rcp_f32, far below the fp16 output resolution): 1 v_rcp + 1 v_exp
// + 6 FMA instead of libm erff's ~40-instruction branchy polynomial.  The GEGLU epilogue runs it
// M*4C times per FeedForward and is VALU-bound, so the 2-wide form below is written on float2 vectors:
// hipcc emits v_pk_mul_f32 / v_pk_fma_f32 for it (half the VALU issue slots of the scalar form).
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float moca_erf(float x) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * ax);
    float y = 1.061405429f;
    y = y * t - 1.453152027f;
    y = y * t + 1.421413741f;
    y = y * t - 0.284496736f;
    y = y * t + 0.254829592f;
    y = 1.0f - y * t * __builtin_amdgcn_exp2f(ax * ax * -1.4426950408889634f);
    return copysignf(y, x);
}
__device__ __forceinline__ float moca_gelu(float x) { return 0.5f * x * (1.0f + moca_erf(x * 0.70710678118654752f)); }

__device__ __forceinline__ f32x2 moca_erf2(f32x2 x) {
    const f32x2 ax = {fabsf(x[0]), fabsf(x[1])};
    const f32x2 d = ax * 0.3275911f + 1.0f;
    const f32x2 t = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
    f32x2 y = t * 1.061405429f - 1.453152027f;
    y = y * t + 1.421413741f;
    y = y * t - 0.284496736f;
    y = y * t + 0.254829592f;
    const f32x2 a2 = ax * ax * -1.4426950408889634f;
    const f32x2 e = {__builtin_amdgcn_exp2f(a2[0]), __builtin_amdgcn_exp2f(a2[1])};
    y = 1.0f - y * t * e;
    return f32x2{copysignf(y[0], x[0]), copysignf(y[1], x[1])};
}
// value * gelu(gate) for two (value, gate) pairs
__device__ __forceinline__ f32x2 moca_geglu2(f32x2 v, f32x2 g) {
    return v * (g * 0.5f) * (moca_erf2(g * 0.70710678118654752f) + 1.0f);
}

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
