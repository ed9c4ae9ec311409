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

// current value of a MOCA_TUNE_* knob (runtime.hip; set through moca_set_tuning)
int moca_tuning_get(int knob);

// gemm_ws.hip: the weight-stationary streaming kernel of the 320 -> 320 linears (dispatched by moca_gemm_f16)
bool moca_gemm_ws_ok(const moca_gemm_params& p);
int moca_gemm_ws_launch(const moca_gemm_params& p, hipStream_t st);

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
// value * gelu(gate) for two (value, gate) pairs.  gelu(g) = g * Phi(g) with Phi from the same A&S 7.1.26 erf, written so that no
// sign has to be restored:  for x = |g| / sqrt(2), t = 1 / (1 + p x), q = (a1 t + ... + a5 t^5) exp(-x^2) = 1 - erf(x):
//     gelu(g) = max(g, 0) - |g| q / 2 = g / 2 + |g| (1 - q) / 2
// (g >= 0: g (1 - q / 2) = g (1 + erf) / 2;  g < 0: g q / 2 = g (1 - erf(|x|)) / 2).  The 1/2 is folded into the coefficients, the
// 1/sqrt(2) into p, exp(-x^2) = exp2(-(g c)^2) with c = sqrt(log2(e) / 2): 12 plain + 2 transcendental instructions per output in
// scalar form, 9 issue slots in the packed form below (the round-4 form took 18.5: copysign, 1 - y, (erf + 1), g / 2 and the moves
// the packed operands needed).
#ifdef MOCA_GELU_OLD
__device__ __forceinline__ f32x2 moca_geglu2(f32x2 v, f32x2 g) {
    return v * (g * 0.5f) * (moca_erf2(g * 0.70710678118654752f) + 1.0f);
}
#else
__device__ __forceinline__ f32x2 moca_geglu2(f32x2 v, f32x2 g) {
    const f32x2 ag = {fabsf(g[0]), fabsf(g[1])};
    const f32x2 d = ag * (0.3275911f * 0.70710678118654752f) + 1.0f;
    const f32x2 t = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
    f32x2 y = t * (0.5f * 1.061405429f) - (0.5f * 1.453152027f);
    y = y * t + (0.5f * 1.421413741f);
    y = y * t - (0.5f * 0.284496736f);
    y = y * t + (0.5f * 0.254829592f);
    y = y * t;
    const f32x2 u = g * 0.84932180028801904f;                    // sqrt(log2(e) / 2)
    const f32x2 s = -u * u;
    const f32x2 e = {__builtin_amdgcn_exp2f(s[0]), __builtin_amdgcn_exp2f(s[1])};
    // max(g, 0) - |g| y e = g / 2 + |g| (1 / 2 - y e)      ((g + |g|) / 2 is exact; no v_max, which costs two instructions under IEEE mode)
    const f32x2 w = 0.5f - y * e;
    return v * (g * 0.5f + ag * w);
}
#endif
// four (value, gate) pairs at once: the same arithmetic on 4-vectors, so that two independent packed chains sit next to each other in
// program order and fill each other's dependent-issue slots (the 2-wide form leaves an s_nop behind every step of the Horner chain)
typedef float f32x4g __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4g moca_geglu4(f32x4g v, f32x4g g) {
    const f32x4g ag = {fabsf(g[0]), fabsf(g[1]), fabsf(g[2]), fabsf(g[3])};
    const f32x4g d = ag * (0.3275911f * 0.70710678118654752f) + 1.0f;
    const f32x4g t = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1]), __builtin_amdgcn_rcpf(d[2]), __builtin_amdgcn_rcpf(d[3])};
    // yn = -(a1 t + ... + a5 t^5) / 2 (negated coefficients: 1 / 2 + yn e needs no sign flip, which the 4-wide form would do with v_xor)
    float c1n = -0.5f * 1.061405429f;
    asm("" : "+s"(c1n));                                         // (opaque: a visible negative factor is rewritten as (-t) * |c|, i.e. 4 v_xor)
    f32x4g yn = t * c1n + (0.5f * 1.453152027f);
    yn = yn * t - (0.5f * 1.421413741f);
    yn = yn * t + (0.5f * 0.284496736f);
    yn = yn * t - (0.5f * 0.254829592f);
    yn = yn * t;
    const f32x4g u = g * 0.84932180028801904f;
    const f32x4g s = u * u;                                      // (the sign rides on v_exp_f32's source modifier)
    const f32x4g e = {__builtin_amdgcn_exp2f(-s[0]), __builtin_amdgcn_exp2f(-s[1]), __builtin_amdgcn_exp2f(-s[2]), __builtin_amdgcn_exp2f(-s[3])};
    const f32x4g w = yn * e + 0.5f;
    return v * (g * 0.5f + ag * w);
}

// GroupNorm statistics accumulated across blocks (MOCA_EP_GSTAT, concat with statistics): 64-bit FIXED-POINT atomics -- integer
// addition is associative, so the finished statistics do not depend on the order in which the producer's blocks arrive and a
// replayed graph reproduces its output bit for bit by construction (f64 atomicAdd did so only "in practice").
// Units (round 6; 2^-30 / 2^-16 before): sum 2^-20, sum of squares 2^-12.  What they must hold: one PARTIAL (a row tile x a
// channel group, <= 320 x 40 = 12 800 values) is accepted up to 2^53 units, a group's TOTAL up to 2^61 units:
//     sum:            |partial| < 8.6e9 (a mean of 6.7e5 over 12 800 values), |total| < 2.2e12      -- beyond fp16 (max 65 504)
//     sum of squares:  partial  < 2.2e12 (rms 1.3e4 over 12 800 values),      total  < 5.6e14 (rms 3.7e4 over the largest group,
//                      409 600 values): a tensor at that rms already holds fp16 infinities.
//   The old units stopped at a partial rms of 3.3e3 -- reached by the latents of bench.py's synthetic video (~4e3), whose statistics
//   were then CLAMPED, i.e. silently wrong.  Precision: rounding a partial to the units changes a group's mean by
//   < 2^-21 * partials / values and its mean of squares by < 2^-13 * partials / values <= 2^-13 / 400 = 3e-7 at worst (a 5 x 8 x
//   10-value partial), below the 1e-6 / 1e-5 eps added to the variance.
// A NON-FINITE partial (NaN / Inf activations) or one beyond +-2^53 units must not turn into finite statistics: it sets bit 62 of
// the group's sum-of-squares accumulator (acc layout [group][2] = {sum, sum of squares}) and adds nothing; OR is idempotent, so the
// bit survives however many partials are poisoned.  moca_gstat_get returns NaN when the sum-of-squares accumulator is >= 2^61 -- the
// poison bit, or accepted adds that came within a factor 2 of it (a group of the UNet receives <= 2^10 partials: rows of a statistics
// group / 160-row tiles x 2 column tiles, so accepted adds stay below 2^63 and cannot wrap): statistics outside the fixed-point
// range read back as NaN, never as a wrong finite number, and finite adds can not alias the poison bit unnoticed.
#define MOCA_GSTAT_SUM_SCALE 1048576.0        /* 2^20 */
#define MOCA_GSTAT_SQ_SCALE 4096.0            /* 2^12 */
#define MOCA_GSTAT_POISON (1ull << 62)
#define MOCA_GSTAT_MAX_UNITS 9007199254740992.0      /* 2^53 */
__device__ __forceinline__ void moca_gstat_add(int64_t* acc, int comp, float partial) {
    if (!(fabsf(partial) <= 3.0e38f)) {            // NaN or Inf
        atomicOr(reinterpret_cast<unsigned long long*>(acc + (1 - comp)), MOCA_GSTAT_POISON);
        return;
    }
    double v = (double)partial * (comp ? MOCA_GSTAT_SQ_SCALE : MOCA_GSTAT_SUM_SCALE);
    if (!(fabs(v) <= MOCA_GSTAT_MAX_UNITS)) {      // out of the fixed-point range: poison, never clamp (a clamped partial would read
        atomicOr(reinterpret_cast<unsigned long long*>(acc + (1 - comp)), MOCA_GSTAT_POISON);      // back as finite, WRONG statistics)
        return;
    }
    atomicAdd(reinterpret_cast<unsigned long long*>(acc), (unsigned long long)__double2ll_rn(v));
}
__device__ __forceinline__ double moca_gstat_get(const int64_t* acc, int comp) {
    if (comp && (unsigned long long)*acc >= (MOCA_GSTAT_POISON >> 1)) return __longlong_as_double(0x7ff8000000000000ll);
    return (double)*acc * (comp ? 1.0 / MOCA_GSTAT_SQ_SCALE : 1.0 / MOCA_GSTAT_SUM_SCALE);
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
