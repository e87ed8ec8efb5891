// Shared host/device helpers for libknnsvc_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/knnsvc_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

extern thread_local char g_knnsvc_err[512];

static inline int knnsvc_fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_knnsvc_err, sizeof(g_knnsvc_err), fmt, ap);
    va_end(ap);
    return code;
}

#define KN_REQUIRE(cond, ...) do { if (!(cond)) return knnsvc_fail(KNNSVC_EINVAL, __VA_ARGS__); } while (0)

static inline int knnsvc_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return knnsvc_fail(KNNSVC_EHIP, "%s: %s", what, hipGetErrorString(e));
    return KNNSVC_OK;
}

static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// GELU(x) = x Phi(x) (torch.nn.functional.gelu, exact erf form: wavlm/modules.py, WavLM.py feature extractor and FFN).
// Phi(-|x|) = 0.5 erfc(|x| / sqrt 2) = P(t) exp(-x^2 / 2), t = 1 / (1 + p |x| / sqrt 2), P a degree-6 polynomial (minimax fit of
// erfc(z) exp(z^2), max error 8e-9 in exact arithmetic; the family of Abramowitz & Stegun 7.1.26); Phi(|x|) = 1 - Phi(-|x|).
// 15 VALU instructions, two of them transcendental (v_rcp_f32, v_exp_f32), against ~45 for the library erff — the step evaluates
// 2.8e9 GELUs (conv feature extractor, FFN1).  Accuracy in fp32 against the exact function: rms 6e-8, max 3.8e-7 over [-8, 8]
// — torch's own fp32 GELU: rms 1.4e-7, max 1.1e-6 — and the negative tail keeps its relative accuracy (no 1 + erf
// cancellation).  -DKN_GELU_ERFF restores the library form.
__device__ __forceinline__ float kn_gelu(float x) {
#ifdef KN_GELU_ERFF
    return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
#else
    // contraction off: whether `1 - p * e` became one fma used to depend on the code around the call, so the same activation could
    // differ by an ulp between two kernels (or two epilogue variants of one kernel); every rounding below is now spelled out
#pragma clang fp contract(off)
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.39f, z, 1.0f));
    float p = -0.1137684788031453f;                       // 0.5 x the fitted coefficients
    p = fmaf(p, t, 0.4433318425354039f);
    p = fmaf(p, t, -0.3176995829519751f);
    p = fmaf(p, t, 0.32477925011974735f);
    p = fmaf(p, t, 0.045694387408764885f);
    p = fmaf(p, t, 0.11766258084307574f);
    p *= t;
    const float h = p * __builtin_amdgcn_exp2f(-(z * z) * 1.4426950408889634f);      // Phi(-|x|)
    return x * (x >= 0.f ? 1.0f - h : h);
#endif
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
