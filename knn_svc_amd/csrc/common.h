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

// Zero-fill as an ORDINARY KERNEL on the caller's stream (bytes % 4 == 0).  Not hipMemsetAsync: with several generators in flight on
// several streams its fills were not ordered with the kernels that follow them on the same stream the way a kernel is (round 5:
// csrc/models.hip, zero_slots — run-to-run different samples in the dataset-mode pipeline until the memset became a kernel).
namespace {
__global__ __launch_bounds__(256) void kn_zero_kernel(unsigned* __restrict__ p, long n) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = 0u;
}
}  // namespace
static inline int kn_zero_async(void* p, size_t bytes, hipStream_t st) {
    const long n = (long)(bytes / 4);
    if (n == 0) return KNNSVC_OK;
    hipLaunchKernelGGL(kn_zero_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (unsigned*)p, n);
    return knnsvc_check_launch("zero_fill");
}

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

// Two activations per instruction where the ISA has a packed fp32 form (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32; the two
// reciprocals and the two exponentials stay scalar): every component goes through exactly the roundings of kn_gelu, so the result
// is bit-identical to it — a change of instruction selection only (the conv stack's first layer evaluates 1e9 of these per step
// and is VALU-bound).
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 kn_gelu2(f32x2 x) {
#ifdef KN_GELU_ERFF
    return (f32x2){kn_gelu(x[0]), kn_gelu(x[1])};
#else
#pragma clang fp contract(off)
    const f32x2 ax = {fabsf(x[0]), fabsf(x[1])};
    const f32x2 z = ax * 0.70710678118654752440f;
    const f32x2 u = __builtin_elementwise_fma((f32x2)(0.39f), z, (f32x2)(1.0f));
    const f32x2 t = {__builtin_amdgcn_rcpf(u[0]), __builtin_amdgcn_rcpf(u[1])};
    f32x2 p = (f32x2)(-0.1137684788031453f);
    p = __builtin_elementwise_fma(p, t, (f32x2)(0.4433318425354039f));
    p = __builtin_elementwise_fma(p, t, (f32x2)(-0.3176995829519751f));
    p = __builtin_elementwise_fma(p, t, (f32x2)(0.32477925011974735f));
    p = __builtin_elementwise_fma(p, t, (f32x2)(0.045694387408764885f));
    p = __builtin_elementwise_fma(p, t, (f32x2)(0.11766258084307574f));
    p = p * t;
    const f32x2 ea = (-(z * z)) * 1.4426950408889634f;
    const f32x2 h = p * (f32x2){__builtin_amdgcn_exp2f(ea[0]), __builtin_amdgcn_exp2f(ea[1])};      // Phi(-|x|)
    const f32x2 omh = 1.0f - h;
    return x * (f32x2){x[0] >= 0.f ? omh[0] : h[0], x[1] >= 0.f ? omh[1] : h[1]};
#endif
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// Wave-wide sum on the VALU: four DPP adds fold each row of 16 lanes (xor 1, xor 2, half-mirror, mirror), four readlanes pick
// the row sums.  ds_bpermute-based __shfl_xor trees cost six dependent LDS round trips per value — in a frame-sequential
// kernel that is latency on the critical path of every frame.  Every lane gets the total.
__device__ __forceinline__ float wave_sum_dpp(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));   // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));   // row_mirror
    const int b = __builtin_bit_cast(int, v);
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0)), r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32)), r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48));
    return (r0 + r1) + (r2 + r3);
}
// the same fold for a double (two 32-bit DPP moves per step)
__device__ __forceinline__ double wave_sum_d_dpp(double v) {
#define KN_DPP_D(CTRL)                                                                                          \
    {                                                                                                          \
        const long b = __builtin_bit_cast(long, v);                                                            \
        const int lo = __builtin_amdgcn_mov_dpp((int)b, CTRL, 0xF, 0xF, true), hi = __builtin_amdgcn_mov_dpp((int)(b >> 32), CTRL, 0xF, 0xF, true); \
        v += __builtin_bit_cast(double, ((long)hi << 32) | (unsigned)lo);                                      \
    }
    KN_DPP_D(0xB1) KN_DPP_D(0x4E) KN_DPP_D(0x141) KN_DPP_D(0x140)
#undef KN_DPP_D
    const long b = __builtin_bit_cast(long, v);
    const int lo = (int)b, hi = (int)(b >> 32);
    double r[4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
        r[k] = __builtin_bit_cast(double, ((long)__builtin_amdgcn_readlane(hi, 16 * k) << 32) | (unsigned)__builtin_amdgcn_readlane(lo, 16 * k));
    return (r[0] + r[1]) + (r[2] + r[3]);
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
