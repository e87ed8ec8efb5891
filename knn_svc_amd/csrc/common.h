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
