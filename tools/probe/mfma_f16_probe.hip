// Probe: does v_mfma_f32_32x32x16_f16 keep fp16 subnormal inputs?  (decides the f16x2 split design)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
__global__ void k(float a, float b, float* out) {
    h8 A, B;
    for (int i = 0; i < 8; ++i) { A[i] = (_Float16)a; B[i] = (_Float16)b; }
    f16v c; for (int i = 0; i < 16; ++i) c[i] = 0.f;
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, c, 0, 0, 0);
    if (threadIdx.x == 0) out[0] = c[0];
}
int main() {
    float* d; hipMalloc(&d, 4);
    const float as[] = {1.0f, 9.5367431640625e-07f /*2^-20*/, 5.9604644775390625e-08f /*2^-24*/, 3e-5f};
    for (float a : as) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, a, 1024.0f, d);
        float h; hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
        printf("a=%g (fp16 %g) x 1024 x 16 -> %g (expect %g)\n", a, (float)(_Float16)a, h, (float)(_Float16)a * 1024.f * 16.f);
    }
    return 0;
}
