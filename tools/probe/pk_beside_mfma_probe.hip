// Probe: are packed-fp32 VALU results (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32) of one workgroup perturbed by MFMA-issuing
// workgroups that share its CU?  (round 3: concat_reselect_pipe_kernel's SLP-vectorised distance sums changed run to run beside
// the generator's convolutions; csrc/Makefile builds the library with -fno-slp-vectorize since.)
// victim: ONE workgroup of 512 threads walks LDS rows the way that kernel's distance phase did (six 16-byte reads, differences,
// squares, sums), once with float2 arithmetic (the compiler emits v_pk_*) and once with scalar arithmetic, R rounds, and writes
// both sums of every round.  runner: workgroups of 256 threads looping v_mfma_f32_32x32x16_f16, enough of them for every CU.
// The victim runs alone (reference) and then N times with the runner on a second stream; sums are compared bit for bit.
//   hipcc --offload-arch=gfx950 -O3 tools/probe/pk_beside_mfma_probe.hip -o tools/probe/pk_beside_mfma_probe && tools/probe/pk_beside_mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
constexpr int D = 1024, ROWS = 24, R = 1500;

__global__ __launch_bounds__(256) void runner(float* sink, int iters) {
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (threadIdx.x + i)); b[i] = (_Float16)(0.002f * (i + 1)); }
    f16v c0, c1;
    for (int i = 0; i < 16; ++i) { c0[i] = 0.f; c1[i] = 0.f; }
    for (int it = 0; it < iters; ++it) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, c1, 0, 0, 0);
    }
    if (c0[0] + c1[3] == 123.456f) sink[0] = c0[1];
}

template <bool PACKED>
__device__ __forceinline__ void sums(const float* L, int lane, int r0, float (&out)[5]) {
#pragma clang fp contract(off)
    const float* cv = L + ((r0 + 0) % ROWS) * D; const float* qv = L + ((r0 + 1) % ROWS) * D;
    const float* p0 = L + ((r0 + 2) % ROWS) * D; const float* p1 = L + ((r0 + 3) % ROWS) * D;
    const float* p2 = L + ((r0 + 4) % ROWS) * D; const float* p3 = L + ((r0 + 5) % ROWS) * D;
    if (PACKED) {
        f2 a[5] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
        for (int c = lane * 4; c < D; c += 256) {
            const f4 x = *(const f4*)(cv + c), q = *(const f4*)(qv + c), y0 = *(const f4*)(p0 + c), y1 = *(const f4*)(p1 + c),
                     y2 = *(const f4*)(p2 + c), y3 = *(const f4*)(p3 + c);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const f2 xx = {x[2 * h], x[2 * h + 1]};
                f2 d;
                d = (f2){q[2 * h], q[2 * h + 1]} - xx; a[0] += d * d;
                d = (f2){y0[2 * h], y0[2 * h + 1]} - xx; a[1] += d * d;
                d = (f2){y1[2 * h], y1[2 * h + 1]} - xx; a[2] += d * d;
                d = (f2){y2[2 * h], y2[2 * h + 1]} - xx; a[3] += d * d;
                d = (f2){y3[2 * h], y3[2 * h + 1]} - xx; a[4] += d * d;
            }
        }
        for (int k = 0; k < 5; ++k) out[k] = a[k][0] + a[k][1];
    } else {
        float a[5][2] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
        for (int c = lane * 4; c < D; c += 256) {
            const f4 x = *(const f4*)(cv + c), q = *(const f4*)(qv + c), y0 = *(const f4*)(p0 + c), y1 = *(const f4*)(p1 + c),
                     y2 = *(const f4*)(p2 + c), y3 = *(const f4*)(p3 + c);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float d;
                d = q[e] - x[e]; a[0][e & 1] += d * d;
                d = y0[e] - x[e]; a[1][e & 1] += d * d;
                d = y1[e] - x[e]; a[2][e & 1] += d * d;
                d = y2[e] - x[e]; a[3][e & 1] += d * d;
                d = y3[e] - x[e]; a[4][e & 1] += d * d;
            }
        }
        for (int k = 0; k < 5; ++k) out[k] = a[k][0] + a[k][1];
    }
}

__global__ __launch_bounds__(512) void victim(const float* src, float* out_pk, float* out_sc) {
    extern __shared__ __attribute__((aligned(16))) float L[];
    for (int i = threadIdx.x; i < ROWS * D; i += 512) L[i] = src[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int r = 0; r < R; ++r) {
        float pk[5], sc[5];
        sums<true>(L, lane, r + wave, pk);
        sums<false>(L, lane, r + wave, sc);
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            out_pk[((long)r * 512 + threadIdx.x) * 5 + k] = pk[k];
            out_sc[((long)r * 512 + threadIdx.x) * 5 + k] = sc[k];
        }
        __syncthreads();
    }
}

int main(int argc, char** argv) {
    const int runs = argc > 1 ? atoi(argv[1]) : 30;
    const size_t n = (size_t)R * 512 * 5;
    std::vector<float> h(ROWS * D);
    unsigned s = 12345u;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (float)((int)(s >> 9) - (1 << 22)) / (float)(1 << 22); }
    float *src, *opk, *osc, *sink;
    hipMalloc(&src, h.size() * 4); hipMalloc(&opk, n * 4); hipMalloc(&osc, n * 4); hipMalloc(&sink, 4);
    hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void*)victim, hipFuncAttributeMaxDynamicSharedMemorySize, ROWS * D * 4);
    hipStream_t s0, s1; hipStreamCreate(&s0); hipStreamCreate(&s1);
    std::vector<float> rpk(n), rsc(n), pk(n), sc(n);
    hipLaunchKernelGGL(victim, dim3(1), dim3(512), ROWS * D * 4, s0, src, opk, osc);
    hipStreamSynchronize(s0);
    hipMemcpy(rpk.data(), opk, n * 4, hipMemcpyDeviceToHost); hipMemcpy(rsc.data(), osc, n * 4, hipMemcpyDeviceToHost);
    size_t pk_vs_sc = 0;
    for (size_t i = 0; i < n; ++i) pk_vs_sc += memcmp(&rpk[i], &rsc[i], 4) != 0;
    int quiet_bad = 0, bad_pk_runs = 0, bad_sc_runs = 0; size_t bad_pk = 0, bad_sc = 0;
    for (int rep = 0; rep < runs + 1; ++rep) {
        const bool with = rep > 0;
        if (with) hipLaunchKernelGGL(runner, dim3(256 * 6), dim3(256), 0, s1, sink, 60000);
        hipLaunchKernelGGL(victim, dim3(1), dim3(512), ROWS * D * 4, s0, src, opk, osc);
        hipDeviceSynchronize();
        hipMemcpy(pk.data(), opk, n * 4, hipMemcpyDeviceToHost); hipMemcpy(sc.data(), osc, n * 4, hipMemcpyDeviceToHost);
        size_t dp = 0, ds = 0;
        for (size_t i = 0; i < n; ++i) { dp += memcmp(&pk[i], &rpk[i], 4) != 0; ds += memcmp(&sc[i], &rsc[i], 4) != 0; }
        if (!with) quiet_bad += (dp + ds) != 0;
        else { bad_pk_runs += dp != 0; bad_sc_runs += ds != 0; bad_pk += dp; bad_sc += ds; }
    }
    printf("{\"runs\": %d, \"quiet_repeat_equal\": %s, \"packed_runs_differ\": %d, \"scalar_runs_differ\": %d, \"packed_values_differ\": %zu, "
           "\"scalar_values_differ\": %zu, \"values_per_run\": %zu, \"packed_vs_scalar_differ_quiet\": %zu}\n",
           runs, quiet_bad ? "false" : "true", bad_pk_runs, bad_sc_runs, bad_pk, bad_sc, n, pk_vs_sc);
    return 0;
}
