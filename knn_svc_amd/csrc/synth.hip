// Additive sinusoidal synthesiser fused with the generator's sin_prenet convolution.
// Reference: ddsp_prematch_dataset.py:131-208 (get_bulk_dsp_choral, upsample, remove_above_nyquist),
// hifigan/ddsp_models.py:416,476 (sin_prenet), hifigan/ddsp_models_f0.py:348-356 (plain sine variant).
#include "common.h"

namespace {

// exclusive fp64 prefix over frames of hop * f0/sr (the reference's per-sample fp64 cumsum,
// regrouped by frame; the regrouping error is ~1e-12 cycles, far below the f32 cast that follows)
__global__ __launch_bounds__(1024) void frame_phase_kernel(const float* __restrict__ f0, long N, int hop, int sr,
                                                          double* __restrict__ ph) {
    __shared__ double part[1024];
    const int tid = threadIdx.x;
    const long per = (N + 1023) / 1024;
    const long lo = tid * per, hi = lo + per < N ? lo + per : N;
    double s = 0.0;
    for (long i = lo; i < hi; ++i) s += (double)hop * ((double)f0[i] / (double)sr);
    part[tid] = s;
    __syncthreads();
    if (tid == 0) {
        double run = 0.0;
        for (int i = 0; i < 1024; ++i) { const double t = part[i]; part[i] = run; run += t; }
    }
    __syncthreads();
    double run = part[tid];
    for (long i = lo; i < hi; ++i) { ph[i] = run; run += (double)hop * ((double)f0[i] / (double)sr); }
}

__device__ __forceinline__ float cubic1(float x) { const float A = -0.75f; return ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f; }
__device__ __forceinline__ float cubic2(float x) { const float A = -0.75f; return ((A * x - 5.f * A) * x + 8.f * A) * x - 4.f * A; }

// one block per frame; hop + 2 excitation samples (one halo sample each side for the k=3 prenet)
__global__ __launch_bounds__(320) void additive_synth_kernel(const float* __restrict__ f0, const float* __restrict__ amp,
                                                            long N, int H, int hop, int sr, int mode,
                                                            const float* __restrict__ pw, const float* __restrict__ pb,
                                                            int n_ch, float* __restrict__ cond, int ld_cond,
                                                            float* __restrict__ exc, const double* __restrict__ fph,
                                                            const int* __restrict__ n_dyn) {
#pragma clang fp contract(off)
    extern __shared__ float sm[];
    float* a5 = sm;                 // [5][H] amplitude rows clamp(n-2 .. n+2)
    float* y = sm + 5 * H;          // [hop + 2]
    const long n = blockIdx.x;
    if (n_dyn) N = *n_dyn < N ? *n_dyn : N;     // valid frames of a launch sized for a bucket: edge clamps and the last sample follow it
    if (n >= N) {            // a frame of the bucket past the valid length: defined (zero) condition rows, nothing else
        for (int j = threadIdx.x; j < hop; j += blockDim.x) {
            const long t = n * hop + j;
            if (exc) exc[t] = 0.f;
            float* cp = cond + t * (long)ld_cond;
            for (int c = 0; c < n_ch; ++c) cp[c] = 0.f;
        }
        return;
    }
    const long L = N * hop;
    if (mode == 0)
        for (int i = threadIdx.x; i < 5 * H; i += blockDim.x) {
            long r = n - 2 + i / H;
            r = r < 0 ? 0 : (r > N - 1 ? N - 1 : r);
            a5[i] = amp[r * H + (i % H)];
        }
    __syncthreads();
    const float scale = (float)N / (float)L;                 // area_pixel_compute_scale, align_corners=False
    for (int jj = threadIdx.x; jj < hop + 2; jj += blockDim.x) {
        const long t = n * hop + jj - 1;                      // global sample index
        float v = 0.f;
        if (t >= 0 && t < L) {
            const long m = t / hop; const int j = (int)(t - m * hop);
            const float f = f0[m];
            const double ph = fph[m] + (double)(j + 1) * ((double)f / (double)sr);
            const float w = (float)(2.0 * 3.14159265358979323846 * (ph - rint(ph)));
            if (mode == 1) v = sinf(w);
            else {
                // torch evaluates area_pixel_compute_source_index (scale * (dst + 0.5) - 0.5) with ONE rounding: its CPU
                // and CUDA builds both contract the expression into an fma.  Two roundings move tx by up to an ulp of the
                // frame index — 3e-5 at frame 512, 6e-5 at 1024 — and with it the cubic weights (2e-5 in the waveform).
                const float rx = __builtin_fmaf(scale, (float)t + 0.5f, -0.5f);
                const float fl = floorf(rx);
                const float tx = rx - fl;
                const long ix = (long)fl;
                const float c0 = cubic2(tx + 1.f), c1 = cubic1(tx), c2 = cubic1(1.f - tx), c3 = cubic2((1.f - tx) + 1.f);
                int r[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    long rr = ix - 1 + e;                      // source frame of this tap, edge clamped
                    rr = rr < 0 ? 0 : (rr > N - 1 ? N - 1 : rr);
                    long slot = rr - (n - 2);                  // LDS slot s holds frame clamp(n-2+s)
                    slot = slot < 0 ? 0 : (slot > 4 ? 4 : slot);
                    r[e] = (int)slot * H;
                }
                for (int k = 0; k < H; ++k) {
                    const float a = ((c0 * a5[r[0] + k] + c1 * a5[r[1] + k]) + c2 * a5[r[2] + k]) + c3 * a5[r[3] + k];
                    const float msk = ((f * (float)(k + 1)) < (float)sr / 2.f ? 1.0f : 0.0f) + 1e-7f;
                    v += sinf(w * (float)(k + 1)) * (a * msk);
                }
            }
        }
        y[jj] = v;
    }
    __syncthreads();
    for (int j = threadIdx.x; j < hop; j += blockDim.x) {
        const long t = n * hop + j;
        const float ym = y[j], y0 = y[j + 1], yp = y[j + 2];
        if (exc) exc[t] = y0;
        float* cp = cond + t * (long)ld_cond;
        for (int c = 0; c < n_ch; ++c) cp[c] = ((pw[c * 3] * ym + pw[c * 3 + 1] * y0) + pw[c * 3 + 2] * yp) + pb[c];
    }
}

}  // namespace

extern "C" int knnsvc_additive_synth(const float* f0, const float* amp, int64_t N, int32_t H, int32_t hop,
                                     int32_t sample_rate, int32_t mode, const float* prenet_w, const float* prenet_b,
                                     int32_t n_ch, float* cond, int32_t ld_cond, float* exc, double* frame_phase,
                                     const int32_t* n_dyn, void* stream) {
    KN_REQUIRE(f0 && prenet_w && prenet_b && cond && frame_phase, "additive_synth: null pointer");
    KN_REQUIRE(mode == 1 || amp, "additive_synth: amp required in additive mode");
    KN_REQUIRE(N > 0 && hop > 0 && sample_rate > 0 && n_ch > 0 && ld_cond >= n_ch, "additive_synth: bad sizes");
    KN_REQUIRE(mode == 1 || (H > 0 && H <= 1024), "additive_synth: bad harmonic count");
    KN_REQUIRE(N * (long)hop < (1L << 31), "additive_synth: sequence too long");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(frame_phase_kernel, dim3(1), dim3(1024), 0, st, f0, (long)N, hop, sample_rate, frame_phase);
    int rc = knnsvc_check_launch("frame_phase");
    if (rc) return rc;
    const int Hs = mode == 0 ? H : 0;
    const size_t lds = (size_t)(5 * Hs + hop + 2) * 4;
    hipLaunchKernelGGL(additive_synth_kernel, dim3((unsigned)N), dim3(320), lds, st, f0, amp, (long)N, Hs, hop,
                       sample_rate, mode, prenet_w, prenet_b, n_ch, cond, ld_cond, exc, (const double*)frame_phase, n_dyn);
    return knnsvc_check_launch("additive_synth");
}
