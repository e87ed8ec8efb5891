// GPU f0 front end.  The reference calls pyworld.harvest(x, sr, f0_floor = 65, f0_ceil = 1047,
// frame_period = 20 ms) and zeroes values below 80 Hz (ddsp_prematch_dataset.py:121-128) when `<stem>_f0.npy` is missing.
// pyworld is not available offline, so this is NOT a restatement of Harvest: it is a YIN estimator (de Cheveigné &
// Kawahara 2002: difference function, cumulative-mean normalisation, absolute threshold, parabolic refinement) with
// Harvest's interface — same frame positions (t * hop samples), same search range, same "< 80 Hz -> 0" rule — so that the
// path runs without an f0 cache.  PARITY UNPINNED against pyworld; validated against synthetic clips of known f0.
//
// One block per frame: the window (W + TAU_MAX samples centred on the frame position, zero outside the signal) sits in
// LDS, thread tau computes d(tau) = sum_{j < W} (x[j] - x[j + tau])^2 (x[j] is a broadcast read, x[j + tau] is conflict-free),
// one wave turns d into the cumulative-mean-normalised d' and picks the first local minimum below the threshold.
#include "common.h"

namespace {

constexpr int F0_W = 768;          // integration window (48 ms at 16 kHz: > 3 periods at 65 Hz)
constexpr int F0_TMAX = 256;       // lags 1 .. 255 (16 kHz / 65 Hz = 246)

__global__ __launch_bounds__(256) void f0_yin_kernel(const float* __restrict__ x, long L, int hop, float sr, int tau_min, int tau_max,
                                                    float threshold, float voiced_bound, float zero_below, float* __restrict__ f0, long n_frames) {
    __shared__ float xs[F0_W + F0_TMAX];
    __shared__ float d[F0_TMAX];
    __shared__ float energy_s;
    const long t = blockIdx.x;
    const int tid = threadIdx.x;
    const long start = t * hop - (F0_W + F0_TMAX) / 2;
    for (int i = tid; i < F0_W + F0_TMAX; i += 256) {
        const long p = start + i;
        xs[i] = (p >= 0 && p < L) ? x[p] : 0.f;
    }
    __syncthreads();
    float acc = 0.f;
    if (tid >= 1 && tid <= tau_max) {
        for (int j = 0; j < F0_W; ++j) { const float v = xs[j] - xs[j + tid]; acc = fmaf(v, v, acc); }
    }
    if (tid == 0) { float e = 0.f; for (int j = 0; j < F0_W; ++j) e = fmaf(xs[j], xs[j], e); energy_s = e; }
    d[tid] = acc;
    __syncthreads();
    if (tid == 0) {
        // cumulative-mean-normalised difference, first dip below the threshold, else unvoiced
        float run = 0.f, gbest = 1e30f;
        int pick = -1, gpick = -1;
        float prev2 = 1.f, prev1 = 1.f;               // d'(tau - 2), d'(tau - 1)
        for (int tau = 1; tau <= tau_max; ++tau) {
            run += d[tau];
            const float dn = run > 0.f ? d[tau] * (float)tau / run : 1.f;
            const bool dip = tau - 1 >= tau_min && prev1 <= prev2 && prev1 <= dn;       // local minimum at tau - 1
            if (pick < 0 && dip && prev1 < threshold) pick = tau - 1;
            if (dip && prev1 < gbest) { gbest = prev1; gpick = tau - 1; }
            prev2 = prev1; prev1 = dn;
        }
        // real voices (vibrato, breath) often never dip below the absolute threshold: fall back to the deepest dip when it is
        // still clearly periodic (YIN step 4 with a looser voicing bound)
        if (pick < 0 && gpick > 0 && gbest < voiced_bound) pick = gpick;
        float out = 0.f;
        if (pick > 0 && energy_s > 1e-10f * (float)F0_W) {       // digital silence only: Harvest voices very quiet frames too
            // parabolic refinement on the raw difference function around the picked lag
            const float a = d[pick - 1], b = d[pick], c = d[pick + 1 <= tau_max ? pick + 1 : pick];
            const float den = a - 2.f * b + c;
            const float shift = den > 0.f ? 0.5f * (a - c) / den : 0.f;
            const float tau_f = (float)pick + fminf(fmaxf(shift, -0.5f), 0.5f);
            out = sr / tau_f;
            if (out < zero_below) out = 0.f;           // f0[f0 < 80] *= 0  (ddsp_prematch_dataset.py:126)
        }
        f0[t] = out;
    }
}

}  // namespace

extern "C" int knnsvc_f0_yin(const float* x, int64_t L, int32_t sample_rate, int32_t hop, float f0_floor, float f0_ceil,
                             float threshold, float zero_below, float* f0, int64_t n_frames, void* stream) {
    KN_REQUIRE(x && f0 && L > 0 && n_frames >= 0, "f0_yin: bad arguments");
    KN_REQUIRE(sample_rate > 0 && hop > 0 && f0_floor > 0.f && f0_ceil > f0_floor && threshold > 0.f, "f0_yin: bad parameters");
    const int tau_min = (int)((float)sample_rate / f0_ceil), tau_max = (int)((float)sample_rate / f0_floor) + 1;
    KN_REQUIRE(tau_min >= 2 && tau_max < F0_TMAX, "f0_yin: search range needs sample_rate / f0_floor < 255 lags");
    if (n_frames == 0) return KNNSVC_OK;
    hipLaunchKernelGGL(f0_yin_kernel, dim3((unsigned)n_frames), dim3(256), 0, (hipStream_t)stream, x, (long)L, hop, (float)sample_rate,
                       tau_min, tau_max, threshold, 3.0f * threshold, zero_below, f0, (long)n_frames);
    return knnsvc_check_launch("f0_yin");
}
