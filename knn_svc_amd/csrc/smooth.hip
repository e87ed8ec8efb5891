// Concatenation-smoothness weight optimisation, fully on the device.
// Reference: ddsp_prematch_dataset.py:574-680 (compute_wavlm_weight), :807-924
// (compute_extended_weight), :426-428 (softmax weights), :449-461 (the two MSE costs).
//
// The reference runs autograd + torch.optim.Adam(amsgrad) with a host sync per iteration and
// re-reads 3 x [N,4,D] gathered features every step.  The loss is a quadratic form in the
// softmax weights, so this implementation
//   1. reduces the features once to two 8x8 Gram matrices per adjacent frame pair
//      (gram_kernel; vectors are centred on their mean first — the weights of each frame sum
//      to one, so centring changes neither loss nor softmax gradient but removes cancellation);
//   2. runs the whole Adam loop — loss, analytic gradient, amsgrad update, best-iterate
//      tracking and the reference's three stopping rules — inside ONE persistent workgroup
//      (adam_kernel), O(N*128) flops per iteration and no host round trips.
#include "common.h"

namespace {

constexpr int KW = 4;           // neighbours per frame
constexpr int GE = 28;          // strict upper triangle of the symmetric 8x8 Gram matrix; the eight centred vectors sum
                                // to zero, so every row of G sums to zero and the diagonal is implied:
                                // (G c)_i = sum_{j != i} G_ij (c_j - c_i)

__device__ __forceinline__ int tri(int i, int j) { return i < j ? i * 7 - i * (i - 1) / 2 + (j - i - 1) : j * 7 - j * (j - 1) / 2 + (i - j - 1); }   // i != j

// block per frame pair t.  Both cost terms are quadratic forms in the SAME coefficient vector
// c = [w[t+1,:], -w[t,:]] (term a over rows F[idx[t+1,k]-1], F[idx[t,k]]; term b over F[idx[t+1,k]], F[idx[t,k]+1]),
// so only the sum of the two centred 8x8 Gram matrices is kept: its 28 off-diagonal entries per pair.
// row_scale (optional) [nq][4]: candidate k of frame t and its +-1 neighbours are multiplied by row_scale[t][k]
// (compute_weight_with_amp, ddsp_prematch_dataset.py:684-713); the centring argument is unchanged because the
// coefficient vector still sums to zero.
__global__ __launch_bounds__(256) void gram_kernel(const long* __restrict__ idx, long nq, const float* __restrict__ pool,
                                                  long np, int dim, int ld, const float* __restrict__ row_scale,
                                                  float* __restrict__ gram) {
    extern __shared__ float sm[];           // [8][dim] centred vectors of the current term
    const long t = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double acc[GE / 4];
#pragma unroll
    for (int q = 0; q < GE / 4; ++q) acc[q] = 0.0;
    float rs[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) rs[r] = row_scale ? (r < 4 ? row_scale[(t + 1) * KW + r] : row_scale[t * KW + (r - 4)]) : 1.f;
    for (int term = 0; term < 2; ++term) {
        __syncthreads();
        for (int c = tid; c < dim; c += 256) {
            float v[8], mean = 0.f;
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                long id;
                if (r < 4) id = idx[(t + 1) * KW + r] + (term == 0 ? -1 : 0);
                else id = idx[t * KW + (r - 4)] + (term == 0 ? 0 : 1);
                id = id < 0 ? 0 : (id > np - 1 ? np - 1 : id);
                v[r] = pool[id * (long)ld + c];
                if (row_scale) v[r] *= rs[r];
                mean += v[r];
            }
            mean *= 0.125f;
#pragma unroll
            for (int r = 0; r < 8; ++r) sm[r * dim + c] = v[r] - mean;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < GE / 4; ++q) {
            const int e = wave + 4 * q;
            int i = 0, rem = e;                       // unpack e -> (i < j)
            while (rem >= 7 - i) { rem -= 7 - i; ++i; }
            const int j = i + 1 + rem;
            double s = 0.0;
            for (int c = lane; c < dim; c += 64) s += (double)sm[i * dim + c] * (double)sm[j * dim + c];
            acc[q] += wave_sum_d(s);
        }
    }
    if (lane == 0) {
#pragma unroll
        for (int q = 0; q < GE / 4; ++q) gram[((long)(wave + 4 * q)) * nq + t] = (float)acc[q];
    }
}

struct LoopState { double min_loss, conv_min; int since; };

// The two per-element pieces of an iteration that were IEEE divides, square roots and expf: 250 of the 410 VALU instructions a
// frame costs per iteration (one CU runs the whole loop).  Inside the loop they run on the hardware's 1-ulp v_exp_f32 / v_rcp_f32 /
// v_sqrt_f32: 4.67 -> 2.65 us per iteration.  The optimiser's trajectory is only defined up to such roundings anyway (a different
// summation order of the loss already moves the weights by 2e-4), and measured against the reference-generated fixtures the two
// forms are level: weights max|d| 3.5e-4 / 2.4e-4 (IEEE) vs 2.9e-4 / 2.8e-4 (WavLM / harmonics, g5), 1.1e-6 vs 1.2e-6
// (amplitude-scaled), the same iteration counts (201 / 201 / 401), file-level waveform rms 5.4e-7 vs 1.2e-6 on the post_opt
// fixture (bar 1e-4), everything else unchanged.  The FINAL weights (softmax of the best iterate) keep expf and the IEEE divide.
__device__ __forceinline__ f32x4 loop_softmax4(f32x4 th, float mx) {
#pragma clang fp contract(off)
    f32x4 e;
#pragma unroll
    for (int k = 0; k < 4; ++k) e[k] = __builtin_amdgcn_exp2f((th[k] - mx) * 1.44269504088896341f);
    const float den = (e[0] + e[1]) + (e[2] + e[3]);       // >= 1: the maximum contributes exp2(0)
    return e * __builtin_amdgcn_rcpf(den);
}
// theta + step_size * m / (sqrt(vmax) / sqrt(1 - b2^t) + eps); rbc2 = 1 / sqrt(1 - b2^t), one IEEE divide per iteration
__device__ __forceinline__ float loop_step(float th, float m, float vmax, float rbc2, float eps, float step_size) {
#pragma clang fp contract(off)
    const float denom = __builtin_amdgcn_sqrtf(vmax) * rbc2 + eps;
    return th + step_size * (m * __builtin_amdgcn_rcpf(denom));
}


__global__ __launch_bounds__(1024) void adam_kernel(long nq, int dim, float scale, int max_iter,
                                                   const float* __restrict__ gram, float* __restrict__ state,
                                                   float* __restrict__ xch_global, int use_lds,
                                                   float* __restrict__ out_w, int* __restrict__ out_iters) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    __shared__ double red[2][16];
    float* xw = use_lds ? sm : xch_global;                 // [nq][4] softmax weights
    float* xg = xw + nq * KW;                              // [nq][4] gradient w.r.t. w from the pair (t-1, t)
    f32x4* theta = (f32x4*)state;
    f32x4* m1 = theta + nq; f32x4* v2 = m1 + nq; f32x4* vmax = v2 + nq; f32x4* best = vmax + nq;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const double S = (double)scale / ((double)dim * (double)(nq - 1));
    const float twoS = (float)(2.0 * S);
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    for (long t = tid; t < nq; t += 1024) { theta[t] = zero; m1[t] = zero; v2[t] = zero; vmax[t] = zero; best[t] = zero; }
    __syncthreads();

    double min_loss = 20000.0, conv_min = 20000.0;
    int since = 0, it = 0;
    const float b2 = 0.999f, omb1 = (float)(1.0 - 0.9), omb2 = (float)(1.0 - 0.999), eps = 1e-8f;
    double pb1 = 1.0, pb2 = 1.0;
    const bool forced = max_iter < 0;            // measurement mode: exactly |max_iter| iterations, stopping rules off
    const int cap = forced ? -max_iter : max_iter;
    for (it = 0; it < cap; ++it) {
        // ---- phase 1: softmax weights ---------------------------------------------------------
        for (long t = tid; t < nq; t += 1024) {
            const f32x4 th = theta[t];
            const float mx = fmaxf(fmaxf(th[0], th[1]), fmaxf(th[2], th[3]));
            *(f32x4*)&xw[t * KW] = loop_softmax4(th, mx);
        }
        __syncthreads();
        // ---- phase 2: per pair quadratic forms, gradient w.r.t. the weights ---------------------
        double lsum = 0.0;
        for (long t = tid; t < nq - 1; t += 1024) {
            const f32x4 w1 = *(const f32x4*)&xw[(t + 1) * KW], w0 = *(const f32x4*)&xw[t * KW];
            const float c[8] = {w1[0], w1[1], w1[2], w1[3], -w0[0], -w0[1], -w0[2], -w0[3]};
            float y[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            {
                float g[GE];
#pragma unroll
                for (int e = 0; e < GE; ++e) g[e] = gram[((long)e) * nq + t];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    float s = 0.f;
#pragma unroll
                    for (int j = 0; j < 8; ++j) if (j != i) s += g[tri(i, j)] * (c[j] - c[i]);
                    y[i] += s;
                }
            }
            float qf = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) qf += c[i] * y[i];
            lsum += (double)qf;
            f32x4 gf = {twoS * y[0], twoS * y[1], twoS * y[2], twoS * y[3]};
            *(f32x4*)&xg[(t + 1) * KW] = gf;                       // d/dw[t+1,:] from pair t
            // d/dw[t,:] from pair t is kept in the (unused) slot layout of the state pass below
            f32x4 gs = {-twoS * y[4], -twoS * y[5], -twoS * y[6], -twoS * y[7]};
            best[nq + t] = gs;                                     // scratch row after `best`
        }
        lsum = wave_sum_d_dpp(lsum);                  // two barriers per iteration: see adam_reg_kernel
        if (lane == 0) red[it & 1][wave] = lsum;
        __syncthreads();
        double tot = 0.0;
#pragma unroll
        for (int i = 0; i < 16; ++i) tot += red[it & 1][i];
        const float loss = (float)(S * tot);
        // ---- phase 3: the reference's loop control (uniform across the block) -------------------
        if (it % 100 == 1) {
            if (!forced && fabs(min_loss - conv_min) < 1e-5) break;
            conv_min = min_loss;
        }
        const bool improved = loss < (float)min_loss;
        if (improved) { min_loss = (double)loss; since = 0; } else ++since;
        if (!forced && since >= 1000) break;
        // ---- phase 4: gradient through softmax + Adam(amsgrad) ----------------------------------
        pb1 *= 0.9; pb2 *= 0.999;
        const float step_size = (float)(-(0.1 / (1.0 - pb1)));
        const float bc2_sqrt = (float)sqrt(1.0 - pb2);
        const float rbc2 = 1.0f / bc2_sqrt;
        for (long t = tid; t < nq; t += 1024) {
            const f32x4 w = *(const f32x4*)&xw[t * KW];
            f32x4 g = zero;
            if (t >= 1) g = *(const f32x4*)&xg[t * KW];
            if (t < nq - 1) g += best[nq + t];
            const float dotwg = (w[0] * g[0] + w[1] * g[1]) + (w[2] * g[2] + w[3] * g[3]);
            f32x4 gt = w * (g - dotwg);
            f32x4 th = theta[t];
            if (improved) best[t] = th;
            f32x4 mm = m1[t], vv = v2[t], vm = vmax[t];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                mm[k] = mm[k] + (gt[k] - mm[k]) * omb1;
                vv[k] = vv[k] * b2 + (omb2 * gt[k]) * gt[k];
                vm[k] = fmaxf(vm[k], vv[k]);
                th[k] = loop_step(th[k], mm[k], vm[k], rbc2, eps, step_size);
            }
            theta[t] = th; m1[t] = mm; v2[t] = vv; vmax[t] = vm;
        }
    }
    // softmax(best)
    __syncthreads();
    for (long t = tid; t < nq; t += 1024) {
        const f32x4 th = best[t];
        const float mx = fmaxf(fmaxf(th[0], th[1]), fmaxf(th[2], th[3]));
        f32x4 e = {expf(th[0] - mx), expf(th[1] - mx), expf(th[2] - mx), expf(th[3] - mx)};
        const float den = (e[0] + e[1]) + (e[2] + e[3]);
        *(f32x4*)&out_w[t * KW] = e / den;
    }
    if (tid == 0 && out_iters) out_iters[0] = it;
}

// Same loop with everything a thread needs for its (up to FPT) frames in registers: the 36-entry Gram of
// each frame pair, theta, Adam moments, amsgrad maximum and the best iterate.  Only the softmax weights and
// the gradient contribution of the left neighbour travel through LDS; an iteration costs three barriers and
// no global memory traffic (the global-state variant streams ~150 KB of Gram data per iteration through one CU).
template <int FPT>
__global__ __launch_bounds__(512) void adam_reg_kernel(long nq, int dim, float scale, int max_iter,
                                                      const float* __restrict__ gram, float* __restrict__ out_w,
                                                      int* __restrict__ out_iters) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    __shared__ double red[2][8];      // by iteration parity: a fast wave may write its next sum while a slow one still reads
    float* xw = sm;                       // [nq][4]
    float* xg = sm + nq * KW;             // [nq][4]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const double S = (double)scale / ((double)dim * (double)(nq - 1));
    const float twoS = (float)(2.0 * S);
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    float g[FPT][GE];
    f32x4 th[FPT], mm[FPT], vv[FPT], vm[FPT], best[FPT], gs[FPT];
#pragma unroll
    for (int f = 0; f < FPT; ++f) {
        const long t = tid + 512L * f;
        th[f] = mm[f] = vv[f] = vm[f] = best[f] = gs[f] = zero;
#pragma unroll
        for (int e = 0; e < GE; ++e) g[f][e] = (t < nq - 1) ? gram[(long)e * nq + t] : 0.f;
    }
    double min_loss = 20000.0, conv_min = 20000.0;
    int since = 0, it = 0;
    const float b2 = 0.999f, omb1 = (float)(1.0 - 0.9), omb2 = (float)(1.0 - 0.999), eps = 1e-8f;
    double pb1 = 1.0, pb2 = 1.0;
    const bool forced = max_iter < 0;            // measurement mode: exactly |max_iter| iterations, stopping rules off
    const int cap = forced ? -max_iter : max_iter;
    for (it = 0; it < cap; ++it) {
#pragma unroll
        for (int f = 0; f < FPT; ++f) {
            const long t = tid + 512L * f;
            if (t < nq) {
                const float mx = fmaxf(fmaxf(th[f][0], th[f][1]), fmaxf(th[f][2], th[f][3]));
                *(f32x4*)&xw[t * KW] = loop_softmax4(th[f], mx);
            }
        }
        __syncthreads();
        double lsum = 0.0;
#pragma unroll
        for (int f = 0; f < FPT; ++f) {
            const long t = tid + 512L * f;
            if (t < nq - 1) {
                const f32x4 w1 = *(const f32x4*)&xw[(t + 1) * KW], w0 = *(const f32x4*)&xw[t * KW];
                const float c[8] = {w1[0], w1[1], w1[2], w1[3], -w0[0], -w0[1], -w0[2], -w0[3]};
                // c[j] - c[i] once per pair: (c[i] - c[j]) is its exact negation (a free source modifier), the sums keep their order
                float dd[GE], y[8];
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int j = i + 1; j < 8; ++j) dd[tri(i, j)] = c[j] - c[i];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    float s = 0.f;
#pragma unroll
                    for (int j = 0; j < 8; ++j) if (j != i) s = fmaf(g[f][tri(i, j)], j > i ? dd[tri(i, j)] : -dd[tri(i, j)], s);
                    y[i] = s;
                }
                float qf = 0.f;
#pragma unroll
                for (int i = 0; i < 8; ++i) qf += c[i] * y[i];
                lsum += (double)qf;
                *(f32x4*)&xg[(t + 1) * KW] = (f32x4){twoS * y[0], twoS * y[1], twoS * y[2], twoS * y[3]};
                gs[f] = (f32x4){-twoS * y[4], -twoS * y[5], -twoS * y[6], -twoS * y[7]};
            }
            __builtin_amdgcn_sched_barrier(0);      // one frame at a time: keeps the matvec temporaries of different frames from coexisting
        }
        // Two barriers per iteration (round 3; four before): the wave sums are DPP folds instead of twelve ds_bpermute round
        // trips, every thread adds the eight wave sums itself (same order everywhere: the loss is uniform) instead of waiting
        // for thread 0, and the barrier that closed the iteration is gone — xw[t] / xg[t + 1] are only rewritten behind the
        // NEXT iteration's first barrier, which their last readers (this iteration's update phase) have to pass first.
        lsum = wave_sum_d_dpp(lsum);
        if (lane == 0) red[it & 1][wave] = lsum;
        __syncthreads();
        double tot = 0.0;
#pragma unroll
        for (int i = 0; i < 8; ++i) tot += red[it & 1][i];
        const float loss = (float)(S * tot);
        if (it % 100 == 1) {
            if (!forced && fabs(min_loss - conv_min) < 1e-5) break;
            conv_min = min_loss;
        }
        const bool improved = loss < (float)min_loss;
        if (improved) { min_loss = (double)loss; since = 0; } else ++since;
        if (!forced && since >= 1000) break;
        pb1 *= 0.9; pb2 *= 0.999;
        const float step_size = (float)(-(0.1 / (1.0 - pb1)));
        const float bc2_sqrt = (float)sqrt(1.0 - pb2);
        const float rbc2 = 1.0f / bc2_sqrt;
#pragma unroll
        for (int f = 0; f < FPT; ++f) {
            const long t = tid + 512L * f;
            if (t < nq) {
                const f32x4 w = *(const f32x4*)&xw[t * KW];
                f32x4 gg = zero;
                if (t >= 1) gg = *(const f32x4*)&xg[t * KW];
                if (t < nq - 1) gg += gs[f];
                const float dotwg = (w[0] * gg[0] + w[1] * gg[1]) + (w[2] * gg[2] + w[3] * gg[3]);
                const f32x4 gt = w * (gg - dotwg);
                if (improved) best[f] = th[f];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    mm[f][k] = mm[f][k] + (gt[k] - mm[f][k]) * omb1;
                    vv[f][k] = vv[f][k] * b2 + (omb2 * gt[k]) * gt[k];
                    vm[f][k] = fmaxf(vm[f][k], vv[f][k]);
                    th[f][k] = loop_step(th[f][k], mm[f][k], vm[f][k], rbc2, eps, step_size);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#pragma unroll
    for (int f = 0; f < FPT; ++f) {
        const long t = tid + 512L * f;
        if (t < nq) {
            const f32x4 b = best[f];
            const float mx = fmaxf(fmaxf(b[0], b[1]), fmaxf(b[2], b[3]));
            f32x4 e = {expf(b[0] - mx), expf(b[1] - mx), expf(b[2] - mx), expf(b[3] - mx)};
            const float den = (e[0] + e[1]) + (e[2] + e[3]);
            *(f32x4*)&out_w[t * KW] = e / den;
        }
    }
    if (tid == 0 && out_iters) out_iters[0] = it;
}

__global__ void fill_quarter_kernel(float* w, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) w[i] = 0.25f;
}

inline size_t ws_floats(long nq) { return (size_t)GE * nq + (size_t)6 * 4 * nq + (size_t)2 * 4 * nq; }

}  // namespace

extern "C" size_t knnsvc_smooth_workspace_bytes(int64_t nq) { return nq > 0 ? ws_floats(nq) * 4 + 64 : 0; }

extern "C" int knnsvc_smooth_weights(const int64_t* idx, int64_t nq, const float* pool, int64_t np, int32_t dim,
                                     int32_t ld, float scale, const float* row_scale, int32_t max_iter, float* out_w,
                                     int32_t* out_iters, void* workspace, size_t workspace_bytes, void* stream) {
    KN_REQUIRE(idx && pool && out_w && workspace, "smooth_weights: null pointer");
    KN_REQUIRE(nq > 0 && np > 0 && dim > 0 && ld >= dim && max_iter != 0, "smooth_weights: bad sizes");
    KN_REQUIRE(((uintptr_t)out_w & 15) == 0, "smooth_weights: out_w must be 16-byte aligned");
    if (workspace_bytes < knnsvc_smooth_workspace_bytes(nq))
        return knnsvc_fail(KNNSVC_EWORKSPACE, "smooth_weights: workspace %zu < %zu bytes", workspace_bytes,
                           knnsvc_smooth_workspace_bytes(nq));
    hipStream_t st = (hipStream_t)stream;
    if (nq < 2) {      // no adjacent pair: the reference's loss is NaN and never improves -> softmax(0)
        hipLaunchKernelGGL(fill_quarter_kernel, dim3(1), dim3(64), 0, st, out_w, (long)nq * 4);
        if (out_iters) (void)kn_zero_async(out_iters, sizeof(int), st);
        return knnsvc_check_launch("smooth_weights(fill)");
    }
    float* base = (float*)(((uintptr_t)workspace + 63) & ~(uintptr_t)63);
    float* gram = base;                                 // [36][nq]
    float* state = gram + (size_t)GE * nq;        // theta, m, v, vmax, best, scratch : 6 x [nq] float4
    float* xch = state + (size_t)6 * 4 * nq;            // [2][nq][4] when LDS is too small
    const size_t gl = (size_t)8 * dim * 4;
    KN_REQUIRE(gl <= 150 * 1024, "smooth_weights: feature dim too large for LDS");
    static size_t gattr = 0, aattr = 0;
    if (gl > gattr) {
        if (hipFuncSetAttribute((const void*)gram_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)gl) != hipSuccess)
            return knnsvc_fail(KNNSVC_EHIP, "smooth_weights: hipFuncSetAttribute failed");
        gattr = gl;
    }
    hipLaunchKernelGGL(gram_kernel, dim3((unsigned)(nq - 1)), dim3(256), gl, st, (const long*)idx, (long)nq, pool, (long)np,
                       dim, ld, row_scale, gram);
    int rc = knnsvc_check_launch("gram");
    if (rc) return rc;
    if (nq <= 1536) {            // register-resident loop
        const size_t rl = (size_t)nq * 2 * KW * 4;
        static size_t rattr = 0;
#define KN_ADAM(F)                                                                                                     \
    {                                                                                                                  \
        if (rl > rattr) {                                                                                              \
            if (hipFuncSetAttribute((const void*)adam_reg_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536) != hipSuccess || \
                hipFuncSetAttribute((const void*)adam_reg_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536) != hipSuccess || \
                hipFuncSetAttribute((const void*)adam_reg_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536) != hipSuccess)   \
                return knnsvc_fail(KNNSVC_EHIP, "smooth_weights: hipFuncSetAttribute failed");                         \
            rattr = 65536;                                                                                             \
        }                                                                                                              \
        hipLaunchKernelGGL(adam_reg_kernel<F>, dim3(1), dim3(512), rl, st, (long)nq, dim, scale, max_iter,              \
                           (const float*)gram, out_w, out_iters);                                                      \
    }
        if (nq <= 512) KN_ADAM(1) else if (nq <= 1024) KN_ADAM(2) else KN_ADAM(3)
#undef KN_ADAM
        return knnsvc_check_launch("adam_reg");
    }
    size_t al = (size_t)nq * 2 * KW * 4;
    int use_lds = al <= 144 * 1024;
    if (!use_lds) al = 0;
    if (al > aattr) {
        if (hipFuncSetAttribute((const void*)adam_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)al) != hipSuccess)
            return knnsvc_fail(KNNSVC_EHIP, "smooth_weights: hipFuncSetAttribute failed");
        aattr = al;
    }
    hipLaunchKernelGGL(adam_kernel, dim3(1), dim3(1024), al, st, (long)nq, dim, scale, max_iter, (const float*)gram, state,
                       xch, use_lds, out_w, out_iters);
    return knnsvc_check_launch("adam");
}
