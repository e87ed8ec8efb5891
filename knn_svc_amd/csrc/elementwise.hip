// Small HBM-bound kernels: layer norm (+GELU), gated rel-pos multiplier, weighted neighbour
// gather, reflect padding, complex magnitude, harmonic-amplitude extraction.
#include "common.h"
#include <cstdlib>
#include "gemm2_core.h"      // f16x2_split4: the split ("A2") activation layout of the emulated-fp32 GEMMs

namespace {

// 4 consecutive channels c..c+3 (c % 4 == 0) of a row in the f16x2 split layout: hi plane at (c/32)*128 + (c%32)*2, lo 64 B on
__device__ __forceinline__ void store_split4(float* row, int c, f32x4 y) {
    g2_u32x2 hi, lo;
    f16x2_split4(y, KN_F16X2_A_SCALE, hi, lo);
    char* ob = (char*)row + (c >> 5) * 128 + (c & 31) * 2;
    *(g2_u32x2*)ob = hi;
    *(g2_u32x2*)(ob + 64) = lo;
}
// 4 consecutive channels back to fp32 (exact: hi + lo spans < 24 bits)
__device__ __forceinline__ f32x4 load_split4(const float* row, int c) {
    const char* ib = (const char*)row + (c >> 5) * 128 + (c & 31) * 2;
    const g2_u32x2 hi = *(const g2_u32x2*)ib, lo = *(const g2_u32x2*)(ib + 64);
    f32x4 y;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const unsigned short hb = (unsigned short)(hi[e >> 1] >> (16 * (e & 1))), lb = (unsigned short)(lo[e >> 1] >> (16 * (e & 1)));
        y[e] = ((float)__builtin_bit_cast(_Float16, hb) + (float)__builtin_bit_cast(_Float16, lb)) * (1.0f / KN_F16X2_A_SCALE);
    }
    return y;
}

__device__ __forceinline__ float gelu_erf(float v) {
#pragma clang fp contract(off)
    return kn_gelu(v);
}

// one wave per R rows, rows kept in registers (R * dim <= 2048 floats per wave), two-pass mean / variance.
// The R rows are independent chains (loads, reductions, stores), interleaved for memory-level parallelism.
template <int MAXV, int R>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, long rows, int dim, int ldx,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       int gelu, float* __restrict__ out, int ldo) {
#pragma clang fp contract(off)      // explicit fmaf only: every row slot runs the same roundings, so results do not
                                    // depend on a row's position in the batch (bitwise batch invariance)
    const int lane = threadIdx.x & 63;
    const long row0 = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * R;
    if (row0 >= rows) return;
    f32x4 v[R][MAXV];
    float s[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        s[r] = 0.f;
        const bool ok = row0 + r < rows;
        const float* xr = x + (row0 + r) * (long)ldx;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int c = lane * 4 + 256 * i;
            if (ok && c < dim) { v[r][i] = *(const f32x4*)(xr + c); s[r] += (v[r][i][0] + v[r][i][1]) + (v[r][i][2] + v[r][i][3]); }
            else v[r][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    }
    float mean[R], q[R];
#pragma unroll
    for (int r = 0; r < R; ++r) mean[r] = wave_sum(s[r]) / (float)dim;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        q[r] = 0.f;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int c = lane * 4 + 256 * i;
            if (c < dim) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float d = v[r][i][e] - mean[r]; q[r] = fmaf(d, d, q[r]); }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const float var = wave_sum(q[r]) / (float)dim;
        const float rstd = 1.0f / sqrtf(var + 1e-5f);
        if (row0 + r >= rows) continue;
        float* orow = out + (row0 + r) * (long)ldo;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int c = lane * 4 + 256 * i;
            if (c < dim) {
                const f32x4 g = *(const f32x4*)(gamma + c), b = *(const f32x4*)(beta + c);
                f32x4 y;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float t = fmaf((v[r][i][e] - mean[r]) * rstd, g[e], b[e]);
                    y[e] = (gelu & 1) ? gelu_erf(t) : t;
                }
                if (gelu & 2) store_split4(orow, c, y);      // flags bit 1: f16x2 split layout for a GEMM consumer
                else *(f32x4*)(orow + c) = y;
            }
        }
    }
}

// First layer of the WavLM feature extractor fused end to end (wavlm/WavLM.py:401-419, layer 0):
// Conv1d(1 -> C, k, stride, no bias) -> LayerNorm over channels -> GELU, written once.
// A lane owns CPL = C/64 consecutive channels and keeps their k taps in registers; a block stages the
// input span of its 64 output rows in LDS; each wave walks 16 rows (LN statistics by wave reduction).
// The unfused route wrote, re-read and re-wrote the [T, C] tensor (4.1 GB per 10 min of audio at C = 512).
template <int CPL, int KMAXT, int RP>
__global__ __launch_bounds__(256) void conv0_ln_gelu_kernel(const float* __restrict__ x, long L, long T, int k, int stride,
                                                           const float* __restrict__ w, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float* __restrict__ out, int split, int subs) {
    __shared__ float xs[2][64 * 8 + 32];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int C = CPL * 64;
    const long b = blockIdx.y;
    const float* xb = x + b * L;
    const int span = 63 * stride + k;
    constexpr int NP = (CPL + 1) / 2;      // channel pairs (CPL = 1: the second half of the one pair repeats the channel and is dropped)
    // channels in PAIRS: the taps, the affine and the GELU run on v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 — per channel the
    // same operations in the same order as one at a time (bit-identical), at half the issue slots; the LayerNorm sums stay scalar
    // chains in channel order
    f32x2 wr[NP][KMAXT], g[NP], be[NP];
#pragma unroll
    for (int c = 0; c < NP; ++c) {
        const int ch = lane * CPL + 2 * c, ch1 = CPL > 1 ? ch + 1 : ch;
        g[c] = (f32x2){gamma[ch], gamma[ch1]}; be[c] = (f32x2){beta[ch], beta[ch1]};
#pragma unroll
        for (int t = 0; t < KMAXT; ++t) wr[c][t] = t < k ? (f32x2){w[ch * k + t], w[ch1 * k + t]} : (f32x2){0.f, 0.f};
    }
    // A workgroup walks `subs` consecutive spans of 64 rows (round 5; one span per workgroup before): the 80 filter taps + 16 affine
    // values a lane loads up front were paid once per 16 rows of its wave — now once per 16 * subs.  Two LDS buffers, one barrier
    // per span.  Same arithmetic per row: bit-identical.
  for (int sub = 0; sub < subs; ++sub) {
    const long m0 = ((long)blockIdx.x * subs + sub) * 64;
    if (m0 >= T) break;
    float* xsb = xs[sub & 1];
    for (int i = threadIdx.x; i < span; i += 256) {
        const long p = m0 * stride + i;
        xsb[i] = p < L ? xb[p] : 0.f;
    }
    __syncthreads();
    // RP rows per pass: independent chains (LDS reads, the two reductions, the GELU's rcp / exp2, the stores) interleave, and the
    // LayerNorm sums are DPP folds + readlanes (common.h) instead of twelve dependent ds_bpermute round trips per row — the kernel
    // is VALU work (80 FMAs, 8 GELUs per lane and row) that sat half idle behind those latencies (RP = 1 with shuffles: 2.04 ms
    // for 21 x 96 015 rows, the VALU floor is ~1.1).
    for (int r = 0; r < 16; r += RP) {
        f32x2 y[RP][NP];
        float s[RP], mean[RP], q[RP], rstd[RP];
#pragma unroll
        for (int u = 0; u < RP; ++u) {
            const int lm = wave * 16 + r + u;          // rows past T read zeros / stale LDS: finite, never stored
            float xv[KMAXT];
#pragma unroll
            for (int t = 0; t < KMAXT; ++t) xv[t] = t < k ? xsb[lm * stride + t] : 0.f;
            s[u] = 0.f;
#pragma unroll
            for (int c = 0; c < NP; ++c) {
                f32x2 a = {0.f, 0.f};
#pragma unroll
                for (int t = 0; t < KMAXT; ++t) a = __builtin_elementwise_fma(wr[c][t], (f32x2)(xv[t]), a);
                y[u][c] = a; s[u] += a[0]; if (CPL > 1) s[u] += a[1];
            }
        }
#pragma unroll
        for (int u = 0; u < RP; ++u) mean[u] = (RP > 1 ? wave_sum_dpp(s[u]) : wave_sum(s[u])) / (float)C;
#pragma unroll
        for (int u = 0; u < RP; ++u) {
            q[u] = 0.f;
#pragma unroll
            for (int c = 0; c < CPL; ++c) { const float d = y[u][c >> 1][c & 1] - mean[u]; q[u] += d * d; }
        }
#pragma unroll
        for (int u = 0; u < RP; ++u) rstd[u] = 1.0f / sqrtf((RP > 1 ? wave_sum_dpp(q[u]) : wave_sum(q[u])) / (float)C + 1e-5f);
#pragma unroll
        for (int u = 0; u < RP; ++u) {
            const long m = m0 + wave * 16 + r + u;
            if (m >= T) break;
            float* o = out + (b * T + m) * C + lane * CPL;
#pragma unroll
            for (int c = 0; c < NP; ++c) y[u][c] = kn_gelu2(__builtin_elementwise_fma((y[u][c] - mean[u]) * rstd[u], g[c], be[c]));
            if (CPL % 4 == 0) {
                if (split) {
                    float* orow = out + (b * T + m) * C;
#pragma unroll
                    for (int c = 0; c < CPL; c += 4) store_split4(orow, lane * CPL + c, (f32x4){y[u][c / 2][0], y[u][c / 2][1], y[u][c / 2 + 1][0], y[u][c / 2 + 1][1]});
                } else {
#pragma unroll
                    for (int c = 0; c < CPL; c += 4) *(f32x4*)(o + c) = (f32x4){y[u][c / 2][0], y[u][c / 2][1], y[u][c / 2 + 1][0], y[u][c / 2 + 1][1]};
                }
            } else {
#pragma unroll
                for (int c = 0; c < CPL; ++c) o[c] = y[u][c >> 1][c & 1];
            }
        }
    }
  }
}

// one wave per row; lane l covers channels 16l..16l+15 (a quarter of a 64-wide head)
__global__ __launch_bounds__(256) void gate_kernel(const float* __restrict__ xn, long rows, int heads, int ldx,
                                                  const float* __restrict__ w2, const float* __restrict__ b2,
                                                  const float* __restrict__ grep_a, float* __restrict__ gate, int split) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int h = lane >> 2, part = lane & 3;
    float sa = 0.f, sb = 0.f;
    if (h < heads) {
        const float* xr = xn + row * (long)ldx + lane * 16;
#pragma unroll
        for (int e = 0; e < 16; e += 4) {
            const f32x4 xv = split ? load_split4(xn + row * (long)ldx, lane * 16 + e) : *(const f32x4*)(xr + e);
            const f32x4 wa = *(const f32x4*)(w2 + part * 16 + e), wb = *(const f32x4*)(w2 + 64 + part * 16 + e);
#pragma unroll
            for (int t = 0; t < 4; ++t) { sa += xv[t] * wa[t]; sb += xv[t] * wb[t]; }
        }
    }
    sa += __shfl_xor(sa, 1, 64); sa += __shfl_xor(sa, 2, 64);
    sb += __shfl_xor(sb, 1, 64); sb += __shfl_xor(sb, 2, 64);
    if (h < heads && part == 0) {
        const float ga = 1.0f / (1.0f + expf(-(sa + b2[0])));
        const float gb = 1.0f / (1.0f + expf(-(sb + b2[1])));
        gate[row * heads + h] = ga * (gb * grep_a[h] - 1.0f) + 2.0f;
    }
}

__global__ __launch_bounds__(256) void weighted_gather_kernel(const long* __restrict__ idx, const float* __restrict__ w,
                                                             long nq, int k, const float* __restrict__ pool, int dim,
                                                             int ld, float* __restrict__ out) {
#pragma clang fp contract(off)
    const long row = blockIdx.x;
    for (int c = threadIdx.x; c < dim; c += 256) {
        float acc = 0.f;
        for (int j = 0; j < k; ++j) {
            const float wj = w ? w[row * k + j] : 1.0f / (float)k;
            const float t = pool[idx[row * k + j] * (long)ld + c] * wj;
            acc = j == 0 ? t : acc + t;
        }
        out[row * (long)dim + c] = acc;
    }
}

__global__ void reflect_pad_kernel(const float* __restrict__ x, long n, int pad, float* __restrict__ out) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n + 2 * (long)pad) return;
    long j = i - pad;
    if (j < 0) j = -j;
    if (j >= n) j = 2 * (n - 1) - j;
    out[i] = x[j];
}

__global__ void complex_mag_kernel(const float* __restrict__ reim, long rows, int bins, int ld, float* __restrict__ out) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * bins) return;
    const long r = i / bins; const int b = (int)(i - r * bins);
    out[i] = hypotf(reim[r * ld + b], reim[r * ld + bins + b]);
}

// ddsp_prematch_dataset.py:391-404
__global__ void harmonic_amps_kernel(const float* __restrict__ spec, const float* __restrict__ f0, long T, int bins,
                                     int n_harm, float* __restrict__ harm) {
#pragma clang fp contract(off)
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= T * n_harm) return;
    const long t = i / n_harm; const int k = (int)(i - t * n_harm) + 1;
    const float f = f0[t];
    const float* s = spec + t * bins;
    float v;
    if (f == 0.f) {
        if (k == 1) { v = s[0]; for (int b = 1; b < bins; ++b) v = fmaxf(v, s[b]); }
        else v = 0.f;
    } else {
        const int nb = bins * 8;
        float pos = ((f * (float)k) * 2.0f) * (float)nb / 16000.0f;       // same operation order as the reference
        pos = fminf(pos, (float)nb);
        const int gi = (int)rintf(pos);                                    // torch.round: half to even
        if (gi >= nb) v = 0.f;                                             // the zero pad bin
        else {
            // F.interpolate(scale_factor=8, mode='linear', align_corners=False): src = (dst + .5)/8 - .5, clamped at 0
            float src = ((float)gi + 0.5f) * 0.125f - 0.5f;
            if (src < 0.f) src = 0.f;
            const int i0 = (int)src;
            const int i1 = i0 + (i0 < bins - 1 ? 1 : 0);
            const float l1 = src - (float)i0, l0 = 1.0f - l1;
            v = l0 * s[i0] + l1 * s[i1];
        }
    }
    harm[i] = 0.0108f * v;
}

// Batched centre padding for the framed-signal STFT GEMM: item b = x[offs[b] .. offs[b+1]) becomes row b of out
// ([batches, stride]): reflect-padded by `pad` on both sides, zeros from n_b + 2 pad on (the GEMM's K padding and the frames of
// shorter items).  One launch per pool instead of one per file.
__global__ __launch_bounds__(256) void reflect_pad_batch_kernel(const float* __restrict__ x, const long* __restrict__ offs, int pad,
                                                                float* __restrict__ out, long stride) {
    const int b = blockIdx.y;
    const long o0 = offs[b], n = offs[b + 1] - o0;
    float* ob = out + (long)b * stride;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < stride; i += (long)gridDim.x * blockDim.x) {
        float v = 0.f;
        if (i < n + 2 * (long)pad) {
            long j = i - pad;
            if (j < 0) j = -j;
            if (j >= n) j = 2 * (n - 1) - j;
            v = x[o0 + j];
        }
        ob[i] = v;
    }
}

// complex_mag + harmonic_amps in one pass, one wave per frame: the [2 bins] DFT row is read once, its magnitudes go to the
// spectrum output AND to LDS, the 49 harmonic taps are gathered from LDS (same arithmetic, operation for operation, as
// complex_mag_kernel / harmonic_amps_kernel; the unvoiced rule's maximum is a wave reduction instead of one thread's loop
// over 200 bins — max is exact in any order).
__global__ __launch_bounds__(256) void spec_harm_kernel(const float* __restrict__ reim, long rows, int bins, int ld,
                                                        const float* __restrict__ f0, int n_harm, float* __restrict__ spec,
                                                        float* __restrict__ harm) {
#pragma clang fp contract(off)
    extern __shared__ float sh[];                                   // 4 waves x bins
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long t = (long)blockIdx.x * 4 + wave;
    if (t >= rows) return;
    float* s = sh + wave * bins;
    const float* r = reim + t * (long)ld;
    float mx = 0.f;                                                 // magnitudes are >= 0
    for (int b = lane; b < bins; b += 64) {
        const float v = hypotf(r[b], r[bins + b]);
        s[b] = v; spec[t * bins + b] = v;
        mx = fmaxf(mx, v);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    // (the LDS row is private to this wave; LDS operations of one wave complete in order)
    const float f = f0[t];
    for (int k = lane + 1; k <= n_harm; k += 64) {
        float v;
        if (f == 0.f) v = k == 1 ? mx : 0.f;
        else {
            const int nb = bins * 8;
            float pos = ((f * (float)k) * 2.0f) * (float)nb / 16000.0f;
            pos = fminf(pos, (float)nb);
            const int gi = (int)rintf(pos);
            if (gi >= nb) v = 0.f;
            else {
                float src = ((float)gi + 0.5f) * 0.125f - 0.5f;
                if (src < 0.f) src = 0.f;
                const int i0 = (int)src;
                const int i1 = i0 + (i0 < bins - 1 ? 1 : 0);
                const float l1 = src - (float)i0, l0 = 1.0f - l1;
                v = l0 * s[i0] + l1 * s[i1];
            }
        }
        harm[t * n_harm + (k - 1)] = 0.0108f * v;
    }
}

// rows t >= lens[b] of a [batches, T, dim] activation become zero (WavLM's x[padding_mask] = 0, wavlm/WavLM.py:353, 574-575)
__global__ __launch_bounds__(256) void mask_rows_kernel(float* __restrict__ x, int T, int dim, int ld, const int* __restrict__ lens) {
    const int b = blockIdx.y;
    const int len = lens[b];
    const long n4 = (long)(T - len) * (dim / 4);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const long r = len + i / (dim / 4); const int c = (int)(i % (dim / 4)) * 4;
        *(f32x4*)(x + ((long)b * T + r) * ld + c) = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
}

}  // namespace

extern "C" int knnsvc_mask_rows(float* x, int32_t batches, int32_t T, int32_t dim, int32_t ld, const int32_t* lens, void* stream) {
    KN_REQUIRE(x && lens && batches > 0 && batches <= 65535 && T > 0 && dim > 0 && dim % 4 == 0 && ld >= dim && ld % 4 == 0 &&
               ((uintptr_t)x & 15) == 0, "mask_rows: bad arguments");
    hipLaunchKernelGGL(mask_rows_kernel, dim3(64, (unsigned)batches), dim3(256), 0, (hipStream_t)stream, x, T, dim, ld, lens);
    return knnsvc_check_launch("mask_rows");
}

extern "C" int knnsvc_layernorm(const float* x, int64_t rows, int32_t dim, int32_t ldx, const float* gamma,
                                const float* beta, int32_t gelu, float* out, int32_t ldo, void* stream) {
    KN_REQUIRE(x && gamma && beta && out, "layernorm: null pointer");
    KN_REQUIRE(dim > 0 && dim % 4 == 0 && dim <= 2048, "layernorm: dim must be a multiple of 4 and <= 2048");
    KN_REQUIRE(ldx % 4 == 0 && ldo % 4 == 0 && ldx >= dim && ldo >= dim, "layernorm: bad row strides");
    KN_REQUIRE(!(gelu & 2) || (dim % 32 == 0 && ldo % 32 == 0), "layernorm: split output needs dim % 32 == 0 and ldo % 32 == 0");
    KN_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)out & 15) == 0 && ((uintptr_t)gamma & 15) == 0 &&
               ((uintptr_t)beta & 15) == 0, "layernorm: pointers must be 16-byte aligned");
    if (rows <= 0) return KNNSVC_OK;
    hipStream_t st = (hipStream_t)stream;
    if (dim <= 512)
        hipLaunchKernelGGL((layernorm_kernel<2, 4>), dim3((unsigned)cdiv64(rows, 16)), dim3(256), 0, st, x, (long)rows, dim, ldx,
                           gamma, beta, gelu, out, ldo);
    else if (dim <= 1024)
        hipLaunchKernelGGL((layernorm_kernel<4, 2>), dim3((unsigned)cdiv64(rows, 8)), dim3(256), 0, st, x, (long)rows, dim, ldx,
                           gamma, beta, gelu, out, ldo);
    else
        hipLaunchKernelGGL((layernorm_kernel<8, 1>), dim3((unsigned)cdiv64(rows, 4)), dim3(256), 0, st, x, (long)rows, dim, ldx,
                           gamma, beta, gelu, out, ldo);
    return knnsvc_check_launch("layernorm");
}

extern "C" int knnsvc_wavlm_gate(const float* xn, int64_t rows, int32_t heads, int32_t head_dim, int32_t ldx,
                                 const float* w2, const float* b2, const float* grep_a, float* gate, int32_t x_f16x2, void* stream) {
    KN_REQUIRE(xn && w2 && b2 && grep_a && gate, "wavlm_gate: null pointer");
    KN_REQUIRE(head_dim == 64 && heads >= 1 && heads <= 16, "wavlm_gate: head_dim must be 64, heads <= 16");
    KN_REQUIRE(ldx % 4 == 0 && ldx >= heads * 64 && ((uintptr_t)xn & 15) == 0 && ((uintptr_t)w2 & 15) == 0,
               "wavlm_gate: alignment");
    if (rows <= 0) return KNNSVC_OK;
    hipLaunchKernelGGL(gate_kernel, dim3((unsigned)cdiv64(rows, 4)), dim3(256), 0, (hipStream_t)stream,
                       xn, (long)rows, heads, ldx, w2, b2, grep_a, gate, x_f16x2);
    return knnsvc_check_launch("wavlm_gate");
}

extern "C" int knnsvc_weighted_gather(const int64_t* idx, const float* w, int64_t nq, int32_t k, const float* pool,
                                      int32_t dim, int32_t ld, int32_t mean_mode, float* out, void* stream) {
    (void)mean_mode;   // mean of k == sum of x * (1/k) bit for bit when k is a power of two (k = 4 on the path)
    KN_REQUIRE(idx && pool && out && k > 0 && dim > 0 && ld >= dim, "weighted_gather: bad arguments");
    KN_REQUIRE(w || (k & (k - 1)) == 0, "weighted_gather: uniform weights need a power-of-two k");
    if (nq <= 0) return KNNSVC_OK;
    hipLaunchKernelGGL(weighted_gather_kernel, dim3((unsigned)nq), dim3(256), 0, (hipStream_t)stream,
                       (const long*)idx, w, (long)nq, k, pool, dim, ld, out);
    return knnsvc_check_launch("weighted_gather");
}

extern "C" int knnsvc_reflect_pad(const float* x, int64_t n, int32_t pad, float* out, void* stream) {
    KN_REQUIRE(x && out && n > pad && pad >= 0, "reflect_pad: needs n > pad >= 0");
    const long tot = n + 2 * (long)pad;
    hipLaunchKernelGGL(reflect_pad_kernel, dim3((unsigned)cdiv64(tot, 256)), dim3(256), 0, (hipStream_t)stream,
                       x, (long)n, pad, out);
    return knnsvc_check_launch("reflect_pad");
}

extern "C" int knnsvc_reflect_pad_batch(const float* x, const int64_t* offs, int32_t batches, int32_t pad, float* out,
                                        int64_t stride, void* stream) {
    KN_REQUIRE(x && offs && out && batches > 0 && batches <= 65535 && pad >= 0 && stride > 2 * (int64_t)pad,
               "reflect_pad_batch: bad arguments (every item needs n > pad; stride > 2 pad)");
    const unsigned gx = (unsigned)(cdiv64(stride, 256) < 2048 ? cdiv64(stride, 256) : 2048);
    hipLaunchKernelGGL(reflect_pad_batch_kernel, dim3(gx, (unsigned)batches), dim3(256), 0, (hipStream_t)stream,
                       x, (const long*)offs, pad, out, (long)stride);
    return knnsvc_check_launch("reflect_pad_batch");
}

extern "C" int knnsvc_spec_harm(const float* reim, int64_t rows, int32_t bins, int32_t ld, const float* f0, int32_t n_harm,
                                float* spec, float* harm, void* stream) {
    KN_REQUIRE(reim && f0 && spec && harm && bins > 0 && bins <= 4096 && ld >= 2 * bins && n_harm > 0, "spec_harm: bad arguments");
    if (rows <= 0) return KNNSVC_OK;
    hipLaunchKernelGGL(spec_harm_kernel, dim3((unsigned)cdiv64(rows, 4)), dim3(256), (size_t)4 * bins * sizeof(float),
                       (hipStream_t)stream, reim, (long)rows, bins, ld, f0, n_harm, spec, harm);
    return knnsvc_check_launch("spec_harm");
}

extern "C" int knnsvc_complex_mag(const float* reim, int64_t rows, int32_t bins, int32_t ld, float* out, void* stream) {
    KN_REQUIRE(reim && out && bins > 0 && ld >= 2 * bins, "complex_mag: bad arguments");
    if (rows <= 0) return KNNSVC_OK;
    hipLaunchKernelGGL(complex_mag_kernel, dim3((unsigned)cdiv64(rows * bins, 256)), dim3(256), 0,
                       (hipStream_t)stream, reim, (long)rows, bins, ld, out);
    return knnsvc_check_launch("complex_mag");
}

extern "C" int knnsvc_harmonic_amps(const float* spec, const float* f0, int64_t T, int32_t bins, int32_t n_harm,
                                    float* harm, void* stream) {
    KN_REQUIRE(spec && f0 && harm && bins > 0 && n_harm > 0, "harmonic_amps: bad arguments");
    if (T <= 0) return KNNSVC_OK;
    hipLaunchKernelGGL(harmonic_amps_kernel, dim3((unsigned)cdiv64(T * n_harm, 256)), dim3(256), 0,
                       (hipStream_t)stream, spec, f0, (long)T, bins, n_harm, harm);
    return knnsvc_check_launch("harmonic_amps");
}

extern "C" int knnsvc_wavlm_conv0(const float* x, int32_t batches, int64_t L, const float* w, int32_t channels, int32_t k,
                                  int32_t stride, const float* gamma, const float* beta, float* out, int32_t out_f16x2, void* stream) {
    KN_REQUIRE(x && w && gamma && beta && out, "wavlm_conv0: null pointer");
    KN_REQUIRE(batches > 0 && batches <= 65535 && L >= k && k >= 1 && k <= 16 && stride >= 1 && stride <= 8,
               "wavlm_conv0: needs 1 <= k <= 16, 1 <= stride <= 8");
    KN_REQUIRE(channels == 64 || channels == 128 || channels == 256 || channels == 512, "wavlm_conv0: channels must be 64/128/256/512");
    KN_REQUIRE(!out_f16x2 || channels >= 256, "wavlm_conv0: split output needs >= 256 channels");
    const long T = (L - k) / stride + 1;
    hipStream_t st = (hipStream_t)stream;
    static int rp = -1, subs_env = -1;        // rows per pass and reduction flavour: 2 (default), 4, or 1 = round 2's kernel (shuffle sums), for A/B
    if (rp < 0) { const char* e = getenv("KNNSVC_CONV0_RP"); rp = e ? atoi(e) : 2; if (rp != 1 && rp != 4) rp = 2; }
    if (subs_env < 0) { const char* e = getenv("KNNSVC_CONV0_SUBS"); subs_env = e ? atoi(e) : 8; if (subs_env < 1 || subs_env > 64) subs_env = 8; }
    // spans of 64 rows per workgroup: 8 (21 x 96 063 rows: 1.54 -> 1.34 ms; 4: 1.36, 2: 1.41) where that still leaves several rounds of workgroups (768 are resident), 1 for short inputs
    const int subs = cdiv64(T, 64L * subs_env) * batches >= 3072 ? subs_env : 1;
    dim3 grid((unsigned)cdiv64(T, 64L * subs), (unsigned)batches);
#define KN_C0R(CPL, KM, RP) hipLaunchKernelGGL((conv0_ln_gelu_kernel<CPL, KM, RP>), grid, dim3(256), 0, st, x, (long)L, T, k, stride, w, gamma, beta, out, out_f16x2, subs)
#define KN_C0(CPL, KM) { if (rp == 2) KN_C0R(CPL, KM, 2); else if (rp == 4) KN_C0R(CPL, KM, 4); else KN_C0R(CPL, KM, 1); }
    if (k <= 10) {        // WavLM's k = 10: no padded taps in the unrolled FIR
        if (channels == 512) KN_C0(8, 10) else if (channels == 256) KN_C0(4, 10) else if (channels == 128) KN_C0(2, 10) else KN_C0(1, 10)
    } else {
        if (channels == 512) KN_C0(8, 16) else if (channels == 256) KN_C0(4, 16) else if (channels == 128) KN_C0(2, 16) else KN_C0(1, 16)
    }
#undef KN_C0R
#undef KN_C0
    return knnsvc_check_launch("wavlm_conv0");
}

// ---------------------------------------------------------------------------------------------------
// Prematch helpers (per_spk_extract, ddsp_prematch_dataset.py:1464-1772)
// ---------------------------------------------------------------------------------------------------
namespace {

// `.half().float()`: round-to-nearest-even to fp16 (overflow -> inf, as torch), widened back
__global__ __launch_bounds__(256) void round_f16_kernel(const float* __restrict__ x, long n, float* __restrict__ out) {
    const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i + 3 < n) {
        const f32x4 v = *(const f32x4*)(x + i);
        f32x4 r;
#pragma unroll
        for (int e = 0; e < 4; ++e) r[e] = (float)(_Float16)v[e];
        *(f32x4*)(out + i) = r;
    } else {
        for (long j = i; j < n; ++j) out[j] = (float)(_Float16)x[j];
    }
}

// one wave per query frame: L1 norm of its own spectrum row and of the k gathered pool rows (sums in f64, rounded once)
__global__ __launch_bounds__(256) void amp_ratio_kernel(const float* __restrict__ spec_q, int ld_q,
                                                       const float* __restrict__ spec_pool, int ld_pool, long np,
                                                       const long* __restrict__ idx, long nq, int k, int bins,
                                                       float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const long t = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= nq) return;
    double s = 0.0;
    for (int c = lane; c < bins; c += 64) s += (double)fabsf(spec_q[t * ld_q + c]);
    const float orig = (float)wave_sum_d(s);
    for (int j = 0; j < k; ++j) {
        long id = idx[t * k + j];
        id = id < 0 ? 0 : (id > np - 1 ? np - 1 : id);
        double a = 0.0;
        for (int c = lane; c < bins; c += 64) a += (double)fabsf(spec_pool[id * ld_pool + c]);
        const float l1 = (float)wave_sum_d(a);
        if (lane == 0) out[t * k + j] = orig / (l1 + 1e-5f);
    }
}

}  // namespace

extern "C" int knnsvc_round_f16(const float* x, int64_t n, float* out, void* stream) {
    KN_REQUIRE(x && out && n >= 0, "round_f16: bad arguments");
    KN_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)out & 15) == 0, "round_f16: pointers must be 16-byte aligned");
    if (n == 0) return KNNSVC_OK;
    hipLaunchKernelGGL(round_f16_kernel, dim3((unsigned)cdiv64(n, 1024)), dim3(256), 0, (hipStream_t)stream, x, (long)n, out);
    return knnsvc_check_launch("round_f16");
}

extern "C" int knnsvc_amp_ratio(const float* spec_q, int32_t ld_q, const float* spec_pool, int32_t ld_pool, int64_t np,
                                const int64_t* idx, int64_t nq, int32_t k, int32_t bins, float* out, void* stream) {
    KN_REQUIRE(spec_q && spec_pool && idx && out, "amp_ratio: null pointer");
    KN_REQUIRE(np > 0 && k > 0 && bins > 0 && ld_q >= bins && ld_pool >= bins, "amp_ratio: bad sizes");
    if (nq <= 0) return KNNSVC_OK;
    hipLaunchKernelGGL(amp_ratio_kernel, dim3((unsigned)cdiv64(nq, 4)), dim3(256), 0, (hipStream_t)stream, spec_q, ld_q,
                       spec_pool, ld_pool, (long)np, (const long*)idx, (long)nq, k, bins, out);
    return knnsvc_check_launch("amp_ratio");
}
