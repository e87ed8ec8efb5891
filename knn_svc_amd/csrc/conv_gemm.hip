// Implicit-GEMM convolution / linear / transposed convolution on fp32 MFMA.
// See include/knnsvc_hip.h (knnsvc_conv_gemm) for the operator definition and the
// reference call sites it replaces.
#include "gemm_core.h"
#include "gemm3_core.h"
#include "gemm2_core.h"

namespace {

thread_local const char* g_last_epilogue = "";    // ... and the epilogue it took ("patch" / "lane" / ""): knnsvc_conv_gemm_last_epilogue
thread_local const char* g_last_kernel = "";      // which kernel the last knnsvc_conv_gemm of this thread launched

struct ConvArgs {
    const float* x; long x_bstride, x_gstride; int ldx, t_in, cin, taps, stride, dil, pad; float a_slope;
    const float* w; long w_gstride; int n;
    const float* bias; long bias_gstride; int bias_period;
    float* out; long o_bstride, o_gstride; int ldo, m;
    int act; float act_slope;
    const float* resid; long r_bstride, r_gstride; int ldr;
    int accumulate; float div;
    int groups;
    int convt_u, convt_cout, convt_pad, t_out;
    int K;
    const unsigned short* w3;   // weights pre-split into 3 bf16 planes ([n][K/32][3][32]) or null
    const unsigned short* w2;   // weights pre-split into 2 fp16 planes ([n][K/32][2][32], scaled) or null
    int x_split;                // A operand already in the f16x2 split layout (same pitch in bytes)
    int out_split;              // epilogue writes the f16x2 split layout (scale 16) instead of fp32 ...
    int split_from;             // ... for output columns >= split_from (a multiple of 32; 0 = every column)
    float out_scale;            // accumulator -> output factor (1 except on the f16x2 path)
    float a_scale;              // f16x2 path: power-of-two factor applied to activations before the split
    int lin;                    // plain output below 2 GiB: the buffer-addressed epilogue applies (conv_epilogue_lin)
    int wide;                   // ... and bias (+ residual) (+ range slot) is all there is, 16-byte rows: conv_epilogue_wide32 (windowed kernels)
    int plain;                  // set by the launcher: one column tile and a row-tile count that is no multiple of 8 -> blockIdx.x IS the row tile
    // dynamic range of the f16x2 path (include/knnsvc_hip.h, "Range"): device slots holding an upper bound of |x| / |w|;
    // when set, the kernel derives the power-of-two operand scale from the slot (kn_pick_scale) instead of a_scale / w_scale
    const float* x_absmax; const float* w_absmax;
    float x_bound_mul, x_bound_add;   // bound of |x| = x_bound_mul * max(x_absmax slot) + x_bound_add (1, 0 = the slot itself)
    float* out_absmax;          // optional: atomicMax of |out| over everything this launch stores (a later launch's x_absmax)
    float w_scale;              // f16x2 path: the scale the weights were split with
    float split_scale;          // scale of the split layout written by the epilogue (out_split)
    // dynamic sequence length (include/knnsvc_hip.h, "Dynamic length"): t_in / m / t_out are affine in a device-side count
    const int* n_dyn; int dyn_tin_mul, dyn_tin_add, dyn_m_mul, dyn_m_add, dyn_tout_mul;
};

// largest power of two s with absmax * s < 2^15 (fp16 tops out at 65504): absmax in [2^E, 2^(E+1)) -> s = 2^(14-E).
// 0 / tiny bounds cap the scale at 2^54; inf / NaN give a tiny scale and stay inf / NaN in the product (loud, not wrong).
__device__ __host__ __forceinline__ float kn_pick_scale(float absmax) {
    unsigned u;
    __builtin_memcpy(&u, &absmax, 4);
    unsigned e = (u >> 23) & 0xFFu;
    e = e < 87u ? 87u : e;
    u = (268u - e) << 23;
    float s;
    __builtin_memcpy(&s, &u, 4);
    return s;
}

// A range slot is KN_STRIPES = 64 floats, one per 128-byte cache line (8 KiB): a producer BLOCK folds its maximum into stripe
// (block id % 64) with one fire-and-forget atomicMax; consumers take the maximum of the 64 (scalar loads, cached per CU).
// Why this shape (generator at 1500 frames, 7.0 ms per forward without range slots): atomics that meet on one cache line
// serialise at ~25 ns each, and the blocks of a launch's first round all finish at about the same time — every one of them
// still sees an empty slot.  One float per slot with a load-and-compare first: +15..20 us per launch (+1.5 ms per forward);
// unconditional per-wave atomics on 16 or 128 adjacent floats: +2.5 ms; a pre-check value fetched at kernel start: +2.7 ms.
// With one atomic per block and 64 separate lines a launch of 3750 blocks queues ~60 atomics per line (~1.5 us), nothing
// loads, nothing waits, and the slot's content is a pure function of the data (max is order-independent).
constexpr int KN_STRIPES = 64, KN_STRIPE_FLOATS = 32;
__device__ __forceinline__ float* kn_stripe(float* slot) { return slot + (blockIdx.x & (KN_STRIPES - 1)) * KN_STRIPE_FLOATS; }
__device__ __forceinline__ float kn_slot_max(const float* slot) {
    unsigned m = 0;
#pragma unroll
    for (int i = 0; i < KN_STRIPES; ++i) { const unsigned v = __float_as_uint(slot[i * KN_STRIPE_FLOATS]) & 0x7FFFFFFFu; m = v > m ? v : m; }
    return __uint_as_float(m);
}

// the kernel parameter struct is a by-value copy: patch the run-time quantities in place before anything reads them
// (uniform scalar loads): the valid sequence length of a launch captured for a whole length bucket, then the operand scales
__device__ __forceinline__ void resolve_dyn(ConvArgs& a) {
    if (a.n_dyn) {
        const int n = *a.n_dyn;
        a.t_in = n * a.dyn_tin_mul + a.dyn_tin_add;
        a.m = n * a.dyn_m_mul + a.dyn_m_add;
        if (a.convt_u) a.t_out = n * a.dyn_tout_mul;
    }
}
__device__ __forceinline__ void resolve_scales(ConvArgs& a) {
    resolve_dyn(a);
    if (a.x_absmax) a.a_scale = kn_pick_scale(fmaf(kn_slot_max(a.x_absmax), a.x_bound_mul, a.x_bound_add));
    if (a.w_absmax) a.w_scale = kn_pick_scale(kn_slot_max(a.w_absmax));
    if (a.x_absmax || a.w_absmax) a.out_scale = 1.0f / (a.a_scale * a.w_scale);
    if (a.out_absmax) a.out_absmax = kn_stripe(a.out_absmax);
}

// |v| as ordered bits: NaN sorts above inf, so a NaN anywhere in the output reaches the slot (fmaxf would drop it)
__device__ __forceinline__ unsigned abs_bits(float v) { return __float_as_uint(v) & 0x7FFFFFFFu; }
// block-level fold: wave maxima meet in LDS, thread 0 sends the block's ONE atomic (zero is never sent: the slot starts there)
__device__ __forceinline__ void publish_absmax(float* stripe, unsigned m) {
    __shared__ unsigned s_wave_max[16];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const unsigned t = (unsigned)__shfl_xor((int)m, o, 64); m = t > m ? t : m; }
    if ((threadIdx.x & 63) == 0) s_wave_max[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned bm = 0;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) bm = s_wave_max[w] > bm ? s_wave_max[w] : bm;
        if (bm) atomicMax((unsigned*)stripe, bm);
    }
}

__device__ __forceinline__ float lrelu(float v, float s) { return v > 0.f ? v : v * s; }

// ---- A operand: im2col view of the channel-last activation matrix ------------------------
template <int VEC, int NF4>
struct ALoader {
    const float* x; int ldx, t_in, cin, stride, dil, pad, K; float slope;
    int rowbase[NF4];          // m*stride - pad, or INT_MIN/2 when m is out of range
    int k_cur, tap, c;         // state of the VEC==4 path (k = slab*32 + (tid&7)*4)
    __device__ __forceinline__ ALoader(const ConvArgs& a, const float* xz, int m0, int tid)
        : x(xz), ldx(a.ldx), t_in(a.t_in), cin(a.cin), stride(a.stride), dil(a.dil), pad(a.pad), K(a.K),
          slope(a.a_slope) {
#pragma unroll
        for (int j = 0; j < NF4; ++j) {
            int m = m0 + (tid >> 3) + 32 * j;
            rowbase[j] = (m < a.m) ? m * stride - pad : -(1 << 30);
        }
        k_cur = (tid & 7) * 4;
        tap = k_cur / cin;
        c = k_cur - tap * cin;
    }
    __device__ __forceinline__ f32x4 operator()(int kt, int j, int) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        const int k = kt * 32 + (threadIdx.x & 7) * 4;
        if (VEC == 4) {
            if (k != k_cur) {                     // advance (tap, c) by whole slabs
                c += k - k_cur;
                k_cur = k;
                while (c >= cin) { c -= cin; ++tap; }
            }
            const int row = rowbase[j] + tap * dil;
            if (k < K && row >= 0 && row < t_in) v = *(const f32x4*)(x + (long)row * ldx + c);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int ke = k + e;
                const int tp = ke / cin, cc = ke - tp * cin;
                const int row = rowbase[j] + tp * dil;
                if (ke < K && row >= 0 && row < t_in) v[e] = x[(long)row * ldx + cc];
            }
        }
        return v;
    }
    __device__ __forceinline__ void begin(int) const {}
    // prologue activation, applied when the slab is written to LDS (keeps the global load in flight)
    __device__ __forceinline__ f32x4 finish(f32x4 v) const {
        if (slope != 1.0f) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = lrelu(v[e], slope);
        }
        return v;
    }
};

// ---- B operand: weights [n][K], K contiguous ----------------------------------------------
template <int VEC, int NF4>
struct BLoader {
    const float* w; int N, K, n0;
    __device__ __forceinline__ BLoader(const float* wz, int N_, int K_, int n0_) : w(wz), N(N_), K(K_), n0(n0_) {}
    __device__ __forceinline__ f32x4 operator()(int kt, int j, int) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        const int n = n0 + (threadIdx.x >> 3) + 32 * j;
        const int k = kt * 32 + (threadIdx.x & 7) * 4;
        if (n < N) {
            if (VEC == 4) {
                if (k < K) v = *(const f32x4*)(w + (long)n * K + k);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (k + e < K) v[e] = w[(long)n * K + k + e];
            }
        }
        return v;
    }
    __device__ __forceinline__ void begin(int) const {}
};

// ---- fast path (cin % 32 == 0): buffer loads with hardware range checking ---------------------
// Every slab of 32 k's lies inside ONE tap, so (tap, channel offset) is wave-uniform scalar state and
// a thread's address is  constant_per_thread + uniform_per_slab.  Rows outside [0, t_in) (conv padding,
// m >= M) and weight rows n >= N fall outside the buffer resource and read as 0 — no branches, no
// 64-bit address arithmetic in the K loop.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int OOB = 0x40000000;

// descriptor built from readfirstlane'd inputs: provably wave-uniform, so hipcc keeps it in SGPRs and
// does not wrap every buffer_load in a waterfall loop (cdna_hip_programming.md T20)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t uniform_rsrc(const void* p, int bytes) {
    const unsigned long long u = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
    bytes = __builtin_amdgcn_readfirstlane(bytes);
    return __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), 0, bytes, 0x00020000);
}      // pushes an offset out of every resource used here (< 1 GiB each)

template <int NF4, int RS = 32>
struct FastALoader {
    int off[NF4]; int cin, step_tap; float slope;
    int c0, uoff;                     // uniform: channel offset inside the tap, byte offset of (tap, c0)
    __device__ __forceinline__ static __amdgpu_buffer_rsrc_t desc(const ConvArgs& a, const float* xz) {
        return uniform_rsrc(xz, (int)(((long)(a.t_in - 1) * a.ldx + a.cin) * 4));
    }
    __device__ __forceinline__ FastALoader(const ConvArgs& a, int m0, int tid) : cin(a.cin), slope(a.a_slope) {
#pragma unroll
        for (int j = 0; j < NF4; ++j) {
            const int m = m0 + (tid >> 3) + RS * j;
            off[j] = (m < a.m) ? ((m * a.stride - a.pad) * a.ldx + (tid & 7) * 4) * 4 : OOB;
        }
        step_tap = (a.dil * a.ldx - a.cin) * 4;       // byte step from the end of one tap to the start of the next
        c0 = 0; uoff = 0;
    }
    __device__ __forceinline__ void begin(int kt) {
        if (kt == 0) return;
        c0 += 32; uoff += 128;
        if (c0 == cin) { c0 = 0; uoff += step_tap; }
    }
    __device__ __forceinline__ f32x4 operator()(int, int j, __amdgpu_buffer_rsrc_t rsrc) const {
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, off[j] + uoff, 0, 0));
    }
    __device__ __forceinline__ f32x4 finish(f32x4 v) const {
        if (slope != 1.0f) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = lrelu(v[e], slope);
        }
        return v;
    }
};

template <int NF4>
struct FastBLoader {
    int off[NF4]; int uoff;
    __device__ __forceinline__ static __amdgpu_buffer_rsrc_t desc(const float* wz, int N, int K) { return uniform_rsrc(wz, (int)((long)N * K * 4)); }
    __device__ __forceinline__ FastBLoader(int N, int K, int n0, int tid) {
#pragma unroll
        for (int j = 0; j < NF4; ++j) {
            const int n = n0 + (tid >> 3) + 32 * j;
            off[j] = (n < N) ? (n * K + (tid & 7) * 4) * 4 : OOB;
        }
        uoff = 0;
    }
    __device__ __forceinline__ void begin(int kt) { uoff = kt * 128; }
    __device__ __forceinline__ f32x4 operator()(int, int j, __amdgpu_buffer_rsrc_t rsrc) const {
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, off[j] + uoff, 0, 0));
    }
};

// split weights [n][K/32][3][32] bf16: thread piece q = tid + 256 j -> row q / 12, 16-byte piece q % 12
template <int NP, int NPIECES>
struct Split3BLoader {
    int off[NP]; int uoff;
    __device__ __forceinline__ static __amdgpu_buffer_rsrc_t desc(const unsigned short* w3, int N, int K) {
        return uniform_rsrc(w3, (int)((long)N * (K / 32) * 192));
    }
    __device__ __forceinline__ Split3BLoader(int N, int K, int n0, int tid) {
        const int row_bytes = (K / 32) * 192;
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const int q = tid + 256 * j;
            const int n = n0 + q / 12;
            off[j] = (q < NPIECES && n < N) ? n * row_bytes + (q % 12) * 16 : OOB;
        }
        uoff = 0;
    }
    __device__ __forceinline__ void begin(int kt) { uoff = kt * 192; }
    __device__ __forceinline__ u32x4_t operator()(int, int j, __amdgpu_buffer_rsrc_t rsrc) const {
        return __builtin_amdgcn_raw_buffer_load_b128(rsrc, off[j] + uoff, 0, 0);
    }
};

// split weights [n][K/32][2][32] fp16: thread piece q = tid + 256 j -> row q / 8, 16-byte piece q % 8
template <int NP, int NPIECES, int NT = 256>
struct Split2BLoader {
    int off[NP]; int uoff;
    __device__ __forceinline__ static __amdgpu_buffer_rsrc_t desc(const unsigned short* w2, int N, int K) {
        return uniform_rsrc(w2, (int)((long)N * (K / 32) * 128));
    }
    __device__ __forceinline__ Split2BLoader(int N, int K, int n0, int tid) {
        const int row_bytes = (K / 32) * 128;
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const int q = tid + NT * j;
            const int n = n0 + (q >> 3);
            off[j] = (q < NPIECES && n < N) ? n * row_bytes + (q & 7) * 16 : OOB;
        }
        uoff = 0;
    }
    __device__ __forceinline__ void begin(int kt) { uoff = kt * 128; }
    __device__ __forceinline__ g2_u32x4 operator()(int, int j, __amdgpu_buffer_rsrc_t rsrc) const {
        return __builtin_amdgcn_raw_buffer_load_b128(rsrc, off[j] + uoff, 0, 0);
    }
};

template <class G>
__device__ __forceinline__ void conv_epilogue(const ConvArgs& a, typename G::acc_t (&acc)[G::TM][G::TN], int m0, int n0, int b, int g) {
    // ---- epilogue -------------------------------------------------------------------------
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* oz = a.out + b * a.o_bstride + g * a.o_gstride;
    const float* rz = a.resid ? a.resid + b * a.r_bstride + g * a.r_gstride : nullptr;
    const float* bz = a.bias ? a.bias + g * a.bias_gstride : nullptr;
    unsigned amax = 0;
#pragma unroll
    for (int j = 0; j < G::TN; ++j) {
        const int n = n0 + G::acc_col(wave, lane, j);
        if (n >= a.n) continue;
        const float bv = bz ? bz[a.bias_period ? n % a.bias_period : n] : 0.f;
        int phase = 0, col = n;
        if (a.convt_u) { phase = n / a.convt_cout; col = n - phase * a.convt_cout; }
#pragma unroll
        for (int i = 0; i < G::TM; ++i) {
            // output row of accumulator element r of this 32x32 tile, or -1 (past M, or a transposed-conv row cropped away)
            auto out_row = [&](int r) -> long {
                const int m = m0 + G::acc_row(wave, lane, i, r);
                if (m >= a.m) return -1;
                if (!a.convt_u) return m;
                const long o = (long)m * a.convt_u + phase - a.convt_pad;
                return (o < 0 || o >= a.t_out) ? -1 : o;
            };
            // residual / accumulate operands: all 16 loads of the tile are issued before the first use (the straight
            // per-element "load, add, store" form serialised on s_waitcnt vmcnt(0) — one memory round trip per element —
            // and made the short-K layers with a residual epilogue-bound: out-proj 177 vs 277 TFLOP/s for QKV)
            constexpr int NR = G::NR;
            float rv[NR], av[NR];
            if (rz) {
#pragma unroll
                for (int r = 0; r < NR; ++r) { const long o = out_row(r); rv[r] = o >= 0 ? rz[o * a.ldr + col] : 0.f; }
            }
            if (a.accumulate) {
#pragma unroll
                for (int r = 0; r < NR; ++r) { const long o = out_row(r); av[r] = o >= 0 ? oz[o * a.ldo + col] : 0.f; }
            }
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                const long orow = out_row(r);
                if (orow < 0) continue;
                float v = fmaf(acc[i][j][r], a.out_scale, bv);      // out_scale is a power of two: exact
                if (a.act == KNNSVC_ACT_GELU) v = kn_gelu(v);
                else if (a.act == KNNSVC_ACT_LRELU) v = lrelu(v, a.act_slope);
                else if (a.act == KNNSVC_ACT_TANH) v = tanhf(v);
                if (a.out_split && n >= a.split_from) {
                    // f16x2 split layout for the next GEMM's A operand: element (row, c) -> hi at (c/32)*128 + (c%32)*2,
                    // lo 64 bytes further.  Lanes n and n^1 hold neighbouring columns of the same row: the even lane
                    // stores both hi halves, the odd lane both lo halves — one 4-byte store per lane, as in fp32 mode.
                    const float xs = v * a.split_scale;
                    const _Float16 h = (_Float16)xs;
                    const _Float16 l = (_Float16)(xs - (float)h);
                    const unsigned hl = (unsigned)__builtin_bit_cast(unsigned short, h) |
                                        ((unsigned)__builtin_bit_cast(unsigned short, l) << 16);
                    const unsigned pr = (unsigned)__builtin_amdgcn_mov_dpp((int)hl, 0xB1, 0xF, 0xF, true);   // lane ^ 1
                    const bool odd = (lane & 1) != 0;
                    const unsigned ow = odd ? ((pr >> 16) | (hl & 0xFFFF0000u)) : ((hl & 0xFFFFu) | (pr << 16));
                    const int ce = col & ~1;
                    char* ob = (char*)oz + orow * (long)a.ldo * 4 + (ce >> 5) * 128 + (ce & 31) * 2 + (odd ? 64 : 0);
                    *(unsigned*)ob = ow;
                    continue;
                }
                if (rz) v += rv[r];
                if (a.accumulate) v += av[r];
                if (a.div != 1.0f) v = v / a.div;
                if (a.out_absmax) { const unsigned ab = abs_bits(v); amax = ab > amax ? ab : amax; }
#ifdef KN_WHATIF_NOSTORE
                if (v == 123456.789f) oz[orow * a.ldo + col] = v;      // timing-only build: keeps the value live, stores nothing
#else
                oz[orow * a.ldo + col] = v;
#endif
            }
        }
    }
    if (a.out_absmax) publish_absmax(a.out_absmax, amax);
}

// Lean epilogue for plain (non-transposed) outputs whose byte extent fits 31 bits: rows are addressed through buffer
// resources sized to the valid rows, so the hardware range check replaces the per-element `m < M` tests (stores past the
// last row are dropped, loads return 0), offsets are 32-bit (row base + a wave-uniform multiple of the pitch per
// accumulator element) and nothing is computed in 64 bits.  With 128 accumulator registers live (128x64 wave tiles) the
// generic epilogue above spills ~350 registers; this one does not.
template <class G>
__device__ __forceinline__ void conv_epilogue_lin(const ConvArgs& a, typename G::acc_t (&acc)[G::TM][G::TN], int m0, int n0, int b, int g) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* oz = a.out + b * a.o_bstride + g * a.o_gstride;
    const float* rz = a.resid ? a.resid + b * a.r_bstride + g * a.r_gstride : nullptr;
    const float* bz = a.bias ? a.bias + g * a.bias_gstride : nullptr;
    const int ldo4 = a.ldo * 4, ldr4 = a.ldr * 4;
    const __amdgpu_buffer_rsrc_t o_rsrc = uniform_rsrc(oz, ((a.m - 1) * a.ldo + a.n) * 4);
    const __amdgpu_buffer_rsrc_t r_rsrc = uniform_rsrc(rz ? (const void*)rz : (const void*)oz, rz ? ((a.m - 1) * a.ldr + a.n) * 4 : 0);
    constexpr int NR = G::NR;
    unsigned amax = 0;
#pragma unroll
    for (int j = 0; j < G::TN; ++j) {
        const int n = n0 + G::acc_col(wave, lane, j);
        const bool nv = n < a.n;
        const float bv = (bz && nv) ? bz[a.bias_period ? n % a.bias_period : n] : 0.f;
        const bool odd = (lane & 1) != 0;
        const int ce = n & ~1;
        // byte offset of this lane's column inside a row: fp32 element, or its 4-byte slot of the f16x2 split layout
        const bool sp = a.out_split && n >= a.split_from;          // uniform per 32-column group
        const int cpart = sp ? (ce >> 5) * 128 + (ce & 31) * 2 + (odd ? 64 : 0) : n * 4;
#pragma unroll
        for (int i = 0; i < G::TM; ++i) {
            const int row0 = m0 + G::acc_row(wave, lane, i, 0);
            const int bo = nv ? row0 * ldo4 + cpart : OOB;
            const int br = nv ? row0 * ldr4 + n * 4 : OOB;
            float rv[NR], av[NR];
            if (rz) {
#pragma unroll
                for (int r = 0; r < NR; ++r)
                    rv[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r_rsrc, br + ((r & 3) + 8 * (r >> 2)) * ldr4, 0, 0));
            }
            if (a.accumulate) {
#pragma unroll
                for (int r = 0; r < NR; ++r)
                    av[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(o_rsrc, bo + ((r & 3) + 8 * (r >> 2)) * ldo4, 0, 0));
            }
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                float v = fmaf(acc[i][j][r], a.out_scale, bv);      // out_scale is a power of two: exact
                if (a.act == KNNSVC_ACT_GELU) v = kn_gelu(v);
                else if (a.act == KNNSVC_ACT_LRELU) v = lrelu(v, a.act_slope);
                else if (a.act == KNNSVC_ACT_TANH) v = tanhf(v);
                const int off = bo + ((r & 3) + 8 * (r >> 2)) * ldo4;
                if (sp) {       // see conv_epilogue: lanes n and n^1 exchange halves, one 4-byte store each
                    const float xs = v * a.split_scale;
                    const _Float16 h = (_Float16)xs;
                    const _Float16 l = (_Float16)(xs - (float)h);
                    const unsigned hl = (unsigned)__builtin_bit_cast(unsigned short, h) |
                                        ((unsigned)__builtin_bit_cast(unsigned short, l) << 16);
                    const unsigned pr = (unsigned)__builtin_amdgcn_mov_dpp((int)hl, 0xB1, 0xF, 0xF, true);   // lane ^ 1
                    const unsigned ow = odd ? ((pr >> 16) | (hl & 0xFFFF0000u)) : ((hl & 0xFFFFu) | (pr << 16));
                    __builtin_amdgcn_raw_buffer_store_b32(ow, o_rsrc, off, 0, 0);
                    continue;
                }
                if (rz) v += rv[r];
                if (a.accumulate) v += av[r];
                if (a.div != 1.0f) v = v / a.div;
                // (rows past M hold act(bias) of an all-zero A row: they are not stored, but may enter the bound — harmless)
                if (a.out_absmax) { const unsigned ab = abs_bits(v); amax = ab > amax ? ab : amax; }
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), o_rsrc, off, 0, 0);
            }
        }
    }
    if (a.out_absmax) publish_absmax(a.out_absmax, amax);
}

template <class G>
__device__ __forceinline__ void conv_epilogue_any(const ConvArgs& a, typename G::acc_t (&acc)[G::TM][G::TN], int m0, int n0, int b, int g) {
    if (a.lin) conv_epilogue_lin<G>(a, acc, m0, n0, b, g);       // wave-uniform
    else conv_epilogue<G>(a, acc, m0, n0, b, g);
}

// Epilogue of the 128x128-wave-tile kernel (Gemm2Quad): 256 accumulator values per lane, one block per CU, nothing else on
// the CU to hide it.  Two things made the column-per-lane epilogue (conv_epilogue_lin) as long as the K = 1024 main loop here:
// 256 four-byte store instructions per wave, and — worse — CODE SIZE: bias / activation / residual arithmetic replicated for
// 256 register-resident elements is ~350 KB of straight-line code that every wave runs through once per tile, i.e. the
// epilogue ran at instruction-fetch speed (a build without any global store took exactly as long).  So:
//   * each 32-row band of the wave tile goes through a wave-private LDS patch (the operand ring is idle by now): the only
//     fully unrolled code is accumulator * scale -> ds_write (3 instructions per element);
//   * a ROLLED loop then takes the band back, four consecutive columns of one row per lane and trip: bias, activation,
//     residual / accumulate operands (16-byte loads), divide, range slot, 16-byte store (two 512-byte row segments per
//     instruction) or the split layout — a few hundred instructions executed 64 times per wave instead of 40 000 once.
// Needs n % 4 == 0, 16-byte aligned rows and a split boundary on a wave-tile column (multiple of 128).
template <class G>
__device__ __forceinline__ void conv_epilogue_wide(const ConvArgs& a, typename G::acc_t (&acc)[G::TM][G::TN], float* lds_generic,
                                                   int m0, int n0, int b, int g) {
    typedef __attribute__((address_space(3))) float lds_f;
    typedef __attribute__((address_space(3))) f32x4 lds_f4;
    constexpr int ETN = G::NR == 16 ? G::TN : G::TN / 2, ETM = G::NR == 16 ? G::TM : G::TM / 2;   // wave tile in 32 x 32 units
    constexpr int PITCH = ETN * 32 + 4;                          // floats per patch row (+4: rows 4 apart land on other banks)
    constexpr int TRIPS = (32 * ETN * 32 / 4) / 64;              // float4 per lane and band
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 31, lh = lane >> 5;
    lds_f* patch = (lds_f*)lds_generic + wave * (32 * PITCH);
    float* oz = a.out + b * a.o_bstride + g * a.o_gstride;
    const float* rz = a.resid ? a.resid + b * a.r_bstride + g * a.r_gstride : nullptr;
    const float* bz = a.bias ? a.bias + g * a.bias_gstride : nullptr;
    const __amdgpu_buffer_rsrc_t o_rsrc = uniform_rsrc(oz, ((a.m - 1) * a.ldo + a.n) * 4);
    const __amdgpu_buffer_rsrc_t r_rsrc = uniform_rsrc(rz ? (const void*)rz : (const void*)oz, rz ? ((a.m - 1) * a.ldr + a.n) * 4 : 0);
    const int wrow0 = m0 + (wave / G::WN) * ETM * 32, wcol0 = n0 + (wave % G::WN) * ETN * 32;
    // in the read-back loop a lane always owns the same four columns: (lane % 32) * 4 .. + 3 of the wave tile
    const int c = (lane & (ETN * 8 - 1)) * 4, n = wcol0 + c;
    const int row_in_trip = lane / (ETN * 8);                   // trip `it` covers rows it * (64 / (TN*8)) + this
    const bool nv = n < a.n;                                       // n % 4 == 0 and a.n % 4 == 0: all four columns or none
    f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
    if (bz && nv) {
        if (a.bias_period) { for (int e = 0; e < 4; ++e) bias4[e] = bz[(n + e) % a.bias_period]; }
        else bias4 = *(const f32x4*)(bz + n);
    }
    const bool sp = a.out_split && wcol0 >= a.split_from;          // wave-uniform (split_from is a multiple of the wave tile width)
    const int act = a.act;
    unsigned amax = 0;
#pragma unroll
    for (int i = 0; i < ETM; ++i) {
        if constexpr (G::NR == 16) {
#pragma unroll
            for (int j = 0; j < G::TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    patch[((r & 3) + 8 * (r >> 2) + 4 * lh) * PITCH + j * 32 + li] = acc[i][j][r] * a.out_scale;   // power of two: exact
        } else {                      // 16x16 tiles: C at col = lane & 15, row = 4 (lane >> 4) + reg; a band is two tile rows
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int j = 0; j < G::TN; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        patch[(t * 16 + 4 * (lane >> 4) + r) * PITCH + j * 16 + (lane & 15)] = acc[2 * i + t][j][r] * a.out_scale;
        }
        // the patch is private to this wave and LDS operations of one wave complete in order: no barrier.
        // Software-pipelined by one trip: the LDS read and the residual / accumulate loads of trip it + 1 are in flight while
        // trip it runs its arithmetic and its store (the loop is rolled, so nothing else would hide their latency).
        const int rstep = 64 / (ETN * 8);
        const int mrow0 = wrow0 + i * 32 + row_in_trip;
        auto ld_patch = [&](int it) -> f32x4 { return *(const lds_f4*)(patch + (it * rstep + row_in_trip) * PITCH + c); };
        auto ld_res = [&](int it) -> f32x4 {
            return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_rsrc, nv ? ((mrow0 + it * rstep) * a.ldr + n) * 4 : OOB, 0, 0));
        };
        auto ld_acc = [&](int it) -> f32x4 {
            return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(o_rsrc, nv ? ((mrow0 + it * rstep) * a.ldo + n) * 4 : OOB, 0, 0));
        };
        const bool use_res = rz != nullptr && !sp, use_acc = a.accumulate && !sp;
        f32x4 vn = ld_patch(0), rn = {0.f, 0.f, 0.f, 0.f}, an = rn;
        if (use_res) rn = ld_res(0);
        if (use_acc) an = ld_acc(0);
#pragma unroll 1
        for (int it = 0; it < TRIPS; ++it) {
            f32x4 v = vn + bias4;
            const f32x4 rv = rn, av = an;
            if (it + 1 < TRIPS) {
                vn = ld_patch(it + 1);
                if (use_res) rn = ld_res(it + 1);
                if (use_acc) an = ld_acc(it + 1);
            }
            if (act == KNNSVC_ACT_GELU) {
#pragma unroll
                for (int e = 0; e < 4; e += 2) { const f32x2 gv = kn_gelu2((f32x2){v[e], v[e + 1]}); v[e] = gv[0]; v[e + 1] = gv[1]; }     // pairs on packed fp32: bit-identical to kn_gelu
            } else if (act == KNNSVC_ACT_LRELU) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = lrelu(v[e], a.act_slope);
            } else if (act == KNNSVC_ACT_TANH) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = tanhf(v[e]);
            }
            const int m = mrow0 + it * rstep;
            if (sp) {       // f16x2 split layout: (row, n) -> hi at (n/32)*128 + (n%32)*2, lo 64 bytes further
                g2_u32x2 hi, lo;
                f16x2_split4(v, a.split_scale, hi, lo);
                const int off = nv ? m * a.ldo * 4 + (n >> 5) * 128 + (n & 31) * 2 : OOB;
                __builtin_amdgcn_raw_buffer_store_b64(hi, o_rsrc, off, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b64(lo, o_rsrc, off == OOB ? OOB : off + 64, 0, 0);
                continue;
            }
            const int off = nv ? (m * a.ldo + n) * 4 : OOB;
            if (use_res) v += rv;
            if (use_acc) v += av;
            if (a.div != 1.0f) v = v / a.div;
            if (a.out_absmax) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { const unsigned ab = abs_bits(v[e]); amax = ab > amax ? ab : amax; }
            }
#ifdef KN_T_NOSTORE       // timing aid: everything but the global store
            if (v[0] == 123456.789f) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), o_rsrc, off, 0, 0);
#else
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), o_rsrc, off, 0, 0);
#endif
        }
    }
    if (a.out_absmax) publish_absmax(a.out_absmax, amax);
}

// Specialised forms of conv_epilogue_wide for the epilogues the encoder's large GEMMs actually have.  An in-kernel phase
// trace (tools/quad_prof.sh, -DKN_QUAD_PROF) showed where a K = 1024 tile's time went: prologue 2 us, main loop 61 us
// (= 560 TFLOP/s fp32-equivalent over the chip), epilogue 26-28 us without and 46-47 us with a residual operand — and the
// cause was not memory: every trip of the rolled loop ran the GENERIC path, if-converted by the compiler — four IEEE
// divisions by `div` (selected away afterwards), tanh / leaky-ReLU / GELU bodies under exec masks, residual and accumulate
// selects — ~250 VALU instructions per trip for 4 useful adds, 64 trips per wave; and the residual load was prefetched only
// one trip ahead, adding one memory latency per trip.  Here the variant is a template parameter:
//   ACT  0 = none, 1 = GELU;   RES: residual operand (fp32 output);   without RES the output may be split from a wave-tile
//   column on (a.out_split / a.split_from, wave-uniform).  No accumulate, div == 1, no range slot — anything else takes the
//   generic form.  With RES the 16 residual pieces of the NEXT band are requested before the current band's trips run.
template <class G, int ACT, bool RES>
__device__ __forceinline__ void conv_epilogue_wide_fast(const ConvArgs& a, typename G::acc_t (&acc)[G::TM][G::TN], float* lds_generic,
                                                        int m0, int n0, int b, int g) {
    static_assert(G::NR == 4, "16x16-tile accumulator layout");
    static_assert(!(RES && ACT != 0), "residual variant has no activation");
    typedef __attribute__((address_space(3))) float lds_f;
    typedef __attribute__((address_space(3))) f32x4 lds_f4;
    constexpr int ETN = G::TN / 2, ETM = G::TM / 2;
    constexpr int PITCH = ETN * 32 + 4;
    constexpr int TRIPS = (32 * ETN * 32 / 4) / 64;
    constexpr int RSTEP = 64 / (ETN * 8);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    lds_f* patch = (lds_f*)lds_generic + wave * (32 * PITCH);
    float* oz = a.out + b * a.o_bstride + g * a.o_gstride;
    const float* rz = RES ? a.resid + b * a.r_bstride + g * a.r_gstride : nullptr;
    const float* bz = a.bias ? a.bias + g * a.bias_gstride : nullptr;
    const __amdgpu_buffer_rsrc_t o_rsrc = uniform_rsrc(oz, ((a.m - 1) * a.ldo + a.n) * 4);
    const __amdgpu_buffer_rsrc_t r_rsrc = uniform_rsrc(RES ? (const void*)rz : (const void*)oz, RES ? ((a.m - 1) * a.ldr + a.n) * 4 : 0);
    const int wrow0 = m0 + (wave / G::WN) * ETM * 32, wcol0 = n0 + (wave % G::WN) * ETN * 32;
    const int c = (lane & (ETN * 8 - 1)) * 4, n = wcol0 + c;
    const int row_in_trip = lane / (ETN * 8);
    const bool nv = n < a.n;
    f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
    if (bz && nv) {
        if (a.bias_period) { for (int e = 0; e < 4; ++e) bias4[e] = bz[(n + e) % a.bias_period]; }
        else bias4 = *(const f32x4*)(bz + n);
    }
    const bool sp = !RES && a.out_split && wcol0 >= a.split_from;          // wave-uniform
    const float oscale = a.out_scale, sscale = a.split_scale;
    const int row_pitch_o = a.ldo * 4 * RSTEP, row_pitch_r = RES ? a.ldr * 4 * RSTEP : 0;
    // residual pieces of up to DEPTH bands in flight (16 x 16 bytes per lane and band): requested DEPTH bands ahead of their
    // use — with one band ahead every band still waited a full (loaded) memory latency: 34 us of epilogue per tile, 4 x ~7 us
    constexpr int DEPTH = 3;
    f32x4 rb[RES ? DEPTH : 1][RES ? TRIPS : 1];
    auto load_band = [&](int i, f32x4 (&dst)[RES ? TRIPS : 1]) {
        if constexpr (RES) {
            int off = nv ? ((wrow0 + i * 32 + row_in_trip) * a.ldr + n) * 4 : OOB;
#pragma unroll
            for (int it = 0; it < TRIPS; ++it) {
                dst[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_rsrc, off, 0, 0));
                if (nv) off += row_pitch_r;
            }
        }
    };
    if constexpr (RES) {
#pragma unroll
        for (int i = 0; i < DEPTH && i < ETM; ++i) load_band(i, rb[i]);
    }
#pragma unroll
    for (int i = 0; i < ETM; ++i) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int j = 0; j < G::TN; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    patch[(t * 16 + 4 * (lane >> 4) + r) * PITCH + j * 16 + (lane & 15)] = acc[2 * i + t][j][r] * oscale;
        f32x4 (&rcur)[RES ? TRIPS : 1] = rb[RES ? i % DEPTH : 0];
        const lds_f* prow = patch + row_in_trip * PITCH + c;
        const int mrow = wrow0 + i * 32 + row_in_trip;
        if (sp) {       // f16x2 split layout: (row, n) -> hi at (n/32)*128 + (n%32)*2, lo 64 bytes further
            int off = nv ? mrow * a.ldo * 4 + (n >> 5) * 128 + (n & 31) * 2 : OOB;
#pragma unroll 4
            for (int it = 0; it < TRIPS; ++it) {
                f32x4 v = *(const lds_f4*)(prow + it * RSTEP * PITCH) + bias4;
                if constexpr (ACT == 1) {
#pragma unroll
                    for (int e = 0; e < 4; e += 2) { const f32x2 gv = kn_gelu2((f32x2){v[e], v[e + 1]}); v[e] = gv[0]; v[e + 1] = gv[1]; }     // pairs on packed fp32: bit-identical to kn_gelu
                }
                g2_u32x2 hi, lo;
                f16x2_split4(v, sscale, hi, lo);
                __builtin_amdgcn_raw_buffer_store_b64(hi, o_rsrc, off, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b64(lo, o_rsrc, off, 64, 0);      // OOB + 64 is still out of range
                if (nv) off += row_pitch_o;
            }
        } else {
            int off = nv ? (mrow * a.ldo + n) * 4 : OOB;
            if constexpr (RES) {
#pragma unroll                      // fully: rcur is a register array
                for (int it = 0; it < TRIPS; ++it) {
                    f32x4 v = *(const lds_f4*)(prow + it * RSTEP * PITCH) + bias4;
                    v += rcur[it];
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), o_rsrc, off, 0, 0);
                    if (nv) off += row_pitch_o;
                }
                if (i + DEPTH < ETM) load_band(i + DEPTH, rb[i % DEPTH]);
            } else {
#pragma unroll 4
                for (int it = 0; it < TRIPS; ++it) {
                    f32x4 v = *(const lds_f4*)(prow + it * RSTEP * PITCH) + bias4;
                    if constexpr (ACT == 1) {
#pragma unroll
                        for (int e = 0; e < 4; e += 2) { const f32x2 gv = kn_gelu2((f32x2){v[e], v[e + 1]}); v[e] = gv[0]; v[e + 1] = gv[1]; }     // pairs on packed fp32: bit-identical to kn_gelu
                    }
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), o_rsrc, off, 0, 0);
                    if (nv) off += row_pitch_o;
                }
            }
        }
    }
}

template <class G, int VEC>
__global__ __launch_bounds__(256) void conv_gemm_kernel(ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    resolve_dyn(a);
    if (a.out_absmax) a.out_absmax = kn_stripe(a.out_absmax);
    const int z = blockIdx.z;
    const int b = z / a.groups, g = z - b * a.groups;
    const int m0 = blockIdx.x * G::BM, n0 = blockIdx.y * G::BN;
    const float* xz = a.x + b * a.x_bstride + g * a.x_gstride;
    const float* wz = a.w + g * a.w_gstride;

    f32x16 acc[G::TM][G::TN];
#pragma unroll
    for (int i = 0; i < G::TM; ++i)
#pragma unroll
        for (int j = 0; j < G::TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if constexpr (VEC == 8) {
        FastALoader<G::A_F4> al(a, m0, threadIdx.x);
        FastBLoader<G::B_F4> bl(a.n, a.K, n0, threadIdx.x);
        G::mainloop(lds, a.K / 32, al, bl, acc, FastALoader<G::A_F4>::desc(a, xz), FastBLoader<G::B_F4>::desc(wz, a.n, a.K));
    } else {
        ALoader<VEC, G::A_F4> al(a, xz, m0, threadIdx.x);
        BLoader<VEC, G::B_F4> bl(wz, a.n, a.K, n0);
        G::mainloop(lds, (a.K + 31) / 32, al, bl, acc, 0, 0);
    }

    conv_epilogue<G>(a, acc, m0, n0, b, g);
}

template <class G>
__global__ __launch_bounds__(256, 3) void conv_gemm3_kernel(ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    resolve_dyn(a);
    if (a.out_absmax) a.out_absmax = kn_stripe(a.out_absmax);
    const int z = blockIdx.z;
    const int b = z / a.groups, g = z - b * a.groups;
    const int m0 = blockIdx.x * G::BM, n0 = blockIdx.y * G::BN;
    const float* xz = a.x + b * a.x_bstride + g * a.x_gstride;
    const unsigned short* wz = a.w3 + (long)g * a.n * (a.K / 32) * 96;      // 96 ushorts = 192 B per (row, slab)

    f32x16 acc[G::TM][G::TN];
#pragma unroll
    for (int i = 0; i < G::TM; ++i)
#pragma unroll
        for (int j = 0; j < G::TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    FastALoader<G::A_F4> al(a, m0, threadIdx.x);
    Split3BLoader<G::B_P, G::B_PIECES> bl(a.n, a.K, n0, threadIdx.x);
    G::mainloop(lds, a.K / 32, al, bl, acc, FastALoader<G::A_F4>::desc(a, xz), Split3BLoader<G::B_P, G::B_PIECES>::desc(wz, a.n, a.K));
    conv_epilogue<G>(a, acc, m0, n0, b, g);
}

template <class G, bool A2>
__global__ __launch_bounds__(256, 3) void conv_gemm2_kernel(ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    resolve_scales(a);
    const int z = blockIdx.z;
    // group-major: the slices that run side by side on an XCD then share a group's weights through its L2 (batch-major, the 64 blocks
    // resident on an XCD spanned all 16 groups of the positional conv — 32 MB of weights against 4 MB of L2)
    const int nbz = (int)gridDim.z / a.groups;
    const int g = z / nbz, b = z - g * nbz;
    // XCD-aware tile order (see conv_gemm2big_kernel): ids congruent mod 8 share an L2; a group is 8 row tiles x all
    // column tiles, so the column tiles of one row tile run on one XCD, next to each other in time
    // and the columns are walked in patches of CW tiles, so an XCD's ~96 resident blocks form a (12 x 8)-ish patch
    // that shares both A and B panels through its L2.
    const int gy = (a.n + G::BN - 1) / G::BN;
    const int gx8 = (int)gridDim.x / gy;                      // row tiles padded to a multiple of 8
    constexpr int CW = 8;
    int L = blockIdx.x;
    const int full = (gy / CW) * CW * gx8;                    // ids covered by full-width column patches
    int c0, cw;
    if (L < full) { c0 = (L / (CW * gx8)) * CW; cw = CW; L -= (c0 / CW) * CW * gx8; }
    else { c0 = (gy / CW) * CW; cw = gy - c0; L -= full; }
    const int grp = L / (8 * cw), rem = L - grp * 8 * cw;
    // (rem + z) & 7: which XCD gets which row tile of a group of 8 rotates with the batch / group index.  The padding rows are the
    // same in every slice of a batched launch: unrotated, the positional conv's 6 row tiles per (chunk, group) — 336 slices — left
    // XCDs 6 and 7 nothing but padding ids, a quarter of the chip idle for the whole launch (1.76 ms against 1.37 for the same work
    // as one long sequence); gemm2_core.h's quad_order_decode rotates for the same reason.
    // a.plain: a single column tile has no operand panel to share between column tiles, so padding the row tiles to groups of 8
    // buys nothing and the padding ids still queue for LDS before they can exit (6 row tiles: a quarter of all ids)
    const int m0 = a.plain ? (int)blockIdx.x * G::BM : (grp * 8 + ((rem + (int)blockIdx.z) & 7)) * G::BM, n0 = a.plain ? 0 : (c0 + (rem >> 3)) * G::BN;
    if (m0 >= a.m) return;                                    // padding of the last group
    const float* xz = a.x + b * a.x_bstride + g * a.x_gstride;
    const unsigned short* wz = a.w2 + (long)g * a.n * (a.K / 32) * 64;      // 64 halves = 128 B per (row, slab)

    typename G::acc_t acc[G::TM][G::TN];
#pragma unroll
    for (int i = 0; i < G::TM; ++i)
#pragma unroll
        for (int j = 0; j < G::TN; ++j)
#pragma unroll
            for (int r = 0; r < G::NR; ++r) acc[i][j][r] = 0.f;
    FastALoader<G::A_F4> al(a, m0, threadIdx.x);
    Split2BLoader<G::B_P, G::B_PIECES> bl(a.n, a.K, n0, threadIdx.x);
    G::template mainloop<A2>(lds, a.K / 32, al, bl, acc, FastALoader<G::A_F4>::desc(a, xz), Split2BLoader<G::B_P, G::B_PIECES>::desc(wz, a.n, a.K), a.a_scale);
    if (a.wide) {                                       // wave-uniform: bias (+ residual) (+ range slot) only, 16-byte rows
        __syncthreads();                                // every wave is done with the operand stage: it becomes the patches
        if (a.convt_u) conv_epilogue_wide32<G, false, true>(a, acc, lds, m0, n0, b, g);
        else if (a.resid) conv_epilogue_wide32<G, true>(a, acc, lds, m0, n0, b, g);
        else conv_epilogue_wide32<G, false>(a, acc, lds, m0, n0, b, g);
    } else conv_epilogue_any<G>(a, acc, m0, n0, b, g);
}

// Epilogue of the windowed kernels for what the generator's ResBlock convolutions have — bias, a residual operand or none, a
// range slot or none — through wave-private LDS patches like conv_epilogue_wide (the operand stages are idle by now), on the
// 32x32-tile accumulator layout: a band of 32 rows is written column-per-lane (one ds_write_b32 per element, scaled: a power
// of two), taken back as four consecutive columns of one row per lane, and leaves as 16-byte stores with 16-byte residual
// loads (requested before the band is written).  The column-per-lane form (conv_epilogue_lin) spends ~27 VALU and two 4-byte
// memory instructions per element — a quarter of a k = 3 launch at C = 128.  Same arithmetic per element, same bits.
// CONVT: a transposed convolution's output (column n = phase * cout + channel, row m -> output row m * u + phase - pad, rows cropped
// to [0, t_out)): the four columns of a lane lie in one phase (cout % 4 == 0); the range slot only sees rows that are stored, as in
// conv_epilogue.  The plain form leaves rows past M to the buffer range check, as conv_epilogue_lin does.
template <class G, bool RES, bool CONVT = false>
__device__ __forceinline__ void conv_epilogue_wide32(const ConvArgs& a, typename G::acc_t (&acc)[G::TM][G::TN], float* lds_generic,
                                                     int m0, int n0, int b, int g) {
    static_assert(G::NR == 16, "32x32-tile accumulator layout");
    static_assert(!(RES && CONVT), "no residual on the transposed form");
    typedef __attribute__((address_space(3))) float lds_f;
    typedef __attribute__((address_space(3))) f32x4 lds_f4;
    constexpr int ETN = G::TN, ETM = G::TM;
    constexpr int PITCH = ETN * 32 + 4;
    constexpr int TRIPS = (32 * ETN * 32 / 4) / 64;
    constexpr int RSTEP = 64 / (ETN * 8);
    static_assert(4 * 32 * PITCH * 4 <= G::LDS_BYTES, "patches fit the operand stages");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 31, lh = lane >> 5;
    lds_f* patch = (lds_f*)lds_generic + wave * (32 * PITCH);
    float* oz = a.out + b * a.o_bstride + g * a.o_gstride;
    const float* rz = RES ? a.resid + b * a.r_bstride + g * a.r_gstride : nullptr;
    const float* bz = a.bias ? a.bias + g * a.bias_gstride : nullptr;
    const __amdgpu_buffer_rsrc_t o_rsrc = uniform_rsrc(oz, CONVT ? ((a.t_out - 1) * a.ldo + a.convt_cout) * 4 : ((a.m - 1) * a.ldo + a.n) * 4);
    const __amdgpu_buffer_rsrc_t r_rsrc = uniform_rsrc(RES ? (const void*)rz : (const void*)oz, RES ? ((a.m - 1) * a.ldr + a.n) * 4 : 0);
    const int wrow0 = m0 + (wave / G::WN) * ETM * 32, wcol0 = n0 + (wave % G::WN) * ETN * 32;
    const int c = (lane & (ETN * 8 - 1)) * 4, n = wcol0 + c;
    const int row_in_trip = lane / (ETN * 8);
    const bool nv = n < a.n;
    int phase = 0, col = n;
    if (CONVT) { phase = n / a.convt_cout; col = n - phase * a.convt_cout; }
    f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
    if (bz && nv) bias4 = *(const f32x4*)(bz + (a.bias_period ? n % a.bias_period : n));
    unsigned amax = 0;
#pragma unroll
    for (int i = 0; i < ETM; ++i) {
        const int mrow0 = wrow0 + i * 32 + row_in_trip;
        f32x4 rv[TRIPS];
        if (RES) {
#pragma unroll
            for (int it = 0; it < TRIPS; ++it)
                rv[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_rsrc, nv ? ((mrow0 + it * RSTEP) * a.ldr + n) * 4 : OOB, 0, 0));
        }
#pragma unroll
        for (int j = 0; j < G::TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                patch[((r & 3) + 8 * (r >> 2) + 4 * lh) * PITCH + j * 32 + li] = acc[i][j][r] * a.out_scale;      // power of two: exact
        // (the patch is private to this wave and LDS operations of one wave complete in order: no barrier)
#pragma unroll
        for (int it = 0; it < TRIPS; ++it) {
            f32x4 v = *(const lds_f4*)(patch + (it * RSTEP + row_in_trip) * PITCH + c) + bias4;
            if (RES) v += rv[it];
            const int m = mrow0 + it * RSTEP;
            bool ok = nv;
            int off;
            if (CONVT) {
                const int o = m * a.convt_u + phase - a.convt_pad;
                ok = ok && m < a.m && o >= 0 && o < a.t_out;
                off = (o * a.ldo + col) * 4;
            } else off = (m * a.ldo + n) * 4;
            if (a.out_absmax && (!CONVT || ok)) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { const unsigned ab = abs_bits(v[e]); amax = ab > amax ? ab : amax; }
            }
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), o_rsrc, ok ? off : OOB, 0, 0);
        }
    }
    if (a.out_absmax) publish_absmax(a.out_absmax, amax);
}

// stride-1 multi-tap convolutions: one window of BM + HALO input rows per 32-channel slab, every tap reads it shifted
// (gemm2_core.h, Gemm2Win)
template <class G>
__device__ __forceinline__ void conv_gemm2win_body(ConvArgs& a, float* lds) {
    resolve_scales(a);
    const int z = blockIdx.z;
    // group-major: the slices that run side by side on an XCD then share a group's weights through its L2 (batch-major, the 64 blocks
    // resident on an XCD spanned all 16 groups of the positional conv — 32 MB of weights against 4 MB of L2)
    const int nbz = (int)gridDim.z / a.groups;
    const int g = z / nbz, b = z - g * nbz;
    const int gy = (a.n + G::BN - 1) / G::BN;                 // same XCD-aware column-patch order as conv_gemm2_kernel
    const int gx8 = (int)gridDim.x / gy;
    constexpr int CW = 8;
    int L = blockIdx.x;
    const int full = (gy / CW) * CW * gx8;
    int c0, cw;
    if (L < full) { c0 = (L / (CW * gx8)) * CW; cw = CW; L -= (c0 / CW) * CW * gx8; }
    else { c0 = (gy / CW) * CW; cw = gy - c0; L -= full; }
    const int grp = L / (8 * cw), rem = L - grp * 8 * cw;
    // (rem + z) & 7: which XCD gets which row tile of a group of 8 rotates with the batch / group index.  The padding rows are the
    // same in every slice of a batched launch: unrotated, the positional conv's 6 row tiles per (chunk, group) — 336 slices — left
    // XCDs 6 and 7 nothing but padding ids, a quarter of the chip idle for the whole launch (1.76 ms against 1.37 for the same work
    // as one long sequence); gemm2_core.h's quad_order_decode rotates for the same reason.
    // a.plain: a single column tile has no operand panel to share between column tiles, so padding the row tiles to groups of 8
    // buys nothing and the padding ids still queue for LDS before they can exit (6 row tiles: a quarter of all ids)
    const int m0 = a.plain ? (int)blockIdx.x * G::BM : (grp * 8 + ((rem + (int)blockIdx.z) & 7)) * G::BM, n0 = a.plain ? 0 : (c0 + (rem >> 3)) * G::BN;
    if (m0 >= a.m) return;
    const float* xz = a.x + b * a.x_bstride + g * a.x_gstride;
    const unsigned short* wz = a.w2 + (long)g * a.n * (a.K / 32) * 64;

    typename G::acc_t acc[G::TM][G::TN];
#pragma unroll
    for (int i = 0; i < G::TM; ++i)
#pragma unroll
        for (int j = 0; j < G::TN; ++j)
#pragma unroll
            for (int r = 0; r < G::NR; ++r) acc[i][j][r] = 0.f;
    Split2BLoader<G::B_P, G::B_PIECES> bl(a.n, a.K, n0, threadIdx.x);
    const int tid = threadIdx.x;
    const int w_off0 = ((m0 - a.pad + (tid >> 3)) * a.ldx + (tid & 7) * 4) * 4;
    if constexpr (G::DEEP)
        G::mainloop_deep(lds, a.cin / 32, a.taps, a.dil, w_off0, a.ldx * 4, bl, acc, FastALoader<1>::desc(a, xz),
                         Split2BLoader<G::B_P, G::B_PIECES>::desc(wz, a.n, a.K), a.a_scale, a.a_slope);
    else
        G::mainloop(lds, a.cin / 32, a.taps, a.dil, w_off0, a.ldx * 4, bl, acc, FastALoader<1>::desc(a, xz),
                    Split2BLoader<G::B_P, G::B_PIECES>::desc(wz, a.n, a.K), a.a_scale, a.a_slope);
    if (a.wide) {                                       // wave-uniform
        __syncthreads();                                // every wave is done with the operand stages: they become the patches
        if (a.resid) conv_epilogue_wide32<G, true>(a, acc, lds, m0, n0, b, g);
        else conv_epilogue_wide32<G, false>(a, acc, lds, m0, n0, b, g);
    } else conv_epilogue_any<G>(a, acc, m0, n0, b, g);
}

template <class G, int MINB>
__global__ __launch_bounds__(256, MINB) void conv_gemm2win_kernel(ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    conv_gemm2win_body<G>(a, lds);
}

// Several convolutions of the same output shape in ONE launch (round 5): blockIdx.y picks the descriptor.  The generator's three
// ResBlock branches of a stage (kernel sizes 3 / 7 / 11: hifigan/ddsp_models.py:206-227) only share their input; until round 4
// they ran on three streams, and whether those streams really overlapped was up to HIP's stream -> hardware-queue mapping (34.8 vs
// 38.3 ms per bench step depending on how many streams the process had created before).  Here the branches' launches of one step
// are one grid: the dispatcher sees all their workgroups at once, nothing depends on queues.  Same body, same descriptors: same
// bits as the separate launches.  The heaviest descriptor (most taps) should come first: workgroups are dispatched y-major.
constexpr int KN_MAX_MULTI = 4;
struct ConvArgsN { ConvArgs b[KN_MAX_MULTI]; };
template <class G, int MINB>
__global__ __launch_bounds__(256, MINB) void conv_gemm2win_multi_kernel(ConvArgsN args) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    ConvArgs a = args.b[blockIdx.y];
    conv_gemm2win_body<G>(a, lds);
}

template <class G, int MINB>
int launch2win(const ConvArgs& a, int batches, hipStream_t st) {
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute((const void*)conv_gemm2win_kernel<G, MINB>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                G::LDS_BYTES) != hipSuccess)
            return knnsvc_fail(KNNSVC_EHIP, "conv_gemm: hipFuncSetAttribute failed");
        attr = true;
    }
    const long gx = cdiv64(a.m, G::BM), gy = cdiv64(a.n, G::BN);
    ConvArgs ap = a;
    ap.plain = gy == 1 && gx % 8 != 0;
    const long gx8 = ap.plain ? gx : cdiv64(gx, 8) * 8;
    dim3 grid((unsigned)(gx8 * gy), 1, (unsigned)(batches * a.groups));
    hipLaunchKernelGGL((conv_gemm2win_kernel<G, MINB>), grid, dim3(256), G::LDS_BYTES, st, ap);
    return knnsvc_check_launch("conv_gemm2win");
}

template <class G, int MINB>
int launch2win_multi(const ConvArgsN& an, int count, hipStream_t st) {
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute((const void*)conv_gemm2win_multi_kernel<G, MINB>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                G::LDS_BYTES) != hipSuccess)
            return knnsvc_fail(KNNSVC_EHIP, "conv_gemm_multi: hipFuncSetAttribute failed");
        attr = true;
    }
    const ConvArgs& a = an.b[0];
    const long gx = cdiv64(a.m, G::BM), gy = cdiv64(a.n, G::BN);
    ConvArgsN ap = an;
    const int plain = gy == 1 && gx % 8 != 0;
    for (int i = 0; i < KN_MAX_MULTI; ++i) ap.b[i].plain = plain;
    const long gx8 = plain ? gx : cdiv64(gx, 8) * 8;
    dim3 grid((unsigned)(gx8 * gy), (unsigned)count, 1);
    hipLaunchKernelGGL((conv_gemm2win_multi_kernel<G, MINB>), grid, dim3(256), G::LDS_BYTES, st, ap);
    return knnsvc_check_launch("conv_gemm2win_multi");
}

// 256x256 block / 128x128 wave tiles, hand-pipelined loop (gemm2_core.h, Gemm2QuadS): A2 activations + split weights
// EPI: 0 = generic epilogue (conv_epilogue_wide), 1 = bias (+ split columns), 2 = bias + GELU (+ split columns), 3 = bias + residual
template <class G, int EPI = 0>
__global__ __launch_bounds__(256, 1) void conv_gemm2quad_kernel(ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
#ifdef KN_QUAD_PROF
    if (threadIdx.x == 0 && blockIdx.x < 8192) kn_quad_prof_buf[blockIdx.x * 4 + 0] = (long long)__builtin_amdgcn_s_memrealtime();
#endif
    resolve_scales(a);
    const int z = blockIdx.z;
    const int b = z / a.groups, g = z - b * a.groups;
    // XCD-aware order (gemm2_core.h, quad_order_decode)
    int mt, nt;
    if (!quad_order_decode((int)blockIdx.x, (a.m + G::BM - 1) / G::BM, (a.n + G::BN - 1) / G::BN, mt, nt, (int)blockIdx.z)) return;
    const int m0 = mt * G::BM, n0 = nt * G::BN;
    const float* xz = a.x + b * a.x_bstride + g * a.x_gstride;
    const unsigned short* wz = a.w2 + (long)g * a.n * (a.K / 32) * 64;

    typename G::acc_t acc[G::TM][G::TN];
#pragma unroll
    for (int i = 0; i < G::TM; ++i)
#pragma unroll
        for (int j = 0; j < G::TN; ++j)
#pragma unroll
            for (int r = 0; r < G::NR; ++r) acc[i][j][r] = 0.f;
    const int M = a.m, row_step = a.stride * a.ldx * 4, row_pad = a.pad * a.ldx * 4;
    auto row_off = [&](int m) -> int { return m < M ? m * row_step - row_pad : G::OOB_OFF; };
    int c_in_tap = 0, uoff = 0;
    const int cin = a.cin, step_tap = (a.dil * a.ldx - a.cin) * 4;
    auto step = [&](int kt) -> int {
        if (kt > 0) { c_in_tap += 32; uoff += 128; if (c_in_tap == cin) { c_in_tap = 0; uoff += step_tap; } }
        return uoff;
    };
    G::mainloop(lds, a.K / 32, row_off, step, FastALoader<1>::desc(a, xz), uniform_rsrc(wz, (int)((long)a.n * (a.K / 32) * 128)),
                a.n, a.K, m0, n0, acc);
#ifdef KN_T_NOEPI
    { float sink = 0.f;
#pragma unroll
      for (int i = 0; i < G::TM; ++i)
#pragma unroll
          for (int j = 0; j < G::TN; ++j)
#pragma unroll
              for (int r = 0; r < G::NR; ++r) sink += acc[i][j][r];
      if (sink == 123456.789f) a.out[threadIdx.x] = sink; }
#else
#ifdef KN_T_LINEPI         // timing aid: the column-per-lane epilogue instead of the LDS-transposed one
    conv_epilogue_lin<G>(a, acc, m0, n0, b, g);
#else
    __syncthreads();                                  // every wave is done with the ring: its stages become the epilogue patches
#ifdef KN_QUAD_PROF
    if (threadIdx.x == 0 && blockIdx.x < 8192) kn_quad_prof_buf[blockIdx.x * 4 + 2] = (long long)__builtin_amdgcn_s_memrealtime();
#endif
    if constexpr (EPI == 1) conv_epilogue_wide_fast<G, 0, false>(a, acc, lds, m0, n0, b, g);
    else if constexpr (EPI == 2) conv_epilogue_wide_fast<G, 1, false>(a, acc, lds, m0, n0, b, g);
    else if constexpr (EPI == 3) conv_epilogue_wide_fast<G, 0, true>(a, acc, lds, m0, n0, b, g);
    else conv_epilogue_wide<G>(a, acc, lds, m0, n0, b, g);
#ifdef KN_QUAD_PROF
    __builtin_amdgcn_s_waitcnt(0);                    // stores acknowledged
    __syncthreads();
    if (threadIdx.x == 0 && blockIdx.x < 8192) kn_quad_prof_buf[blockIdx.x * 4 + 3] = (long long)__builtin_amdgcn_s_memrealtime();
#endif
#endif
#endif
}

template <class G, int EPI = 0>
int launch2quad(const ConvArgs& a, int batches, hipStream_t st) {
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute((const void*)conv_gemm2quad_kernel<G, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                G::LDS_BYTES) != hipSuccess)
            return knnsvc_fail(KNNSVC_EHIP, "conv_gemm: hipFuncSetAttribute failed");
        attr = true;
    }
    dim3 grid((unsigned)quad_order_ids(cdiv64(a.m, G::BM), cdiv64(a.n, G::BN)), 1, (unsigned)(batches * a.groups));
    hipLaunchKernelGGL((conv_gemm2quad_kernel<G, EPI>), grid, dim3(256), G::LDS_BYTES, st, a);
    return knnsvc_check_launch("conv_gemm2quad");
}

template <class G, bool A2>
int launch2v(const ConvArgs& a, int batches, hipStream_t st) {
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute((const void*)conv_gemm2_kernel<G, A2>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                G::LDS_BYTES) != hipSuccess)
            return knnsvc_fail(KNNSVC_EHIP, "conv_gemm: hipFuncSetAttribute failed");
        attr = true;
    }
    const long gx = cdiv64(a.m, G::BM), gy = cdiv64(a.n, G::BN);
    ConvArgs ap = a;
    ap.plain = gy == 1 && gx % 8 != 0;
    const long gx8 = ap.plain ? gx : cdiv64(gx, 8) * 8;          // row tiles padded to whole groups of 8 (one per XCD)
    dim3 grid((unsigned)(gx8 * gy), 1, (unsigned)(batches * a.groups));
    hipLaunchKernelGGL((conv_gemm2_kernel<G, A2>), grid, dim3(256), G::LDS_BYTES, st, ap);
    return knnsvc_check_launch("conv_gemm2");
}
template <class G>
int launch2(const ConvArgs& a, int batches, hipStream_t st) {
    return a.x_split ? launch2v<G, true>(a, batches, st) : launch2v<G, false>(a, batches, st);
}

// one thread per 4 consecutive k of one weight row: scale * fp32 -> (hi, lo) fp16 planes, round to nearest
__global__ void split_weight2_kernel(const float* __restrict__ w, long n, int K, float scale, const float* __restrict__ absmax,
                                     unsigned short* __restrict__ out) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long groups_per_row = K / 4;
    if (absmax) scale = kn_pick_scale(kn_slot_max(absmax));
    if (i >= n * groups_per_row) return;
    const long row = i / groups_per_row; const int k = (int)(i - row * groups_per_row) * 4;
    g2_u32x2 hi, lo;
    f16x2_split4(*(const f32x4*)(w + row * K + k), scale, hi, lo);
    unsigned short* o = out + row * (long)(K / 32) * 64 + (k / 32) * 64 + (k % 32);
    *(g2_u32x2*)o = hi;
    *(g2_u32x2*)(o + 32) = lo;
}

template <class G>
int launch3(const ConvArgs& a, int batches, hipStream_t st) {
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute((const void*)conv_gemm3_kernel<G>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                G::LDS_BYTES) != hipSuccess)
            return knnsvc_fail(KNNSVC_EHIP, "conv_gemm: hipFuncSetAttribute failed");
        attr = true;
    }
    dim3 grid((unsigned)cdiv64(a.m, G::BM), (unsigned)cdiv64(a.n, G::BN), (unsigned)(batches * a.groups));
    hipLaunchKernelGGL((conv_gemm3_kernel<G>), grid, dim3(256), G::LDS_BYTES, st, a);
    return knnsvc_check_launch("conv_gemm3");
}

// one thread per 4 consecutive k of one weight row: fp32 -> three truncated bf16 planes
__global__ void split_weight_kernel(const float* __restrict__ w, long n, int K, unsigned short* __restrict__ out) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;          // index of a group of 4 floats
    const long groups_per_row = K / 4;
    if (i >= n * groups_per_row) return;
    const long row = i / groups_per_row; const int k = (int)(i - row * groups_per_row) * 4;
    const f32x4 v = *(const f32x4*)(w + row * K + k);
    unsigned short* o = out + row * (long)(K / 32) * 96 + (k / 32) * 96 + (k % 32);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const unsigned x = __float_as_uint(v[e]);
        const float r1 = v[e] - __uint_as_float(x & 0xFFFF0000u);
        const unsigned y = __float_as_uint(r1);
        const float r2 = r1 - __uint_as_float(y & 0xFFFF0000u);
        o[e] = (unsigned short)(x >> 16);
        o[32 + e] = (unsigned short)(y >> 16);
        o[64 + e] = (unsigned short)(__float_as_uint(r2) >> 16);
    }
}

template <class G, int VEC>
int launch(const ConvArgs& a, int batches, hipStream_t st) {
    dim3 grid((unsigned)cdiv64(a.m, G::BM), (unsigned)cdiv64(a.n, G::BN), (unsigned)(batches * a.groups));
    hipLaunchKernelGGL((conv_gemm_kernel<G, VEC>), grid, dim3(256), G::LDS_BYTES, st, a);
    return knnsvc_check_launch("conv_gemm");
}

using G128 = GemmTile<128, 128, 2, 2, 2, 2>;
using G64 = GemmTile<128, 64, 4, 1, 1, 2>;
using G32 = GemmTile<128, 32, 4, 1, 1, 1>;
using H128 = Gemm3Tile<128, 128, 2, 2, 2, 2>;
using H64 = Gemm3Tile<256, 64, 4, 1, 2, 2>;     // small-N layers: taller tiles so that a wave still issues 48 / 24
using H32 = Gemm3Tile<256, 32, 4, 1, 2, 1>;     // MFMAs between the two barriers of a slab
using F128 = Gemm2Tile<128, 128, 2, 2, 2, 2>;
using F64S = Gemm2Tile<64, 64, 2, 2, 1, 1>;       // launches that would put < 256 blocks of 128x128 on the chip
using F64 = Gemm2Tile<256, 64, 4, 1, 2, 2>;
using F32 = Gemm2Tile<256, 32, 4, 1, 2, 1>;
using Q256S = Gemm2QuadS;          // 256x256 block, 16x16x32 MFMA, 32-k slabs
using W128 = Gemm2Win<128, 128, 2, 2, 2, 2, 64>;      // window 192 rows (27 KB) + weights 18 KB: 3 blocks / CU
using W160 = Gemm2Win<160, 128, 1, 4, 5, 1, 64>;      // 224-row window (32 KB) + 18 KB: still 3 blocks / CU; see the dispatch rule
using W128S = Gemm2Win<64, 128, 2, 2, 1, 2, 64>;      // short time axes (first generator stage): twice the blocks, 37 KB
using W128D = Gemm2Win<64, 128, 2, 2, 1, 2, 64, 1>;   // the same tile with the weight slabs requested two steps ahead (mainloop_deep)
using W64 = Gemm2Win<256, 64, 4, 1, 2, 2, 64>;        // 320 rows (45 KB) + 9 KB
using W32 = Gemm2Win<256, 32, 4, 1, 2, 1, 64>;        // 320 rows + 4.5 KB
using W64P = Gemm2Win<256, 64, 4, 1, 2, 2, 128>;      // k = 128 positional conv: 384 rows (54 KB) + 9 KB


template <class G>
int prepare() {   // opt in to > 64 KiB of dynamic LDS once per kernel
    static bool done4 = false, done1 = false, done8 = false;
    if (!done8) {
        if (hipFuncSetAttribute((const void*)conv_gemm_kernel<G, 8>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                G::LDS_BYTES) != hipSuccess) return 1;
        done8 = true;
    }
    if (!done4) {
        if (hipFuncSetAttribute((const void*)conv_gemm_kernel<G, 4>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                G::LDS_BYTES) != hipSuccess) return 1;
        done4 = true;
    }
    if (!done1) {
        if (hipFuncSetAttribute((const void*)conv_gemm_kernel<G, 1>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                G::LDS_BYTES) != hipSuccess) return 1;
        done1 = true;
    }
    return 0;
}

// A/B switches of the dispatcher, read from the environment ONCE (first launch) into this struct; knnsvc_reload_knobs() re-reads
// them (tests that switch a route inside one process call it through ops.reload_knobs()).  Round 3 called getenv ~10 times per
// launch.  Defaults are the product configuration; none of these changes results beyond what its comment says.
struct Knobs {
    int quad = 1;                  // KNNSVC_QUAD: 0 = no 256x256 kernel, 1 = by shape (default), 2 = every qualifying launch
    bool quad_epi = true;          // KNNSVC_QUAD_EPI=0: generic epilogue behind the quad kernel (bit-identical, slower)
    bool generic_epilogue = false; // KNNSVC_EPILOGUE=g: the generic (64-bit addressed) epilogue everywhere
    bool win = true;               // KNNSVC_WIN=0: tap-major kernel instead of the windowed one (other summation order)
    bool win_wide = true;          // KNNSVC_WIN_WIDE=0: the windowed kernels keep the column-per-lane epilogue
    bool win_small = true, win_deep = true, win160 = true;      // KNNSVC_WIN_SMALL / _DEEP / WIN160 = 0: tile-shape rules off
    bool gemm_small = true;        // KNNSVC_GEMM_SMALL=0: no 64x64 tile for launches below one round of 128x128 tiles
    bool loaded = false;
};
Knobs g_knobs;
bool env_off(const char* name) { const char* e = getenv(name); return e && e[0] == '0'; }
void load_knobs() {
    Knobs k;
    const char* q = getenv("KNNSVC_QUAD");
    k.quad = q ? atoi(q) : 1;
    k.quad_epi = !env_off("KNNSVC_QUAD_EPI");
    const char* ep = getenv("KNNSVC_EPILOGUE");
    k.generic_epilogue = ep && ep[0] == 'g';
    k.win = !env_off("KNNSVC_WIN"); k.win_small = !env_off("KNNSVC_WIN_SMALL"); k.win_deep = !env_off("KNNSVC_WIN_DEEP");
    k.win_wide = !env_off("KNNSVC_WIN_WIDE");
    k.win160 = !env_off("KNNSVC_WIN160"); k.gemm_small = !env_off("KNNSVC_GEMM_SMALL");
    k.loaded = true;
    g_knobs = k;
}
const Knobs& knobs() { if (!g_knobs.loaded) load_knobs(); return g_knobs; }

}  // namespace

extern "C" int knnsvc_reload_knobs(void) { load_knobs(); return KNNSVC_OK; }

// descriptor -> kernel arguments (validation included); `fast`: the buffer-load fast path applies, `vec4`: 16-byte vectors do
static int conv_prep(const knnsvc_conv_desc* d, ConvArgs& a, bool& fast, bool& vec4, bool& quad_ok) {
    KN_REQUIRE(d && d->x && d->w && d->out, "conv_gemm: null operand");
    KN_REQUIRE(d->cin > 0 && d->taps > 0 && d->n > 0 && d->m >= 0 && d->t_in >= 0, "conv_gemm: bad sizes");
    KN_REQUIRE(d->batches > 0 && d->groups > 0, "conv_gemm: batches/groups must be positive");
    // rows may overlap (ldx < cin) only without taps: the framed-signal view of an STFT (row t = x[t*hop .. t*hop+cin))
    KN_REQUIRE((d->ldx >= d->cin || (d->taps == 1 && d->ldx > 0)) && d->stride > 0, "conv_gemm: ldx < cin or stride <= 0");
    KN_REQUIRE((long)d->batches * d->groups <= 65535, "conv_gemm: batches*groups > 65535");
    KN_REQUIRE(cdiv64(d->n, 32) <= 65535, "conv_gemm: n too large for grid.y");
    if (d->convt_u) {
        KN_REQUIRE(d->convt_cout > 0 && d->n % d->convt_cout == 0 && d->n / d->convt_cout == d->convt_u,
                   "conv_gemm: transposed mode needs n == u*cout");
        KN_REQUIRE(d->ldo >= d->convt_cout, "conv_gemm: ldo < cout");
    } else {
        KN_REQUIRE(d->ldo >= d->n, "conv_gemm: ldo < n");
    }
    if (d->resid) KN_REQUIRE(d->ldr > 0, "conv_gemm: resid without ldr");

    a.x = d->x; a.x_bstride = d->x_bstride; a.x_gstride = d->x_gstride; a.ldx = d->ldx; a.t_in = d->t_in;
    a.cin = d->cin; a.taps = d->taps; a.stride = d->stride; a.dil = d->dil; a.pad = d->pad; a.a_slope = d->a_slope;
    a.w = d->w; a.w_gstride = d->w_gstride; a.n = d->n;
    a.bias = d->bias; a.bias_gstride = d->bias_gstride; a.bias_period = d->bias_period;
    a.out = d->out; a.o_bstride = d->o_bstride; a.o_gstride = d->o_gstride; a.ldo = d->ldo; a.m = d->m;
    a.act = d->act; a.act_slope = d->act_slope;
    a.resid = d->resid; a.r_bstride = d->r_bstride; a.r_gstride = d->r_gstride; a.ldr = d->ldr;
    a.accumulate = d->accumulate; a.div = d->div == 0.f ? 1.0f : d->div;
    a.groups = d->groups;
    a.convt_u = d->convt_u; a.convt_cout = d->convt_cout; a.convt_pad = d->convt_pad; a.t_out = d->t_out;
    a.K = d->cin * d->taps;
    a.w3 = (const unsigned short*)d->w_bf16x3;
    a.w2 = (const unsigned short*)d->w_f16x2;
    a.out_scale = 1.0f; a.a_scale = 1.0f; a.w_scale = 1.0f;
    a.x_absmax = nullptr; a.w_absmax = nullptr; a.out_absmax = d->out_absmax; a.x_bound_mul = 1.0f; a.x_bound_add = 0.f;
    a.n_dyn = d->n_dyn; a.dyn_tin_mul = d->dyn_t_in_mul; a.dyn_tin_add = d->dyn_t_in_add; a.dyn_m_mul = d->dyn_m_mul;
    a.dyn_m_add = d->dyn_m_add; a.dyn_tout_mul = d->dyn_t_out_mul;
    if (d->n_dyn) KN_REQUIRE(d->dyn_t_in_mul >= 0 && d->dyn_m_mul >= 0 && d->dyn_t_in_add >= 0 && d->dyn_m_add >= 0 && d->batches == 1,
                             "conv_gemm: dynamic length needs non-negative affine coefficients and batches == 1");
    a.split_scale = d->out_f16x2_scale > 0.f ? d->out_f16x2_scale : KN_F16X2_A_SCALE;
    a.x_split = d->x_f16x2; a.out_split = d->out_f16x2 != 0; a.split_from = d->out_f16x2 > 1 ? d->out_f16x2 : 0;
    KN_REQUIRE(d->out_f16x2 >= 0 && (d->out_f16x2 <= 1 || d->out_f16x2 % 32 == 0), "conv_gemm: out_f16x2 is 0, 1 or the first split column (a multiple of 32)");
    a.wide = 0; a.plain = 0;
    a.lin = !d->convt_u && (long)d->m * d->ldo * 4 < (1L << 31) && (!d->resid || (long)d->m * d->ldr * 4 < (1L << 31)) &&
            !knobs().generic_epilogue;      // KNNSVC_EPILOGUE=g: generic epilogue (A/B)

    // 16-byte vector path needs every float4 of A and W to be aligned and inside one tap
    vec4 = (d->cin % 4 == 0) && (d->ldx % 4 == 0) && (((uintptr_t)d->x & 15) == 0) &&
                      (((uintptr_t)d->w & 15) == 0) && (d->x_bstride % 4 == 0) && (d->x_gstride % 4 == 0) &&
                      (d->w_gstride % 4 == 0);
    if (prepare<G128>() || prepare<G64>() || prepare<G32>())
        return knnsvc_fail(KNNSVC_EHIP, "conv_gemm: hipFuncSetAttribute failed");
    // buffer-load fast path: every slab inside one tap, resources below 1 GiB
    fast = vec4 && (d->cin % 32 == 0) && ((long)d->t_in * d->ldx * 4 < (1L << 30)) &&
                      ((long)d->n * a.K * 4 < (1L << 30)) && ((long)d->m * d->stride * d->ldx * 4 < (1L << 30));
    if (a.out_split)
        KN_REQUIRE(!d->resid && !d->accumulate && a.div == 1.0f && !d->convt_u && d->n % 32 == 0 && d->ldo % 32 == 0 &&
                   ((uintptr_t)d->out & 15) == 0, "conv_gemm: out_f16x2 needs a plain [m, n % 32 == 0] output (no resid/accumulate/div/convt)");
    if (a.out_split) {
        int e2 = 0;
        KN_REQUIRE(frexpf(a.split_scale, &e2) == 0.5f, "conv_gemm: out_f16x2_scale must be a power of two");
        KN_REQUIRE(!a.out_absmax, "conv_gemm: out_absmax is for fp32 outputs (a split output's scale is chosen by the caller)");
    }
    if (a.x_split)      // the scale the producer split with: a_f16x2_scale (0 = 16), or kn_pick_scale(*x_absmax)
        KN_REQUIRE(fast && a.w2 && d->a_slope == 1.0f,
                   "conv_gemm: x_f16x2 needs the f16x2 fast path (cin % 32 == 0, split weights) and a_slope 1");
    if (fast && a.w2) {        // fp32 emulated on the fp16 matrix cores (gemm2_core.h)
        KN_REQUIRE(d->w_f16x2_scale > 0.f || d->w_absmax, "conv_gemm: w_f16x2 without its scale");
        a.a_scale = d->a_f16x2_scale > 0.f ? d->a_f16x2_scale : KN_F16X2_A_SCALE;
        a.w_scale = d->w_f16x2_scale > 0.f ? d->w_f16x2_scale : 1.0f;
        { int e2 = 0; KN_REQUIRE(frexpf(a.a_scale, &e2) == 0.5f && frexpf(a.w_scale, &e2) == 0.5f, "conv_gemm: f16x2 scales must be powers of two"); }
        a.out_scale = 1.0f / (a.a_scale * a.w_scale);
        a.x_absmax = d->x_absmax; a.w_absmax = d->w_absmax;      // device-side scales override the two above
        a.x_bound_mul = d->x_bound_mul > 0.f ? d->x_bound_mul : 1.0f; a.x_bound_add = d->x_bound_add;
        KN_REQUIRE(d->x_bound_add >= 0.f && d->x_bound_mul >= 0.f, "conv_gemm: x_bound_mul / x_bound_add must be non-negative");
        quad_ok = a.x_split && a.lin && d->n % 4 == 0 && d->ldo % 4 == 0 && (!d->resid || d->ldr % 4 == 0) && ((uintptr_t)d->out & 15) == 0 &&
            (!d->resid || ((uintptr_t)d->resid & 15) == 0) && d->o_bstride % 4 == 0 && d->o_gstride % 4 == 0 && d->r_bstride % 4 == 0 &&
            d->r_gstride % 4 == 0 && (!a.out_split || a.split_from % 128 == 0) && (!d->bias || d->bias_period || ((uintptr_t)d->bias & 15) == 0) &&
            d->bias_gstride % 4 == 0;
        const bool plain_t = d->convt_u && !d->resid && d->convt_cout % 4 == 0 && (long)d->t_out * d->ldo * 4 < (1L << 31) &&
                             (long)d->m * d->convt_u < (1L << 30);
        a.wide = knobs().win_wide && (a.lin || plain_t) && !a.out_split && d->act == KNNSVC_ACT_NONE && !d->accumulate && a.div == 1.0f &&
            d->n % 4 == 0 && d->ldo % 4 == 0 && ((uintptr_t)d->out & 15) == 0 && d->o_bstride % 4 == 0 && d->o_gstride % 4 == 0 &&
            (!d->resid || (d->ldr % 4 == 0 && ((uintptr_t)d->resid & 15) == 0 && d->r_bstride % 4 == 0 && d->r_gstride % 4 == 0)) &&
            (!d->bias || (d->bias_period % 4 == 0 && ((uintptr_t)d->bias & 15) == 0 && d->bias_gstride % 4 == 0));
        KN_REQUIRE(d->fixed_tile != 2 || quad_ok, "conv_gemm: fixed_tile 2 needs the quad kernel's operand layout (split A, 16-byte rows, n % 4 == 0)");
    }
    return KNNSVC_OK;
}

// the windowed kernel's tile shape for `z` convolutions of [m, n] in one grid (z = batches * groups * descriptors)
enum WinShape { WS_NONE, WS_128D, WS_128S, WS_160, WS_128, WS_64, WS_32, WS_64P };
static WinShape win_shape(const knnsvc_conv_desc* d, const ConvArgs& a, long z) {
    if (a.x_split || d->convt_u || d->stride != 1 || d->taps < 3 || d->dil < 1 || d->ldx % 4 != 0 || !knobs().win) return WS_NONE;
    const int halo = (d->taps - 1) * d->dil;
    if (halo <= 64) {
        if (d->n > 64) {
            // fewer than ~2/3 of the chip's 768 resident slots at 128-row tiles: halve the tile height
            if (knobs().win_small && cdiv64(d->m, 128) * cdiv64(d->n, 128) * z < 512) {
                // two steps of weight prefetch where a launch leaves at most one workgroup per CU (nothing else hides the L2
                // round trip: 3750 x 256, k = 11: 70 -> 59 us); with more workgroups per CU the neighbours already do and the
                // second register set only costs (15 000 x 256, k = 7: 61 -> 66 us).  Same products, same order: same bits.
                if (knobs().win_deep && cdiv64(d->m, 64) * cdiv64(d->n, 128) * z <= 256) return WS_128D;
                return WS_128S;
            }
            // Tile-count quantisation: 768 blocks are resident at once (3 per CU); a launch of 938 128-row tiles (the
            // generator's C = 128 stage at 30 s: 120 000 rows) runs two rounds, the second 22 % full.  160-row tiles make
            // it 750 — one round.  Pick the height with the fewer (rounds x rows per round).  KNNSVC_WIN160=0: off.
            const long zz = z * cdiv64(d->n, 128);
            const long r128 = cdiv64(cdiv64(d->m, 128) * zz, 768) * 128, r160 = cdiv64(cdiv64(d->m, 160) * zz, 768) * 160;
            return (knobs().win160 && r160 < r128) ? WS_160 : WS_128;
        }
        return d->n > 32 ? WS_64 : WS_32;
    }
    if (halo <= 128 && d->n > 32 && d->n <= 64) return WS_64P;
    return WS_NONE;
}

extern "C" int knnsvc_conv_gemm(const knnsvc_conv_desc* d, void* stream) {
    ConvArgs a;
    bool fast = false, vec4 = false, quad_ok = false;
    const int rc = conv_prep(d, a, fast, vec4, quad_ok);
    if (rc) return rc;
    if (d->m == 0) return KNNSVC_OK;
    hipStream_t st = (hipStream_t)stream;
    g_last_epilogue = "";
    if (fast && a.w2) {        // fp32 emulated on the fp16 matrix cores (gemm2_core.h)
        if (d->fixed_tile != 1 && quad_ok) {
            // 256x256 block, 128x128 wave tiles, hand-pipelined loop (Gemm2QuadS: v_mfma_f32_16x16x32_f16, 32-k slabs): every A2
            // launch with K >= 1024 and n >= 256 — FFN1 / FFN2 / QKV / out-proj, the conv stack, the kNN's dot matrix.
            // KNNSVC_QUAD=0 switches it off, =2 forces it for every qualifying launch (tests, A/B runs).
            // The rule looks at the layer's shape (n, K) only, never at m or the batch: the quad kernel sums over K in another
            // grouping than the 128x128 kernels, so a rule that counted tiles (round 2) made an utterance's features depend, in
            // their last bits, on how many chunks were encoded with it — enough to flip near-tied neighbours and to make a source
            // converted in a batch differ from the same source converted alone (tests/test_gpu_product.py).
            const Knobs& kb = knobs();
            if (d->fixed_tile == 2 || kb.quad == 2 || (kb.quad == 1 && d->n >= 256 && a.K >= 1024)) {
                // specialised epilogues (conv_epilogue_wide_fast): bias (+ split), GELU (+ split), residual; KNNSVC_QUAD_EPI=0: generic
                const bool fast_epi = kb.quad_epi && !d->accumulate && a.div == 1.0f && !a.out_absmax &&
                                      (d->act == KNNSVC_ACT_NONE || d->act == KNNSVC_ACT_GELU) && !(d->resid && d->act != KNNSVC_ACT_NONE);
                g_last_kernel = "Q256S";
                if (fast_epi && d->resid) return launch2quad<Q256S, 3>(a, d->batches, st);
                if (fast_epi && d->act == KNNSVC_ACT_GELU) return launch2quad<Q256S, 2>(a, d->batches, st);
                if (fast_epi) return launch2quad<Q256S, 1>(a, d->batches, st);
                return launch2quad<Q256S>(a, d->batches, st);
            }
        }
        g_last_epilogue = a.wide ? "patch" : "lane";      // the windowed and tap-major f16x2 kernels below
        // stride-1 multi-tap convolutions on fp32 input: windowed kernel (A staged once per channel slab, not once per tap)
        switch (win_shape(d, a, (long)d->batches * d->groups)) {
            case WS_128D: g_last_kernel = "W128D"; return launch2win<W128D, 4>(a, d->batches, st);
            case WS_128S: g_last_kernel = "W128S"; return launch2win<W128S, 4>(a, d->batches, st);
            case WS_160: g_last_kernel = "W160"; return launch2win<W160, 3>(a, d->batches, st);
            case WS_128: g_last_kernel = "W128"; return launch2win<W128, 3>(a, d->batches, st);
            case WS_64: g_last_kernel = "W64"; return launch2win<W64, 2>(a, d->batches, st);
            case WS_32: g_last_kernel = "W32"; return launch2win<W32, 3>(a, d->batches, st);
            case WS_64P: g_last_kernel = "W64P"; return launch2win<W64P, 2>(a, d->batches, st);
            case WS_NONE: break;
        }
        if (d->n > 64 && cdiv64(d->m, 128) * cdiv64(d->n, 128) * d->batches * d->groups < 256) {
            if (knobs().gemm_small) { g_last_kernel = "F64S"; return launch2<F64S>(a, d->batches, st); }
        }
        g_last_kernel = d->n > 64 ? (a.x_split ? "F128a2" : "F128") : d->n > 32 ? "F64" : "F32";
        if (d->n > 64) return launch2<F128>(a, d->batches, st);
        if (d->n > 32) return launch2<F64>(a, d->batches, st);
        return launch2<F32>(a, d->batches, st);
    }
    if (fast && a.w3) {        // fp32 emulated on the bf16 matrix cores (gemm3_core.h)
        g_last_kernel = d->n > 64 ? "H128" : d->n > 32 ? "H64" : "H32";
        if (d->n > 64) return launch3<H128>(a, d->batches, st);
        if (d->n > 32) return launch3<H64>(a, d->batches, st);
        return launch3<H32>(a, d->batches, st);
    }
    g_last_kernel = d->n > 64 ? (fast ? "G128v8" : vec4 ? "G128v4" : "G128v1") : d->n > 32 ? (fast ? "G64v8" : vec4 ? "G64v4" : "G64v1") : (fast ? "G32v8" : vec4 ? "G32v4" : "G32v1");
    if (d->n > 64) return fast ? launch<G128, 8>(a, d->batches, st) : vec4 ? launch<G128, 4>(a, d->batches, st) : launch<G128, 1>(a, d->batches, st);
    if (d->n > 32) return fast ? launch<G64, 8>(a, d->batches, st) : vec4 ? launch<G64, 4>(a, d->batches, st) : launch<G64, 1>(a, d->batches, st);
    return fast ? launch<G32, 8>(a, d->batches, st) : vec4 ? launch<G32, 4>(a, d->batches, st) : launch<G32, 1>(a, d->batches, st);
}

extern "C" const char* knnsvc_conv_gemm_last_kernel(void) { return g_last_kernel; }
extern "C" const char* knnsvc_conv_gemm_last_epilogue(void) { return g_last_epilogue; }

extern "C" int knnsvc_conv_gemm_multi(const knnsvc_conv_desc* descs, int32_t count, void* stream) {
    KN_REQUIRE(descs && count >= 1 && count <= KN_MAX_MULTI, "conv_gemm_multi: 1..4 descriptors");
    ConvArgsN an;
    bool one_grid = true;
    for (int i = 0; i < count; ++i) {
        bool fast = false, vec4 = false, quad_ok = false;
        const int rc = conv_prep(&descs[i], an.b[i], fast, vec4, quad_ok);
        if (rc) return rc;
        const knnsvc_conv_desc& d = descs[i];
        // one grid needs one output shape and the windowed kernel for every descriptor; anything else: one launch each
        one_grid = one_grid && fast && an.b[i].w2 && d.batches == 1 && d.groups == 1 && d.m == descs[0].m && d.n == descs[0].n && d.m > 0 &&
                   win_shape(&d, an.b[i], count) != WS_NONE && win_shape(&d, an.b[i], count) == win_shape(&descs[0], an.b[0], count);
    }
    if (!one_grid || count == 1) {
        for (int i = 0; i < count; ++i) { const int rc = knnsvc_conv_gemm(&descs[i], stream); if (rc) return rc; }
        return KNNSVC_OK;
    }
    for (int i = count; i < KN_MAX_MULTI; ++i) an.b[i] = an.b[0];
    hipStream_t st = (hipStream_t)stream;
    switch (win_shape(&descs[0], an.b[0], count)) {
        case WS_128D: g_last_kernel = "W128Dx"; return launch2win_multi<W128D, 4>(an, count, st);
        case WS_128S: g_last_kernel = "W128Sx"; return launch2win_multi<W128S, 4>(an, count, st);
        case WS_160: g_last_kernel = "W160x"; return launch2win_multi<W160, 3>(an, count, st);
        case WS_128: g_last_kernel = "W128x"; return launch2win_multi<W128, 3>(an, count, st);
        case WS_64: g_last_kernel = "W64x"; return launch2win_multi<W64, 2>(an, count, st);
        case WS_32: g_last_kernel = "W32x"; return launch2win_multi<W32, 3>(an, count, st);
        case WS_64P: g_last_kernel = "W64Px"; return launch2win_multi<W64P, 2>(an, count, st);
        case WS_NONE: break;
    }
    return knnsvc_fail(KNNSVC_EINVAL, "conv_gemm_multi: no windowed kernel for this shape");
}

extern "C" int knnsvc_split_weight_bf16x3(const float* w, int64_t rows, int32_t K, void* out, void* stream) {
    KN_REQUIRE(w && out && rows > 0 && K > 0 && K % 32 == 0, "split_weight: K must be a positive multiple of 32");
    KN_REQUIRE(((uintptr_t)w & 15) == 0, "split_weight: 16-byte alignment");
    const long groups = rows * (K / 4);
    hipLaunchKernelGGL(split_weight_kernel, dim3((unsigned)cdiv64(groups, 256)), dim3(256), 0, (hipStream_t)stream,
                       w, (long)rows, K, (unsigned short*)out);
    return knnsvc_check_launch("split_weight");
}

extern "C" int knnsvc_split_weight_f16x2(const float* w, int64_t rows, int32_t K, float scale, void* out, void* stream) {
    KN_REQUIRE(w && out && rows > 0 && K > 0 && K % 32 == 0, "split_weight: K must be a positive multiple of 32");
    KN_REQUIRE(((uintptr_t)w & 15) == 0 && ((uintptr_t)out & 15) == 0, "split_weight: 16-byte alignment");
    int e = 0;
    KN_REQUIRE(scale > 0.f && frexpf(scale, &e) == 0.5f, "split_weight: scale must be a power of two");
    const long groups = rows * (K / 4);
    hipLaunchKernelGGL(split_weight2_kernel, dim3((unsigned)cdiv64(groups, 256)), dim3(256), 0, (hipStream_t)stream,
                       w, (long)rows, K, scale, (const float*)nullptr, (unsigned short*)out);
    return knnsvc_check_launch("split_weight2");
}

extern "C" int knnsvc_split_f16x2_dyn(const float* w, int64_t rows, int32_t K, const float* absmax, void* out, void* stream) {
    KN_REQUIRE(w && out && absmax && rows > 0 && K > 0 && K % 32 == 0, "split_f16x2_dyn: K must be a positive multiple of 32");
    KN_REQUIRE(((uintptr_t)w & 15) == 0 && ((uintptr_t)out & 15) == 0, "split_f16x2_dyn: 16-byte alignment");
    const long groups = rows * (K / 4);
    hipLaunchKernelGGL(split_weight2_kernel, dim3((unsigned)cdiv64(groups, 256)), dim3(256), 0, (hipStream_t)stream,
                       w, (long)rows, K, 1.0f, absmax, (unsigned short*)out);
    return knnsvc_check_launch("split_f16x2_dyn");
}

namespace {
// grid-stride over 4-float groups of the rows (vec: 16-byte aligned rows), scalar otherwise; one atomicMax per wave
__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ x, long rows, int cols, int ld, int vec,
                                                     float* __restrict__ slot) {
    const int c4 = vec ? cols >> 2 : 0;
    slot = kn_stripe(slot);
    unsigned m = 0;
    const long n4 = rows * c4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const long r = i / c4; const int c = (int)(i - r * c4) * 4;
        const f32x4 v = *(const f32x4*)(x + r * ld + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) { const unsigned ab = abs_bits(v[e]); m = ab > m ? ab : m; }
    }
    const int tail = cols - c4 * 4;
    if (tail) {
        for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < rows * tail; i += (long)gridDim.x * blockDim.x) {
            const long r = i / tail; const int c = c4 * 4 + (int)(i - r * tail);
            const unsigned ab = abs_bits(x[r * ld + c]); m = ab > m ? ab : m;
        }
    }
    publish_absmax(slot, m);
}
}  // namespace

extern "C" int knnsvc_absmax(const float* x, int64_t rows, int32_t cols, int32_t ld, float* slot, void* stream) {
    KN_REQUIRE(x && slot && rows >= 0 && cols > 0 && ld >= cols, "absmax: bad arguments");
    if (rows == 0) return KNNSVC_OK;
    const int vec = ((((uintptr_t)x & 15) == 0) && (ld % 4 == 0)) ? 1 : 0;
    const long work = rows * (long)(vec ? (cols + 3) / 4 : cols);
    long blocks = cdiv64(work, 256 * 8);
    blocks = blocks < 1 ? 1 : blocks > 2048 ? 2048 : blocks;
    hipLaunchKernelGGL(absmax_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, (long)rows, cols, ld, vec, slot);
    return knnsvc_check_launch("absmax");
}

namespace {
// -------------------------------------------------------------------------------------------------
// Fused ResBlock pair of the generator (hifigan/ddsp_models.py:13-44, one iteration of ResBlock1.forward):
//     t1  = lrelu(conv1_{k, dil}(lrelu(x)) + b1)          out = conv2_{k, 1}(t1) + b2 + x
// in ONE launch: a block computes BM - (k - 1) output rows for all C channels.  Phase 1 is the windowed main loop of
// conv_gemm2win_kernel on the BM rows [t0 - (k-1)/2, ...) the second convolution needs; its epilogue applies bias + leaky ReLU,
// zeroes rows outside [0, T) (conv2 pads t1 with zeros, not x) and writes the f16x2 split image of t1 — [slab][row][144 B], the
// layout the window has — into LDS over the dead window; phase 2 is the same tap loop with its A fragments read from that image
// (Gemm2Win::mainloop_resident), only the weights stream; its epilogue adds bias and the residual x and stores.  t1 never
// reaches HBM: per pair the unfused form moves 5 activation tensors (read x, write t1 | read t1, read x, write out) and exposes
// four un-overlapped bursts in two single-round launches; this moves 3 (x is read twice) and exposes two.
// Same products, same order, same roundings as the two launches (same main loop, same epilogue arithmetic, same split): the
// result is bit-identical (tests/test_gpu_kernels.py::test_resblock_pair_equals_two_launches).
struct PairArgs {
    const float* x; int ldx; int T; int C; int taps; int dil;
    const unsigned short* w1; const unsigned short* w2; const float* b1; const float* b2;
    float* out; int ldo;
    float slope; float w1_scale, w2_scale;
    const float* x_absmax; float bound_mul, bound_add; float a1_scale, a2_scale;
    float* out_absmax;
    const int* n_dyn; int dyn_mul;
};

template <class G>
__device__ __forceinline__ void conv_pair_body(const PairArgs& a, float* lds) {
    typedef __attribute__((address_space(3))) char lds_c;
    typedef __attribute__((address_space(3))) unsigned short lds_h;
    static_assert(G::NR == 16, "32x32 accumulator tiles");
    constexpr int IR = G::BM + 16;                              // image rows: the last taps of the discarded tail rows read past BM
    int T = a.T;
    if (a.n_dyn) T = *a.n_dyn * a.dyn_mul;
    float s1 = a.a1_scale, s2 = a.a2_scale;
    if (a.x_absmax) {
        const float xmax = kn_slot_max(a.x_absmax);
        s1 = kn_pick_scale(xmax);
        s2 = kn_pick_scale(fmaf(xmax, a.bound_mul, a.bound_add));
    }
    const float os1 = 1.0f / (s1 * a.w1_scale), os2 = 1.0f / (s2 * a.w2_scale);
    const int taps = a.taps, h2 = (taps - 1) / 2, pad1 = a.dil * (taps - 1) / 2, BMo = G::BM - (taps - 1);
    const int t0 = (int)blockIdx.x * BMo;
    if (t0 >= T) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int C = a.C, K = C * taps, ncs = C / 32;
    float* out_slot = a.out_absmax ? kn_stripe(a.out_absmax) : nullptr;

    // ---- phase 1: t1 rows [t0 - h2, t0 - h2 + BM)
    typename G::acc_t acc[G::TM][G::TN];
#pragma unroll
    for (int i = 0; i < G::TM; ++i)
#pragma unroll
        for (int j = 0; j < G::TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const __amdgpu_buffer_rsrc_t x_rsrc = uniform_rsrc(a.x, ((T - 1) * a.ldx + C) * 4);
    {
        Split2BLoader<G::B_P, G::B_PIECES> bl(C, K, 0, tid);
        const int w_off0 = ((t0 - h2 - pad1 + (tid >> 3)) * a.ldx + (tid & 7) * 4) * 4;
        G::mainloop(lds, ncs, taps, a.dil, w_off0, a.ldx * 4, bl, acc, x_rsrc, Split2BLoader<G::B_P, G::B_PIECES>::desc(a.w1, C, K), s1, a.slope);
    }
    // (the main loop ends behind a barrier: every wave is done with the window and the weight stage)
    lds_c* img = (lds_c*)lds;
    const int img_slab = IR * G::PITCH;
#pragma unroll
    for (int j = 0; j < G::TN; ++j) {
        const int col = G::acc_col(wave, lane, j);
        const float bv = a.b1 ? a.b1[col] : 0.f;
        lds_c* base = img + (col >> 5) * img_slab + (col & 31) * 2;
#pragma unroll
        for (int i = 0; i < G::TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = G::acc_row(wave, lane, i, r);
                const int g = t0 - h2 + row;
                float v = fmaf(acc[i][j][r], os1, bv);                  // as conv_epilogue_lin: out_scale is a power of two
                v = lrelu(v, a.slope);
                if (g < 0 || g >= T) v = 0.f;                           // conv2's own zero padding
                const float xs = v * s2;                                // as f16x2_split4
                const _Float16 h = (_Float16)xs;
                const _Float16 l = (_Float16)(xs - (float)h);
                *(lds_h*)(base + row * G::PITCH) = __builtin_bit_cast(unsigned short, h);
                *(lds_h*)(base + row * G::PITCH + 64) = __builtin_bit_cast(unsigned short, l);
            }
    }
    __syncthreads();

    // ---- phase 2: out rows [t0, t0 + BMo)
#pragma unroll
    for (int i = 0; i < G::TM; ++i)
#pragma unroll
        for (int j = 0; j < G::TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    {
        Split2BLoader<G::B_P, G::B_PIECES> bl(C, K, 0, tid);
        G::mainloop_resident(lds, ncs, taps, img_slab, ncs * img_slab, bl, acc, Split2BLoader<G::B_P, G::B_PIECES>::desc(a.w2, C, K));
    }
    const __amdgpu_buffer_rsrc_t o_rsrc = uniform_rsrc(a.out, ((T - 1) * a.ldo + C) * 4);
    const int ldo4 = a.ldo * 4, ldx4 = a.ldx * 4;
    unsigned amax = 0;
    // (column-per-lane on purpose: the LDS-patch form that pays in the single-round windowed launches — conv_epilogue_wide32 — made
    //  the merged pair launches 2-3 % SLOWER, 286 against 276 us at C = 64: their workgroups are spread over three rounds and an
    //  epilogue already runs under other workgroups' main loops; the patches only add LDS traffic there)
#pragma unroll
    for (int j = 0; j < G::TN; ++j) {
        const int col = G::acc_col(wave, lane, j);
        const float bv = a.b2 ? a.b2[col] : 0.f;
#pragma unroll
        for (int i = 0; i < G::TM; ++i) {
            float rv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = G::acc_row(wave, lane, i, r);
                rv[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(x_rsrc, row < BMo ? (t0 + row) * ldx4 + col * 4 : OOB, 0, 0));
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = G::acc_row(wave, lane, i, r);
                float v = fmaf(acc[i][j][r], os2, bv);
                v += rv[r];
                if (row < BMo && t0 + row < T) { const unsigned ab = abs_bits(v); amax = ab > amax ? ab : amax; }
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), o_rsrc, row < BMo ? (t0 + row) * ldo4 + col * 4 : OOB, 0, 0);
            }
        }
    }
    if (out_slot) publish_absmax(out_slot, amax);
}

template <class G, int MINB>
__global__ __launch_bounds__(256, MINB) void conv_pair_kernel(PairArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    conv_pair_body<G>(a, lds);
}
// the pairs of several ResBlock branches in ONE launch: blockIdx.y picks the branch (see conv_gemm2win_multi_kernel)
struct PairArgsN { PairArgs b[KN_MAX_MULTI]; };
template <class G, int MINB>
__global__ __launch_bounds__(256, MINB) void conv_pair_multi_kernel(PairArgsN args) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const PairArgs a = args.b[blockIdx.y];
    conv_pair_body<G>(a, lds);
}

template <class G, int MINB>
int launch_pair_multi(const PairArgsN& an, int count, hipStream_t st) {
    constexpr int IR = G::BM + 16;
    const int ncs = an.b[0].C / 32;
    const int lds_bytes = G::LDS_BYTES > (ncs * IR + G::BN) * G::PITCH ? G::LDS_BYTES : (ncs * IR + G::BN) * G::PITCH;
    static int attr = 0;
    if (attr < lds_bytes) {
        if (hipFuncSetAttribute((const void*)conv_pair_multi_kernel<G, MINB>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) != hipSuccess)
            return knnsvc_fail(KNNSVC_EHIP, "resblock_pair_multi: hipFuncSetAttribute failed");
        attr = lds_bytes;
    }
    int taps_max = 1;
    for (int i = 0; i < count; ++i) taps_max = an.b[i].taps > taps_max ? an.b[i].taps : taps_max;
    // a branch with fewer taps yields more rows per workgroup and needs fewer of them: its surplus workgroups leave at once (t0 >= T)
    const int BMo = G::BM - (taps_max - 1);
    hipLaunchKernelGGL((conv_pair_multi_kernel<G, MINB>), dim3((unsigned)cdiv64(an.b[0].T, BMo), (unsigned)count), dim3(256), lds_bytes, st, an);
    return knnsvc_check_launch("resblock_pair_multi");
}

template <class G, int MINB>
int launch_pair(const PairArgs& a, hipStream_t st) {
    constexpr int IR = G::BM + 16;
    const int ncs = a.C / 32;
    const int lds_bytes = G::LDS_BYTES > (ncs * IR + G::BN) * G::PITCH ? G::LDS_BYTES : (ncs * IR + G::BN) * G::PITCH;
    static int attr = 0;
    if (attr < lds_bytes) {
        if (hipFuncSetAttribute((const void*)conv_pair_kernel<G, MINB>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) != hipSuccess)
            return knnsvc_fail(KNNSVC_EHIP, "resblock_pair: hipFuncSetAttribute failed");
        attr = lds_bytes;
    }
    const int BMo = G::BM - (a.taps - 1);
    hipLaunchKernelGGL((conv_pair_kernel<G, MINB>), dim3((unsigned)cdiv64(a.T, BMo)), dim3(256), lds_bytes, st, a);
    return knnsvc_check_launch("resblock_pair");
}

using P64 = Gemm2Win<128, 64, 4, 1, 1, 2, 64>;        // C = 64: window 27.6 KB / image 41.5 KB + weights 9.2 KB: 3 blocks / CU
using P32 = Gemm2Win<256, 32, 4, 1, 2, 1, 64>;        // C = 32: window 46 KB + 4.6 KB / image 39 KB: 3 blocks / CU
using P128 = Gemm2Win<64, 128, 2, 2, 1, 2, 64>;       // C = 128: 64-row tiles (image 4 x 80 rows = 46 KB + weights 18.4 KB): 2 blocks / CU
}  // namespace

static int pair_prep(const knnsvc_pair_desc* d, PairArgs& a) {
    KN_REQUIRE(d && d->x && d->w1_f16x2 && d->w2_f16x2 && d->out, "resblock_pair: null operand");
    KN_REQUIRE(d->channels == 32 || d->channels == 64 || d->channels == 128, "resblock_pair: 32, 64 or 128 channels (other widths take the two-launch form)");
    KN_REQUIRE(d->taps >= 1 && d->taps <= 11 && (d->taps & 1) && d->dil >= 1 && d->dil * (d->taps - 1) <= 64, "resblock_pair: odd taps <= 11, dil * (taps - 1) <= 64");
    KN_REQUIRE(d->t >= 0 && d->ldx >= d->channels && d->ldo >= d->channels && d->ldx % 4 == 0 && ((uintptr_t)d->x & 15) == 0, "resblock_pair: bad layout");
    KN_REQUIRE((long)d->t * d->ldx * 4 < (1L << 30) && (long)d->t * d->ldo * 4 < (1L << 30), "resblock_pair: tensors must stay below 1 GiB");
    KN_REQUIRE(d->w1_scale > 0.f && d->w2_scale > 0.f, "resblock_pair: weight scales");
    KN_REQUIRE(d->x_absmax || (d->a1_scale > 0.f && d->a2_scale > 0.f), "resblock_pair: a range slot or both activation scales");
    if (d->n_dyn) KN_REQUIRE(d->dyn_mul > 0, "resblock_pair: dyn_mul");
    a.x = d->x; a.ldx = d->ldx; a.T = d->t; a.C = d->channels; a.taps = d->taps; a.dil = d->dil;
    a.w1 = (const unsigned short*)d->w1_f16x2; a.w2 = (const unsigned short*)d->w2_f16x2; a.b1 = d->b1; a.b2 = d->b2;
    a.out = d->out; a.ldo = d->ldo; a.slope = d->slope; a.w1_scale = d->w1_scale; a.w2_scale = d->w2_scale;
    a.x_absmax = d->x_absmax; a.bound_mul = d->t1_bound_mul > 0.f ? d->t1_bound_mul : 1.0f; a.bound_add = d->t1_bound_add;
    a.a1_scale = d->a1_scale; a.a2_scale = d->a2_scale; a.out_absmax = d->out_absmax; a.n_dyn = d->n_dyn; a.dyn_mul = d->dyn_mul;
    return KNNSVC_OK;
}

extern "C" int knnsvc_resblock_pair(const knnsvc_pair_desc* d, void* stream) {
    PairArgs a;
    const int rc = pair_prep(d, a);
    if (rc) return rc;
    if (d->t == 0) return KNNSVC_OK;
    if (d->channels == 128) return launch_pair<P128, 2>(a, (hipStream_t)stream);
    if (d->channels == 64) return launch_pair<P64, 3>(a, (hipStream_t)stream);
    return launch_pair<P32, 3>(a, (hipStream_t)stream);
}

extern "C" int knnsvc_resblock_pair_multi(const knnsvc_pair_desc* descs, int32_t count, void* stream) {
    KN_REQUIRE(descs && count >= 1 && count <= KN_MAX_MULTI, "resblock_pair_multi: 1..4 descriptors");
    PairArgsN an;
    bool one_grid = true;
    for (int i = 0; i < count; ++i) {
        const int rc = pair_prep(&descs[i], an.b[i]);
        if (rc) return rc;
        one_grid = one_grid && descs[i].channels == descs[0].channels && descs[i].t == descs[0].t && descs[i].t > 0 &&
                   descs[i].n_dyn == descs[0].n_dyn && descs[i].dyn_mul == descs[0].dyn_mul;
    }
    if (!one_grid || count == 1) {
        for (int i = 0; i < count; ++i) { const int rc = knnsvc_resblock_pair(&descs[i], stream); if (rc) return rc; }
        return KNNSVC_OK;
    }
    for (int i = count; i < KN_MAX_MULTI; ++i) an.b[i] = an.b[0];
    if (descs[0].channels == 128) return launch_pair_multi<P128, 2>(an, count, (hipStream_t)stream);
    if (descs[0].channels == 64) return launch_pair_multi<P64, 3>(an, count, (hipStream_t)stream);
    return launch_pair_multi<P32, 3>(an, count, (hipStream_t)stream);
}

namespace {
// out = (c + (b + a)) / div over the first rows * cols floats of contiguous tensors, with max |out| folded into a range slot:
// the sum of the three parallel ResBlock branches of a generator stage (hifigan/ddsp_models.py:218-227: xs = rb0 + rb1 + rb2,
// x = xs / num_kernels), in the association the serial accumulate epilogue used — the branches now run on streams of their own.
__global__ __launch_bounds__(256) void mean3_kernel(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ c,
                                                    long n4, float div, float* __restrict__ out, float* __restrict__ slot,
                                                    const int* __restrict__ n_dyn, long dyn_mul4) {
#pragma clang fp contract(off)
    if (n_dyn) { const long v = (long)n_dyn[0] * dyn_mul4; n4 = v < n4 ? v : n4; }
    unsigned m = 0;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const f32x4 va = ((const f32x4*)a)[i], vb = ((const f32x4*)b)[i], vc = ((const f32x4*)c)[i];
        f32x4 v = (vc + (vb + va)) / div;
        ((f32x4*)out)[i] = v;
#pragma unroll
        for (int e = 0; e < 4; ++e) { const unsigned ab = abs_bits(v[e]); m = ab > m ? ab : m; }
    }
    if (slot) publish_absmax(kn_stripe(slot), m);
}
}  // namespace

namespace {
// out = alpha * x (+ out): the weighted sum over WavLM layer outputs of a general layer weighting
__global__ __launch_bounds__(256) void axpy_kernel(const float* __restrict__ x, long n4, float alpha, int accumulate, float* __restrict__ out) {
#pragma clang fp contract(off)
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        f32x4 v = ((const f32x4*)x)[i] * alpha;
        if (accumulate) v = ((const f32x4*)out)[i] + v;
        ((f32x4*)out)[i] = v;
    }
}
}  // namespace

extern "C" int knnsvc_axpy(const float* x, int64_t n, float alpha, int32_t accumulate, float* out, void* stream) {
    KN_REQUIRE(x && out && n >= 0 && n % 4 == 0 && (((uintptr_t)x | (uintptr_t)out) & 15) == 0, "axpy: n % 4 == 0, 16-byte aligned");
    if (n == 0) return KNNSVC_OK;
    long blocks = cdiv64(n / 4, 256 * 4);
    blocks = blocks < 1 ? 1 : blocks > 4096 ? 4096 : blocks;
    hipLaunchKernelGGL(axpy_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, (long)(n / 4), alpha, accumulate, out);
    return knnsvc_check_launch("axpy");
}

extern "C" int knnsvc_mean3(const float* a, const float* b, const float* c, int64_t n, float div, float* out, float* out_absmax,
                            const int32_t* n_dyn, int64_t dyn_mul, void* stream) {
    KN_REQUIRE(a && b && c && out && n >= 0 && n % 4 == 0 && dyn_mul % 4 == 0 && div != 0.f, "mean3: bad arguments (n and dyn_mul are multiples of 4)");
    KN_REQUIRE((((uintptr_t)a | (uintptr_t)b | (uintptr_t)c | (uintptr_t)out) & 15) == 0, "mean3: 16-byte alignment");
    if (n == 0) return KNNSVC_OK;
    long blocks = cdiv64(n / 4, 256 * 4);
    blocks = blocks < 1 ? 1 : blocks > 4096 ? 4096 : blocks;
    hipLaunchKernelGGL(mean3_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a, b, c, (long)(n / 4), div, out,
                       out_absmax, n_dyn, (long)(dyn_mul / 4));
    return knnsvc_check_launch("mean3");
}

#ifdef KN_QUAD_PROF
extern "C" int knnsvc_debug_quad_prof(long long* host, int n_blocks) {
    if (n_blocks > 8192) n_blocks = 8192;
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(kn_quad_prof_buf), (size_t)n_blocks * 4 * sizeof(long long)) == hipSuccess ? 0 : 3;
}
#endif
