// WavLM self-attention with the gated, bucketed relative-position bias, flash style, fp32 MFMA.
// Reference: wavlm/modules.py:504-506 (compute_bias once, reused by every layer), :523-535 (gate),
// :540-563 (F.multi_head_attention_forward with the bias as a float additive mask).
//
// One block = 4 waves = 128 queries of one (batch, head); each wave owns 32 queries.
// Scores are built TRANSPOSED, S^T[key][query] = K . Q^T, so that after
// v_mfma_f32_32x32x2_f32 every lane holds 16 key scores of ONE query (its column): the online
// softmax is lane-local plus one exchange with lane^32, and exp(S^T) is already laid out as the
// B operand of the next product O^T[d][query] += V^T[d][key] . P^T[key][query] — no LDS round
// trip for P.  The bias is gate[query] * table[key - query + T - 1], read from a per-head LDS
// copy of the (2T-1)-entry table; nothing T x T ever reaches HBM.
#include "common.h"
#include "gemm2_core.h"
#include <stdlib.h>

namespace {

constexpr int HD = 64;          // head dim
constexpr int KT = 64;          // keys per LDS tile
constexpr int LDKK = 68;        // K tile pitch (floats): 16 consecutive rows hit 16 distinct 4-bank slots
constexpr int LDV = 64;

__global__ __launch_bounds__(256) void attention_kernel(const float* __restrict__ qkv, const float* __restrict__ gate,
                                                       const float* __restrict__ table, const int* __restrict__ kv_len, int T, int heads,
                                                       float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* Ks = lds;                      // [KT][LDKK]
    float* Vs = Ks + KT * LDKK;           // [KT][LDV]
    float* tb = Vs + KT * LDV;            // [2T-1]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int head = blockIdx.y, b = blockIdx.z;
    const int E = heads * HD;
    const long ld = 3L * E;
    const float* base = qkv + (long)b * T * ld;
    const int qi = blockIdx.x * 128 + wave * 32 + li;          // this lane's query
    const bool qvalid = qi < T;
    // keys this batch row may attend to: WavLM's key_padding_mask (wavlm/WavLM.py:311-321, modules.py:540-563) for a chunk that
    // sits zero-padded inside a longer bucket; rows >= Tk still get (unused, finite) outputs
    int Tk = T;
    if (kv_len) { Tk = kv_len[b]; Tk = Tk < 1 ? 1 : (Tk > T ? T : Tk); }

    for (int i = tid; i < 2 * T - 1; i += 256) tb[i] = table[(long)head * (2 * T - 1) + i];

    // Q fragment: B operand of S^T = K.Q^T; lane half h supplies d = 8g + 4h + e at step 4g + e
    f32x4 qf[8];
    const float scaling = 0.125f;      // 64^-0.5
#pragma unroll
    for (int g = 0; g < 8; ++g) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (qvalid) v = *(const f32x4*)(base + (long)qi * ld + head * HD + g * 8 + lh * 4);
        qf[g] = v * scaling;
    }
    const float g_i = qvalid ? gate[((long)b * T + qi) * heads + head] : 0.f;

    f32x16 o[2];
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
    float m_run = -__builtin_inff(), l_run = 0.f;

    // staging: thread -> rows (tid>>4) + 16*j, cols 4*(tid&15)
    const int srow = tid >> 4, scol = (tid & 15) * 4;
    f32x4 rk[4], rv[4];
    auto gload = [&](int k0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int key = k0 + srow + 16 * j;
            f32x4 kk = {0.f, 0.f, 0.f, 0.f}, vv = {0.f, 0.f, 0.f, 0.f};
            if (key < T) {
                const float* p = base + (long)key * ld + head * HD + scol;
                kk = *(const f32x4*)(p + E);
                vv = *(const f32x4*)(p + 2 * E);
            }
            rk[j] = kk; rv[j] = vv;
        }
    };
    gload(0);
    const int ntiles = (Tk + KT - 1) / KT;
    for (int t = 0; t < ntiles; ++t) {
        __syncthreads();                       // previous tile's LDS reads are done (also covers tb on t == 0)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            *(f32x4*)&Ks[(srow + 16 * j) * LDKK + scol] = rk[j];
            *(f32x4*)&Vs[(srow + 16 * j) * LDV + scol] = rv[j];
        }
        __syncthreads();
        if (t + 1 < ntiles) gload((t + 1) * KT);

#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            const int kbase = t * KT + sub * 32;           // first key of this 32-key sub-tile
            if (kbase >= Tk) break;
            // ---- S^T = K . Q^T  (A = K rows from LDS, B = Q fragment) ------------------------
            f32x16 s;
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = 0.f;
            const float* kp = &Ks[(sub * 32 + li) * LDKK + lh * 4];
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                const f32x4 ka = *(const f32x4*)(kp + g * 8);
#pragma unroll
                for (int e = 0; e < 4; ++e) s = __builtin_amdgcn_mfma_f32_32x32x2f32(ka[e], qf[g][e], s, 0, 0, 0);
            }
            // ---- bias, mask, online softmax (lane = one query, 16 keys here + 16 in lane^32) ---
            float mx = -__builtin_inff();
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kbase + (r & 3) + 8 * (r >> 2) + 4 * lh;
                float v = -__builtin_inff();
                if (key < Tk && qvalid) v = s[r] + g_i * tb[key - qi + T - 1];
                s[r] = v;
                mx = fmaxf(mx, v);
            }
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float m_new = fmaxf(m_run, mx);
            // a lane whose query is out of range keeps m_new = -inf: make its exponents finite
            const float m_use = qvalid ? m_new : 0.f;
            const float alpha = qvalid ? __expf(m_run - m_use) : 1.f;
            float ps = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = __expf(s[r] - m_use); ps += s[r]; }
            ps += __shfl_xor(ps, 32, 64);
            l_run = l_run * alpha + ps;
            m_run = m_new;
#pragma unroll
            for (int d = 0; d < 2; ++d)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[d][r] *= alpha;
            // ---- O^T += V^T . P^T  (A = V columns from LDS, B = P straight from the registers) ---
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int krow = sub * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                const float v0 = Vs[krow * LDV + li], v1 = Vs[krow * LDV + 32 + li];
                o[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(v0, s[r], o[0], 0, 0, 0);
                o[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(v1, s[r], o[1], 0, 0, 0);
            }
        }
    }
    if (qvalid) {
        const float inv = 1.0f / l_run;
        float* op = out + ((long)b * T + qi) * E + head * HD;
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                f32x4 v = {o[d][r4 * 4 + 0] * inv, o[d][r4 * 4 + 1] * inv, o[d][r4 * 4 + 2] * inv, o[d][r4 * 4 + 3] * inv};
                *(f32x4*)(op + d * 32 + r4 * 8 + lh * 4) = v;
            }
    }
}


// ---------------------------------------------------------------------------------------------
// bf16x3 variant: the same transposed flash schedule, but every fp32 product of S^T = K.Q^T and
// O^T += V^T.P^T is evaluated as six v_mfma_f32_32x32x16_bf16 on truncation-split operands
// (x = hi + mid + lo, see gemm3_core.h): fp32-level accuracy at 6/16 of the fp32-MFMA cycles.
//   K tile  : LDS [64 keys][3 planes][64 d] bf16, row pitch 400 B  (A operand of K.Q^T, ds_read_b128)
//   V tile  : LDS [64 d][3 planes][64 keys] bf16, row pitch 392 B  (transposed while staged; A operand of
//             V^T.P^T needs 8 keys per lane: two runs of 4 keys, in the order the S^T accumulator
//             registers hold them -> two ds_read_b64)
//   Q       : split once into registers (B operand of K.Q^T)
//   P       : exp() results are split in registers and used directly as the B operand of V^T.P^T
// ---------------------------------------------------------------------------------------------
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned au32x4 __attribute__((ext_vector_type(4)));
typedef unsigned au32x2 __attribute__((ext_vector_type(2)));
constexpr int KP3 = 400;      // K row pitch (bytes)
constexpr int VP3 = 392;      // V^T row pitch (bytes)

__device__ __forceinline__ void split3(float x, unsigned& h, unsigned& m, unsigned& l) {
    h = __float_as_uint(x);
    const float r1 = x - __uint_as_float(h & 0xFFFF0000u);
    m = __float_as_uint(r1);
    const float r2 = r1 - __uint_as_float(m & 0xFFFF0000u);
    l = __float_as_uint(r2);
}
__device__ __forceinline__ unsigned pack_hi(unsigned b, unsigned a) { return __builtin_amdgcn_perm(b, a, 0x07060302u); }

__global__ __launch_bounds__(256, 2) void attention3_kernel(const float* __restrict__ qkv, const float* __restrict__ gate,
                                                           const float* __restrict__ table, const int* __restrict__ kv_len, int T, int heads,
                                                           float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    typedef __attribute__((address_space(3))) char lc;
    typedef __attribute__((address_space(3))) au32x4 l_u4;
    typedef __attribute__((address_space(3))) au32x2 l_u2;
    typedef __attribute__((address_space(3))) unsigned short l_u16;
    typedef __attribute__((address_space(3))) float l_f;
    lc* Ks = (lc*)lds;                       // 64 * 400
    lc* Vs = Ks + KT * KP3;                  // 64 * 392
    l_f* tb = (l_f*)(Vs + HD * VP3);         // [2T-1]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int head = blockIdx.y, b = blockIdx.z;
    const int E = heads * HD;
    const long ld = 3L * E;
    const float* base = qkv + (long)b * T * ld;
    const int qi = blockIdx.x * 128 + wave * 32 + li;
    const bool qvalid = qi < T;
    // keys this batch row may attend to: WavLM's key_padding_mask (wavlm/WavLM.py:311-321, modules.py:540-563) for a chunk that
    // sits zero-padded inside a longer bucket; rows >= Tk still get (unused, finite) outputs
    int Tk = T;
    if (kv_len) { Tk = kv_len[b]; Tk = Tk < 1 ? 1 : (Tk > T ? T : Tk); }

    // bias table + 64 zero floats: the last key tile indexes past 2T-2 before it is masked
    for (int i = tid; i < 2 * T - 1 + 64; i += 256) tb[i] = i < 2 * T - 1 ? table[(long)head * (2 * T - 1) + i] : 0.f;

    // scores live in the log2 domain (softmax via v_exp_f32): Q carries 1/8 * log2(e)
    const float L2E = 1.44269504088896341f;
    const float qscale = 0.125f * L2E;
    // Q (scaled) split into three planes: qf[s][p] = 8 bf16 of d = 16 s + 8 h + 0..7
    au32x4 qf[4][3];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = v0;
        if (qvalid) {
            const float* p = base + (long)qi * ld + head * HD + s * 16 + lh * 8;
            v0 = *(const f32x4*)p; v1 = *(const f32x4*)(p + 4);
        }
        unsigned h[8], m[8], l[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) { split3(v0[e] * qscale, h[e], m[e], l[e]); split3(v1[e] * qscale, h[4 + e], m[4 + e], l[4 + e]); }
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            qf[s][0][w] = pack_hi(h[2 * w + 1], h[2 * w]);
            qf[s][1][w] = pack_hi(m[2 * w + 1], m[2 * w]);
            qf[s][2][w] = pack_hi(l[2 * w + 1], l[2 * w]);
        }
    }
    const float g_i = qvalid ? gate[((long)b * T + qi) * heads + head] * L2E : 0.f;
    const l_f* tbq = tb + (T - 1 - (qvalid ? qi : T - 1)) + 4 * lh;     // tbq[key] = table[key - qi + T - 1]

    f32x16 o[2];
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
    float m_run = -__builtin_inff(), l_run = 0.f;

    const int srow = tid >> 4, scol = (tid & 15) * 4;       // staging: rows (keys) srow + 16 j, d = scol..scol+3
    f32x4 rk[4], rv[4];
    auto gload = [&](int k0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int key = k0 + srow + 16 * j;
            f32x4 kk = {0.f, 0.f, 0.f, 0.f}, vv = {0.f, 0.f, 0.f, 0.f};
            if (key < T) {
                const float* p = base + (long)key * ld + head * HD + scol;
                kk = *(const f32x4*)(p + E);
                vv = *(const f32x4*)(p + 2 * E);
            }
            rk[j] = kk; rv[j] = vv;
        }
    };
    gload(0);
    const int ntiles = (Tk + KT - 1) / KT;
    for (int t = 0; t < ntiles; ++t) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int key = srow + 16 * j;
            unsigned h[4], m[4], l[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) split3(rk[j][e], h[e], m[e], l[e]);
            lc* kd = Ks + key * KP3 + scol * 2;
            *(l_u2*)(kd) = (au32x2){pack_hi(h[1], h[0]), pack_hi(h[3], h[2])};
            *(l_u2*)(kd + 128) = (au32x2){pack_hi(m[1], m[0]), pack_hi(m[3], m[2])};
            *(l_u2*)(kd + 256) = (au32x2){pack_hi(l[1], l[0]), pack_hi(l[3], l[2])};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                split3(rv[j][e], h[e], m[e], l[e]);
                lc* vd = Vs + (scol + e) * VP3 + key * 2;              // transposed: row = d, column = key
                *(l_u16*)(vd) = (unsigned short)(h[e] >> 16);
                *(l_u16*)(vd + 128) = (unsigned short)(m[e] >> 16);
                *(l_u16*)(vd + 256) = (unsigned short)(l[e] >> 16);
            }
        }
        __syncthreads();
        if (t + 1 < ntiles) gload((t + 1) * KT);

#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            const int kbase = t * KT + sub * 32;
            if (kbase >= Tk) break;
            // ---- S^T = K . Q^T : 4 d-steps x 6 products -----------------------------------------------
            f32x16 s;
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = 0.f;
            const lc* kp = Ks + (sub * 32 + li) * KP3 + lh * 16;
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                const bf16x8_t k0 = __builtin_bit_cast(bf16x8_t, *(const l_u4*)(kp + st * 32));
                const bf16x8_t k1 = __builtin_bit_cast(bf16x8_t, *(const l_u4*)(kp + 128 + st * 32));
                const bf16x8_t k2 = __builtin_bit_cast(bf16x8_t, *(const l_u4*)(kp + 256 + st * 32));
                const bf16x8_t q0 = __builtin_bit_cast(bf16x8_t, qf[st][0]), q1 = __builtin_bit_cast(bf16x8_t, qf[st][1]),
                               q2 = __builtin_bit_cast(bf16x8_t, qf[st][2]);
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k1, q1, s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k2, q0, s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k0, q2, s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k1, q0, s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k0, q1, s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k0, q0, s, 0, 0, 0);
            }
            // ---- bias, mask, online softmax --------------------------------------------------------------
            // rows past T hold zero Q (and zero gate): their scores are finite, nothing is written for them
            const l_f* tp = tbq + kbase;
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = fmaf(g_i, tp[(r & 3) + 8 * (r >> 2)], s[r]);
            if (kbase + 32 > Tk) {                                    // wave-uniform: only the last key tile masks
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (kbase + (r & 3) + 8 * (r >> 2) + 4 * lh >= Tk) s[r] = -__builtin_inff();
            }
            float mx = fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3]));
#pragma unroll
            for (int r = 4; r < 16; r += 4) mx = fmaxf(mx, fmaxf(fmaxf(s[r], s[r + 1]), fmaxf(s[r + 2], s[r + 3])));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float m_new = fmaxf(m_run, mx);                    // key 0 is always valid: finite from the first tile on
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            float ps = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = __builtin_amdgcn_exp2f(s[r] - m_new); ps += s[r]; }
            ps += __shfl_xor(ps, 32, 64);
            l_run = l_run * alpha + ps;
            m_run = m_new;
            if (__builtin_amdgcn_ballot_w64(alpha != 1.f)) {
#pragma unroll
                for (int d = 0; d < 2; ++d)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[d][r] *= alpha;
            }
            // ---- O^T += V^T . P^T : 2 key-steps x 2 d-tiles x 6 products ---------------------------------------
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                unsigned h[8], m[8], l[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) split3(s[8 * st + e], h[e], m[e], l[e]);
                au32x4 p0, p1, p2;
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    p0[w] = pack_hi(h[2 * w + 1], h[2 * w]); p1[w] = pack_hi(m[2 * w + 1], m[2 * w]); p2[w] = pack_hi(l[2 * w + 1], l[2 * w]);
                }
                const bf16x8_t b0 = __builtin_bit_cast(bf16x8_t, p0), b1 = __builtin_bit_cast(bf16x8_t, p1), b2 = __builtin_bit_cast(bf16x8_t, p2);
                // element e of this lane's fragment is key sub*32 + 16 st + 8 (e>>2) + 4 h + (e&3): two runs of four keys
                const int kcol = (sub * 32 + 16 * st + 4 * lh) * 2;
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    const lc* vp = Vs + (dt * 32 + li) * VP3 + kcol;
                    au32x4 a0, a1, a2;
                    const au32x2 x0 = *(const l_u2*)(vp), x1 = *(const l_u2*)(vp + 16);
                    const au32x2 y0 = *(const l_u2*)(vp + 128), y1 = *(const l_u2*)(vp + 128 + 16);
                    const au32x2 z0 = *(const l_u2*)(vp + 256), z1 = *(const l_u2*)(vp + 256 + 16);
                    a0 = (au32x4){x0[0], x0[1], x1[0], x1[1]};
                    a1 = (au32x4){y0[0], y0[1], y1[0], y1[1]};
                    a2 = (au32x4){z0[0], z0[1], z1[0], z1[1]};
                    const bf16x8_t v0 = __builtin_bit_cast(bf16x8_t, a0), v1 = __builtin_bit_cast(bf16x8_t, a1), v2 = __builtin_bit_cast(bf16x8_t, a2);
                    f32x16 c = o[dt];
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v1, b1, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v2, b0, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v0, b2, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v1, b0, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v0, b1, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v0, b0, c, 0, 0, 0);
                    o[dt] = c;
                }
            }
        }
    }
    if (qvalid) {
        const float inv = 1.0f / l_run;
        float* op = out + ((long)b * T + qi) * E + head * HD;
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                f32x4 v = {o[d][r4 * 4 + 0] * inv, o[d][r4 * 4 + 1] * inv, o[d][r4 * 4 + 2] * inv, o[d][r4 * 4 + 3] * inv};
                *(f32x4*)(op + d * 32 + r4 * 8 + lh * 4) = v;
            }
    }
}

// ---------------------------------------------------------------------------------------------
// f16x2 variant (default): the schedule of attention3_kernel with every fp32 product evaluated as
// three v_mfma_f32_32x32x16_f16 on (hi, lo) fp16 splits of power-of-two-scaled operands
// (gemm2_core.h): half the matrix-core work of bf16x3 at the same accuracy class.
//   K tile : LDS [64 keys][2 planes][64 d] fp16, row pitch 272 B;  V tile : LDS [64 d][2 planes][64 keys], pitch 264 B
//   scales : Q' = 16 (log2 e / 8) Q, K' = 16 K  (scores x 256, undone inside the exp2 argument);
//            P' = 2^14 P (folded into the exp2 argument, cancels in O / l);  V' = 16 V (undone at the end)
// ---------------------------------------------------------------------------------------------
typedef short short4v __attribute__((ext_vector_type(4)));
constexpr int KP2 = 272;
constexpr int VP2 = 320;          // V image: row = key, [hi 64 d | lo 64 d] + pad; 80 dwords = 16 (mod 64): the four rows of a
                                  // ds_read_b64_tr_b16 block cover the 64 banks exactly once

__global__ __launch_bounds__(256, 3) void attention2_kernel(const float* __restrict__ qkv, const float* __restrict__ gate,
                                                           const float* __restrict__ table, const int* __restrict__ kv_len, int T, int heads,
                                                           float* __restrict__ out, int out_split, int kv_split) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    typedef __attribute__((address_space(3))) char lc;
    typedef __attribute__((address_space(3))) au32x4 l_u4;
    typedef __attribute__((address_space(3))) au32x2 l_u2;
    typedef __attribute__((address_space(3))) float l_f;
    lc* Ks = (lc*)lds;                       // 64 keys * 272 B
    lc* Vs = Ks + KT * KP2;                  // 64 keys * 320 B, row-major like K (read transposed, see the PV product)
    l_f* tb = (l_f*)(Vs + KT * VP2);         // [2T-1+64]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int head = blockIdx.y, b = blockIdx.z;
    const int E = heads * HD;
    const long ld = 3L * E;
    const float* base = qkv + (long)b * T * ld;
    const int qi = blockIdx.x * 128 + wave * 32 + li;
    const bool qvalid = qi < T;
    // keys this batch row may attend to: WavLM's key_padding_mask (wavlm/WavLM.py:311-321, modules.py:540-563) for a chunk that
    // sits zero-padded inside a longer bucket; rows >= Tk still get (unused, finite) outputs
    int Tk = T;
    if (kv_len) { Tk = kv_len[b]; Tk = Tk < 1 ? 1 : (Tk > T ? T : Tk); }

    // bias table + 64 zero floats: the last key tile indexes past 2T-2 before it is masked
    for (int i = tid; i < 2 * T - 1 + 64; i += 256) tb[i] = i < 2 * T - 1 ? table[(long)head * (2 * T - 1) + i] : 0.f;

    // scores live in the log2 domain (softmax via v_exp_f32) and carry the operand scales: Q' = 16 log2(e)/8 Q,
    // K' = 16 K  ->  accumulator = 256 * log2-score; the bias enters as 256 log2(e) g b, exp2 undoes the 256.
    const float L2E = 1.44269504088896341f;
    const float qscale = 0.125f * L2E * 16.0f;
    // Q (scaled) split into two fp16 planes: qf[s][p] = 8 halves of d = 16 s + 8 h + 0..7
    au32x4 qf[4][2];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = v0;
        if (qvalid) {
            const float* p = base + (long)qi * ld + head * HD + s * 16 + lh * 8;
            v0 = *(const f32x4*)p; v1 = *(const f32x4*)(p + 4);
        }
        g2_u32x2 h0, l0, h1, l1;
        f16x2_split4(v0, qscale, h0, l0); f16x2_split4(v1, qscale, h1, l1);
        qf[s][0] = (au32x4){h0[0], h0[1], h1[0], h1[1]};
        qf[s][1] = (au32x4){l0[0], l0[1], l1[0], l1[1]};
    }
    const float g_i = qvalid ? gate[((long)b * T + qi) * heads + head] * (L2E * 256.0f) : 0.f;
    const l_f* tbq = tb + (T - 1 - (qvalid ? qi : T - 1)) + 4 * lh;     // tbq[key] = table[key - qi + T - 1]
    // transposed-read address of this lane inside a 16-key x 32-d block of V: key 4 (lane >> 5) + ((lane & 15) >> 2),
    // columns 16 ((lane >> 4) & 1) + 4 (lane & 3)
    const lc* v_tr = Vs + (4 * (lane >> 5) + ((lane & 15) >> 2)) * VP2 + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;


    f32x16 o[2];
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
    float m_run = -__builtin_inff(), l_run = 0.f;

    const int srow = tid >> 4, scol = (tid & 15) * 4;       // staging: rows (keys) srow + 16 j, d = scol..scol+3
    f32x4 rk[4], rv[4];
    auto gload = [&](int k0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int key = k0 + srow + 16 * j;
            f32x4 kk = {0.f, 0.f, 0.f, 0.f}, vv = {0.f, 0.f, 0.f, 0.f};
            if (key < T) {
                const float* p = base + (long)key * ld + head * HD + scol;
                kk = *(const f32x4*)(p + E);
                vv = *(const f32x4*)(p + 2 * E);
            }
            rk[j] = kk; rv[j] = vv;
        }
    };
    gload(0);
    const int ntiles = (Tk + KT - 1) / KT;
    for (int t = 0; t < ntiles; ++t) {
#ifdef KN_ATT_NOSTAGE      // timing aid: the first tile's K / V serve every tile, no barriers after it
        if (t == 0) {
#endif
        __syncthreads();
        if (kv_split) {
            // K and V columns of `qkv` already hold the f16x2 split layout (written by the QKV GEMM's epilogue, scale 16):
            // the 16 bytes a thread loaded are piece pc = tid & 15 of its key's 256-byte head slice — 8 halves of plane
            // (pc >> 2) & 1, channels 32 (pc >> 3) + 8 (pc & 3) .. + 7.  Staging is a copy: every query block of a head
            // (12 at T = 1500) used to redo the same split of the same keys (~110 VALU per thread and tile).
            const int pc = tid & 15, plane = (pc >> 2) & 1, d0 = (pc >> 3) * 32 + (pc & 3) * 8;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int key = srow + 16 * j;
                *(l_u4*)(Ks + key * KP2 + plane * 128 + d0 * 2) = __builtin_bit_cast(au32x4, rk[j]);
                *(l_u4*)(Vs + key * VP2 + plane * 128 + d0 * 2) = __builtin_bit_cast(au32x4, rv[j]);
            }
        } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int key = srow + 16 * j;
            g2_u32x2 hi, lo;
            f16x2_split4(rk[j], 16.0f, hi, lo);
            lc* kd = Ks + key * KP2 + scol * 2;
            *(l_u2*)(kd) = hi;
            *(l_u2*)(kd + 128) = lo;
            f16x2_split4(rv[j], 16.0f, hi, lo);
            lc* vd = Vs + key * VP2 + scol * 2;
            *(l_u2*)(vd) = hi;
            *(l_u2*)(vd + 128) = lo;
        }
        }
        __syncthreads();
#ifdef KN_ATT_NOSTAGE
        }
#else
        if (t + 1 < ntiles) gload((t + 1) * KT);
#endif

#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            const int kbase = t * KT + sub * 32;
            if (kbase >= Tk) break;
            // ---- S^T = K . Q^T : 4 d-steps x 6 products -----------------------------------------------
            f32x16 s;
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = 0.f;
            const lc* kp = Ks + (sub * 32 + li) * KP2 + lh * 16;
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                const f16x8 k0 = __builtin_bit_cast(f16x8, *(const l_u4*)(kp + st * 32));
                const f16x8 k1 = __builtin_bit_cast(f16x8, *(const l_u4*)(kp + 128 + st * 32));
                const f16x8 q0 = __builtin_bit_cast(f16x8, qf[st][0]), q1 = __builtin_bit_cast(f16x8, qf[st][1]);
#ifdef KN_ATT_NOS          // timing aid: one product instead of twelve
                if (st == 0) s = __builtin_amdgcn_mfma_f32_32x32x16_f16(k1, q0, s, 0, 0, 0);
                else { s[st] += __builtin_bit_cast(float, __builtin_bit_cast(au32x4, k0)[0] ^ __builtin_bit_cast(au32x4, k1)[1] ^ __builtin_bit_cast(au32x4, q1)[0]); }
#else
                s = __builtin_amdgcn_mfma_f32_32x32x16_f16(k1, q0, s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_32x32x16_f16(k0, q1, s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_32x32x16_f16(k0, q0, s, 0, 0, 0);
#endif
            }
            // rows past T hold zero Q (and zero gate): their scores are finite, nothing is written for them
            const l_f* tp = tbq + kbase;
#ifndef KN_ATT_NOBIAS
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = fmaf(g_i, tp[(r & 3) + 8 * (r >> 2)], s[r]);
#endif
            if (kbase + 32 > Tk) {                                    // wave-uniform: only the last key tile masks
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (kbase + (r & 3) + 8 * (r >> 2) + 4 * lh >= Tk) s[r] = -__builtin_inff();
            }
#ifdef KN_ATT_NOEXP       // timing aid: no max / exp / sums
            float mx = s[0];
#else
            float mx = fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3]));
#pragma unroll
            for (int r = 4; r < 16; r += 4) mx = fmaxf(mx, fmaxf(fmaxf(s[r], s[r + 1]), fmaxf(s[r + 2], s[r + 3])));
#endif
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float m_new = fmaxf(m_run, mx);                    // key 0 is always valid: finite from the first tile on
            const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * (1.0f / 256.0f));
            const float mneg = fmaf(m_new, -1.0f / 256.0f, 14.0f);       // P carries 2^14 (cancels in O / l)
            float ps = 0.f;
#pragma unroll
#ifdef KN_ATT_NOEXP
            for (int r = 0; r < 16; ++r) { s[r] = s[r] * 1e-9f; } ps = s[3] + mneg;
#else
            for (int r = 0; r < 16; ++r) { s[r] = __builtin_amdgcn_exp2f(fmaf(s[r], 1.0f / 256.0f, mneg)); ps += s[r]; }
#endif
            ps += __shfl_xor(ps, 32, 64);
            l_run = l_run * alpha + ps;
            m_run = m_new;
            if (__builtin_amdgcn_ballot_w64(alpha != 1.f)) {
#pragma unroll
                for (int d = 0; d < 2; ++d)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[d][r] *= alpha;
            }
            // ---- O^T += V^T . P^T : 2 key-steps x 2 d-tiles x 6 products ---------------------------------------
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                g2_u32x2 h0, l0, h1, l1;
                f16x2_split4((f32x4){s[8 * st + 0], s[8 * st + 1], s[8 * st + 2], s[8 * st + 3]}, 1.0f, h0, l0);
                f16x2_split4((f32x4){s[8 * st + 4], s[8 * st + 5], s[8 * st + 6], s[8 * st + 7]}, 1.0f, h1, l1);
                const f16x8 b0 = __builtin_bit_cast(f16x8, (au32x4){h0[0], h0[1], h1[0], h1[1]});
                const f16x8 b1 = __builtin_bit_cast(f16x8, (au32x4){l0[0], l0[1], l1[0], l1[1]});
                // element e of this lane's fragment is key sub*32 + 16 st + 8 (e>>2) + 4 h + (e&3): two runs of four keys.
                // V sits row-major in LDS (row = key) and is read TRANSPOSED by the hardware (ds_read_b64_tr_b16): each
                // 16-lane group fetches a block of 4 keys x 16 d and lane i of the group receives d = base + i of the four
                // keys; lane 4 q + p supplies the address of key q, columns 4 p .. 4 p + 3 (tools/probe/tr_read_probe.hip).
                // The transposed 2-byte stores this replaces put 64 lanes on ~4 banks: 48 % of the kernel's LDS cycles
                // were bank conflicts.  EXEC is full here (no lane-divergent control flow around the reads).
                const lc* vrow = v_tr + (sub * 32 + 16 * st) * VP2;
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    typedef __attribute__((address_space(3))) short4v l_s4;
                    const lc* vp = vrow + dt * 64;
                    const au32x2 x0 = __builtin_bit_cast(au32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((l_s4*)(vp)));
                    const au32x2 x1 = __builtin_bit_cast(au32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((l_s4*)(vp + 8 * VP2)));
                    const au32x2 y0 = __builtin_bit_cast(au32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((l_s4*)(vp + 128)));
                    const au32x2 y1 = __builtin_bit_cast(au32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((l_s4*)(vp + 8 * VP2 + 128)));
                    const f16x8 v0 = __builtin_bit_cast(f16x8, (au32x4){x0[0], x0[1], x1[0], x1[1]});
                    const f16x8 v1 = __builtin_bit_cast(f16x8, (au32x4){y0[0], y0[1], y1[0], y1[1]});
                    f32x16 c = o[dt];
#ifdef KN_ATT_NOPV         // timing aid: one product per step instead of six
                    if (dt == 0) c = __builtin_amdgcn_mfma_f32_32x32x16_f16(v1, b0, c, 0, 0, 0);
                    else c[st] += __builtin_bit_cast(float, __builtin_bit_cast(au32x4, v0)[0] ^ __builtin_bit_cast(au32x4, v1)[1] ^ __builtin_bit_cast(au32x4, b1)[0]);
#else
                    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(v1, b0, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(v0, b1, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(v0, b0, c, 0, 0, 0);
#endif
                    o[dt] = c;
                }
            }
        }
    }
    if (qvalid) {
        const float inv = 0.0625f / l_run;          // V carried a factor 16
        float* op = out + ((long)b * T + qi) * E + head * HD;
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                f32x4 v = {o[d][r4 * 4 + 0] * inv, o[d][r4 * 4 + 1] * inv, o[d][r4 * 4 + 2] * inv, o[d][r4 * 4 + 3] * inv};
                if (out_split) {          // f16x2 split layout for the output projection (4 consecutive channels of the row)
                    g2_u32x2 hi, lo;
                    f16x2_split4(v, KN_F16X2_A_SCALE, hi, lo);
                    const int c = head * HD + d * 32 + r4 * 8 + lh * 4;
                    char* ob = (char*)(out + ((long)b * T + qi) * E) + (c >> 5) * 128 + (c & 31) * 2;
                    *(g2_u32x2*)ob = hi;
                    *(g2_u32x2*)(ob + 64) = lo;
                } else
                    *(f32x4*)(op + d * 32 + r4 * 8 + lh * 4) = v;
            }
    }
}

// ---------------------------------------------------------------------------------------------
// attention2_kernel with QB blocks of 32 queries per wave (QB = 2: 64 queries per wave, 256 per workgroup).  Every K
// fragment and every transposed V fragment read from LDS feeds QB independent products, and a staged K / V tile (and its
// two barriers) serves twice the queries: half the LDS reads and half the staging per unit of work, two independent
// S / softmax / PV chains per wave to interleave, and 2016 workgroups on 512 resident slots (3.94 rounds; the 128-query
// kernel: 4032 on 768 = 5.25 -> the sixth round a quarter full).  Costs a wave per SIMD (2 instead of 3: ~230 VGPRs).
// Per query the operations and their order are those of attention2_kernel: results are bit-identical.
// ---------------------------------------------------------------------------------------------
template <int QB>
__global__ __launch_bounds__(256, 2) void attention2q_kernel(const float* __restrict__ qkv, const float* __restrict__ gate,
                                                            const float* __restrict__ table, const int* __restrict__ kv_len, int T, int heads,
                                                            float* __restrict__ out, int out_split, int kv_split) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    typedef __attribute__((address_space(3))) char lc;
    typedef __attribute__((address_space(3))) au32x4 l_u4;
    typedef __attribute__((address_space(3))) au32x2 l_u2;
    typedef __attribute__((address_space(3))) float l_f;
    lc* Ks = (lc*)lds;
    lc* Vs = Ks + KT * KP2;
    l_f* tb = (l_f*)(Vs + KT * VP2);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int head = blockIdx.y, b = blockIdx.z;
    const int E = heads * HD;
    const long ld = 3L * E;
    const float* base = qkv + (long)b * T * ld;
    int Tk = T;
    if (kv_len) { Tk = kv_len[b]; Tk = Tk < 1 ? 1 : (Tk > T ? T : Tk); }

    for (int i = tid; i < 2 * T - 1 + 64; i += 256) tb[i] = i < 2 * T - 1 ? table[(long)head * (2 * T - 1) + i] : 0.f;

    const float L2E = 1.44269504088896341f;
    const float qscale = 0.125f * L2E * 16.0f;
    int qi[QB]; bool qvalid[QB];
    au32x4 qf[QB][4][2];
    float g_i[QB];
    const l_f* tbq[QB];
#pragma unroll
    for (int u = 0; u < QB; ++u) {
        qi[u] = blockIdx.x * (128 * QB) + wave * (32 * QB) + u * 32 + li;
        qvalid[u] = qi[u] < T;
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = v0;
            if (qvalid[u]) {
                const float* p = base + (long)qi[u] * ld + head * HD + s4 * 16 + lh * 8;
                v0 = *(const f32x4*)p; v1 = *(const f32x4*)(p + 4);
            }
            g2_u32x2 h0, l0, h1, l1;
            f16x2_split4(v0, qscale, h0, l0); f16x2_split4(v1, qscale, h1, l1);
            qf[u][s4][0] = (au32x4){h0[0], h0[1], h1[0], h1[1]};
            qf[u][s4][1] = (au32x4){l0[0], l0[1], l1[0], l1[1]};
        }
        g_i[u] = qvalid[u] ? gate[((long)b * T + qi[u]) * heads + head] * (L2E * 256.0f) : 0.f;
        tbq[u] = tb + (T - 1 - (qvalid[u] ? qi[u] : T - 1)) + 4 * lh;
    }
    const lc* v_tr = Vs + (4 * (lane >> 5) + ((lane & 15) >> 2)) * VP2 + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;

    f32x16 o[QB][2];
    float m_run[QB], l_run[QB];
#pragma unroll
    for (int u = 0; u < QB; ++u) {
        m_run[u] = -__builtin_inff(); l_run[u] = 0.f;
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[u][d][r] = 0.f;
    }

    const int srow = tid >> 4, scol = (tid & 15) * 4;
    f32x4 rk[4], rv[4];
    auto gload = [&](int k0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int key = k0 + srow + 16 * j;
            f32x4 kk = {0.f, 0.f, 0.f, 0.f}, vv = {0.f, 0.f, 0.f, 0.f};
            if (key < T) {
                const float* p = base + (long)key * ld + head * HD + scol;
                kk = *(const f32x4*)(p + E);
                vv = *(const f32x4*)(p + 2 * E);
            }
            rk[j] = kk; rv[j] = vv;
        }
    };
    gload(0);
    const int ntiles = (Tk + KT - 1) / KT;
#ifdef KN_ATT_PROF          // timing aid (tools/attn_whatif.sh): where one wave's cycles go, printed by a few workgroups
    unsigned long long pf[4] = {0, 0, 0, 0}, tq = __builtin_readcyclecounter();
    const unsigned long long t_begin = tq;
#define KN_ATICK(K) { const unsigned long long now = __builtin_readcyclecounter(); pf[K] += now - tq; tq = now; }
#else
#define KN_ATICK(K)
#endif
    for (int t = 0; t < ntiles; ++t) {
        __syncthreads();
        if (kv_split) {
            const int pc = tid & 15, plane = (pc >> 2) & 1, d0 = (pc >> 3) * 32 + (pc & 3) * 8;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int key = srow + 16 * j;
                *(l_u4*)(Ks + key * KP2 + plane * 128 + d0 * 2) = __builtin_bit_cast(au32x4, rk[j]);
                *(l_u4*)(Vs + key * VP2 + plane * 128 + d0 * 2) = __builtin_bit_cast(au32x4, rv[j]);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int key = srow + 16 * j;
                g2_u32x2 hi, lo;
                f16x2_split4(rk[j], 16.0f, hi, lo);
                lc* kd = Ks + key * KP2 + scol * 2;
                *(l_u2*)(kd) = hi;
                *(l_u2*)(kd + 128) = lo;
                f16x2_split4(rv[j], 16.0f, hi, lo);
                lc* vd = Vs + key * VP2 + scol * 2;
                *(l_u2*)(vd) = hi;
                *(l_u2*)(vd + 128) = lo;
            }
        }
        __syncthreads();
        if (t + 1 < ntiles) gload((t + 1) * KT);
        KN_ATICK(0)

#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            const int kbase = t * KT + sub * 32;
            if (kbase >= Tk) break;

            // ---- S^T = K . Q^T for the QB query blocks: every K fragment pair is read once --------------
            f32x16 s[QB];
#pragma unroll
            for (int u = 0; u < QB; ++u)
#pragma unroll
                for (int r = 0; r < 16; ++r) s[u][r] = 0.f;
            const lc* kp = Ks + (sub * 32 + li) * KP2 + lh * 16;
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                const f16x8 k0 = __builtin_bit_cast(f16x8, *(const l_u4*)(kp + st * 32));
                const f16x8 k1 = __builtin_bit_cast(f16x8, *(const l_u4*)(kp + 128 + st * 32));
#pragma unroll
                for (int u = 0; u < QB; ++u) {
                    const f16x8 q0 = __builtin_bit_cast(f16x8, qf[u][st][0]), q1 = __builtin_bit_cast(f16x8, qf[u][st][1]);
                    s[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(k1, q0, s[u], 0, 0, 0);
                    s[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(k0, q1, s[u], 0, 0, 0);
                    s[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(k0, q0, s[u], 0, 0, 0);
                }
            }
            KN_ATICK(1)
            // ---- online softmax per query block (lane-local + one exchange with lane ^ 32) ----------------
#pragma unroll
            for (int u = 0; u < QB; ++u) {
                const l_f* tp = tbq[u] + kbase;
#pragma unroll
                for (int r = 0; r < 16; ++r) s[u][r] = fmaf(g_i[u], tp[(r & 3) + 8 * (r >> 2)], s[u][r]);
                if (kbase + 32 > Tk) {
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (kbase + (r & 3) + 8 * (r >> 2) + 4 * lh >= Tk) s[u][r] = -__builtin_inff();
                }
                float mx = fmaxf(fmaxf(s[u][0], s[u][1]), fmaxf(s[u][2], s[u][3]));
#pragma unroll
                for (int r = 4; r < 16; r += 4) mx = fmaxf(mx, fmaxf(fmaxf(s[u][r], s[u][r + 1]), fmaxf(s[u][r + 2], s[u][r + 3])));
                mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
                const float m_new = fmaxf(m_run[u], mx);
                const float alpha = __builtin_amdgcn_exp2f((m_run[u] - m_new) * (1.0f / 256.0f));
                const float mneg = fmaf(m_new, -1.0f / 256.0f, 14.0f);
                float ps = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) { s[u][r] = __builtin_amdgcn_exp2f(fmaf(s[u][r], 1.0f / 256.0f, mneg)); ps += s[u][r]; }
                ps += __shfl_xor(ps, 32, 64);
                l_run[u] = l_run[u] * alpha + ps;
                m_run[u] = m_new;
                if (__builtin_amdgcn_ballot_w64(alpha != 1.f)) {
#pragma unroll
                    for (int d = 0; d < 2; ++d)
#pragma unroll
                        for (int r = 0; r < 16; ++r) o[u][d][r] *= alpha;
                }
            }
            KN_ATICK(2)
            // ---- O^T += V^T . P^T: every transposed V fragment is read once -------------------------------
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                f16x8 b0[QB], b1[QB];
#pragma unroll
                for (int u = 0; u < QB; ++u) {
                    g2_u32x2 h0, l0, h1, l1;
                    f16x2_split4((f32x4){s[u][8 * st + 0], s[u][8 * st + 1], s[u][8 * st + 2], s[u][8 * st + 3]}, 1.0f, h0, l0);
                    f16x2_split4((f32x4){s[u][8 * st + 4], s[u][8 * st + 5], s[u][8 * st + 6], s[u][8 * st + 7]}, 1.0f, h1, l1);
                    b0[u] = __builtin_bit_cast(f16x8, (au32x4){h0[0], h0[1], h1[0], h1[1]});
                    b1[u] = __builtin_bit_cast(f16x8, (au32x4){l0[0], l0[1], l1[0], l1[1]});
                }
                const lc* vrow = v_tr + (sub * 32 + 16 * st) * VP2;
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    typedef __attribute__((address_space(3))) short4v l_s4;
                    const lc* vp = vrow + dt * 64;
                    const au32x2 x0 = __builtin_bit_cast(au32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((l_s4*)(vp)));
                    const au32x2 x1 = __builtin_bit_cast(au32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((l_s4*)(vp + 8 * VP2)));
                    const au32x2 y0 = __builtin_bit_cast(au32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((l_s4*)(vp + 128)));
                    const au32x2 y1 = __builtin_bit_cast(au32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((l_s4*)(vp + 8 * VP2 + 128)));
                    const f16x8 v0 = __builtin_bit_cast(f16x8, (au32x4){x0[0], x0[1], x1[0], x1[1]});
                    const f16x8 v1 = __builtin_bit_cast(f16x8, (au32x4){y0[0], y0[1], y1[0], y1[1]});
#pragma unroll
                    for (int u = 0; u < QB; ++u) {
                        f32x16 c = o[u][dt];
                        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(v1, b0[u], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(v0, b1[u], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(v0, b0[u], c, 0, 0, 0);
                        o[u][dt] = c;
                    }
                }
            }
            KN_ATICK(3)
        }
    }
#ifdef KN_ATT_PROF
    if (tid == 0 && blockIdx.x == 2 && blockIdx.y == 3 && (blockIdx.z % 5) == 0)
        printf("att prof (cycles per 64-key tile, wave 0 of batch %d): stage+barriers %.0f  S (K reads + issue) %.0f  softmax (incl. wait for S) %.0f  PV (split, V reads, issue) %.0f  | whole block %.0f\n",
               (int)blockIdx.z, (double)pf[0] / ntiles, (double)pf[1] / ntiles, (double)pf[2] / ntiles, (double)pf[3] / ntiles, (double)(__builtin_readcyclecounter() - t_begin));
#endif
#undef KN_ATICK
#pragma unroll
    for (int u = 0; u < QB; ++u) {
        if (!qvalid[u]) continue;
        const float inv = 0.0625f / l_run[u];
        float* op = out + ((long)b * T + qi[u]) * E + head * HD;
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                f32x4 v = {o[u][d][r4 * 4 + 0] * inv, o[u][d][r4 * 4 + 1] * inv, o[u][d][r4 * 4 + 2] * inv, o[u][d][r4 * 4 + 3] * inv};
                if (out_split) {
                    g2_u32x2 hi, lo;
                    f16x2_split4(v, KN_F16X2_A_SCALE, hi, lo);
                    const int c = head * HD + d * 32 + r4 * 8 + lh * 4;
                    char* ob = (char*)(out + ((long)b * T + qi[u]) * E) + (c >> 5) * 128 + (c & 31) * 2;
                    *(g2_u32x2*)ob = hi;
                    *(g2_u32x2*)(ob + 64) = lo;
                } else
                    *(f32x4*)(op + d * 32 + r4 * 8 + lh * 4) = v;
            }
    }
}

// ---------------------------------------------------------------------------------------------
// attention2q_kernel with EIGHT waves per workgroup (512 queries, one workgroup per CU) and the two waves of a SIMD out of
// phase.  Round 3's phase counters: a wave spends ~11 000 cycles per 64-key tile — 3 072 of MFMA, ~3 700 of VALU (softmax,
// split of P), 2 000-3 500 around the staging and its two barriers — and the two waves of a SIMD, released by the same
// barriers, want the matrix pipe at the same time and the VALU at the same time: the sum, not the maximum.
//   * K / V tiles in THREE LDS buffers (3 x 37 KB + the bias table: one workgroup per CU), staged two tiles ahead by all
//     512 threads (two float4 of K and of V each: a staged tile serves eight waves); ONE barrier per tile.
//   * every wave walks  softmax(g) -> PV(g) -> S(g + 1)  (the S of step 0 up front); waves 0-3 arrive at the tile's barrier
//     BEFORE the S that opens the next tile, waves 4-7 (their SIMD neighbours) BEHIND it: when the barrier opens one group
//     starts on the matrix pipe and the other on the VALU.  That S may read the first half of the NEXT tile before this
//     tile's barrier, which is why tiles are staged two ahead (that tile was published by the previous barrier).
// Per query the same operations in the same order as attention2q_kernel / attention2_kernel: results are bit-identical.
// ---------------------------------------------------------------------------------------------
template <int QB>
__global__ __launch_bounds__(512, 1) void attention2w_kernel(const float* __restrict__ qkv, const float* __restrict__ gate,
                                                            const float* __restrict__ table, const int* __restrict__ kv_len, int T, int heads,
                                                            float* __restrict__ out, int out_split, int kv_split) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    typedef __attribute__((address_space(3))) char lc;
    typedef __attribute__((address_space(3))) au32x4 l_u4;
    typedef __attribute__((address_space(3))) au32x2 l_u2;
    typedef __attribute__((address_space(3))) float l_f;
    constexpr int NT = 512, NJ = KT * 16 / NT;        // 16 threads per key row: 32 rows per pass, two passes
    constexpr int KVB = KT * KP2 + KT * VP2;          // bytes of one K + V tile
    lc* Ks = (lc*)lds;
    lc* Vs = Ks + KT * KP2;
    l_f* tb = (l_f*)(Ks + 3 * KVB);

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool late = wave >= 4;                      // the group that starts a step with the softmax
    const int li = lane & 31, lh = lane >> 5;
    const int head = blockIdx.y, b = blockIdx.z;
    const int E = heads * HD;
    const long ld = 3L * E;
    const float* base = qkv + (long)b * T * ld;
    int Tk = T;
    if (kv_len) { Tk = kv_len[b]; Tk = Tk < 1 ? 1 : (Tk > T ? T : Tk); }

    for (int i = tid; i < 2 * T - 1 + 64; i += NT) tb[i] = i < 2 * T - 1 ? table[(long)head * (2 * T - 1) + i] : 0.f;

    const float L2E = 1.44269504088896341f;
    const float qscale = 0.125f * L2E * 16.0f;
    int qi[QB]; bool qvalid[QB];
    au32x4 qf[QB][4][2];
    float g_i[QB];
    const l_f* tbq[QB];
#pragma unroll
    for (int u = 0; u < QB; ++u) {
        qi[u] = blockIdx.x * (8 * 32 * QB) + wave * (32 * QB) + u * 32 + li;
        qvalid[u] = qi[u] < T;
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = v0;
            if (qvalid[u]) {
                const float* p = base + (long)qi[u] * ld + head * HD + s4 * 16 + lh * 8;
                v0 = *(const f32x4*)p; v1 = *(const f32x4*)(p + 4);
            }
            g2_u32x2 h0, l0, h1, l1;
            f16x2_split4(v0, qscale, h0, l0); f16x2_split4(v1, qscale, h1, l1);
            qf[u][s4][0] = (au32x4){h0[0], h0[1], h1[0], h1[1]};
            qf[u][s4][1] = (au32x4){l0[0], l0[1], l1[0], l1[1]};
        }
        g_i[u] = qvalid[u] ? gate[((long)b * T + qi[u]) * heads + head] * (L2E * 256.0f) : 0.f;
        tbq[u] = tb + (T - 1 - (qvalid[u] ? qi[u] : T - 1)) + 4 * lh;
    }
    const lc* v_tr = Vs + (4 * (lane >> 5) + ((lane & 15) >> 2)) * VP2 + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;

    f32x16 o[QB][2];
    float m_run[QB], l_run[QB];
#pragma unroll
    for (int u = 0; u < QB; ++u) {
        m_run[u] = -__builtin_inff(); l_run[u] = 0.f;
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[u][d][r] = 0.f;
    }

    const int srow = tid >> 4, scol = (tid & 15) * 4;
    f32x4 rk[NJ], rv[NJ];
    auto gload = [&](int k0) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int key = k0 + srow + (NT / 16) * j;
            f32x4 kk = {0.f, 0.f, 0.f, 0.f}, vv = {0.f, 0.f, 0.f, 0.f};
            if (key < T) {
                const float* p = base + (long)key * ld + head * HD + scol;
                kk = *(const f32x4*)(p + E);
                vv = *(const f32x4*)(p + 2 * E);
            }
            rk[j] = kk; rv[j] = vv;
        }
    };
    auto stage_tile = [&](int boff) {             // the tile in rk / rv -> LDS at byte offset boff
        if (kv_split) {
            const int pc = tid & 15, plane = (pc >> 2) & 1, d0 = (pc >> 3) * 32 + (pc & 3) * 8;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int key = srow + (NT / 16) * j;
                *(l_u4*)(Ks + boff + key * KP2 + plane * 128 + d0 * 2) = __builtin_bit_cast(au32x4, rk[j]);
                *(l_u4*)(Vs + boff + key * VP2 + plane * 128 + d0 * 2) = __builtin_bit_cast(au32x4, rv[j]);
            }
        } else {
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int key = srow + (NT / 16) * j;
                g2_u32x2 hi, lo;
                f16x2_split4(rk[j], 16.0f, hi, lo);
                lc* kd = Ks + boff + key * KP2 + scol * 2;
                *(l_u2*)(kd) = hi;
                *(l_u2*)(kd + 128) = lo;
                f16x2_split4(rv[j], 16.0f, hi, lo);
                lc* vd = Vs + boff + key * VP2 + scol * 2;
                *(l_u2*)(vd) = hi;
                *(l_u2*)(vd + 128) = lo;
            }
        }
    };
    const int ntiles = (Tk + KT - 1) / KT;
    const int nsteps = (Tk + 31) / 32;            // 32-key steps g; step g lives in tile g >> 1, buffer (g >> 1) % 3

    f32x16 s[QB];
    // ---- S^T = K . Q^T for the QB query blocks: every K fragment pair is read once --------------
    auto do_S = [&](int boff, int sub) {
#pragma unroll
        for (int u = 0; u < QB; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) s[u][r] = 0.f;
        const lc* kp = Ks + boff + (sub * 32 + li) * KP2 + lh * 16;
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            const f16x8 k0 = __builtin_bit_cast(f16x8, *(const l_u4*)(kp + st * 32));
            const f16x8 k1 = __builtin_bit_cast(f16x8, *(const l_u4*)(kp + 128 + st * 32));
#pragma unroll
            for (int u = 0; u < QB; ++u) {
                const f16x8 q0 = __builtin_bit_cast(f16x8, qf[u][st][0]), q1 = __builtin_bit_cast(f16x8, qf[u][st][1]);
                s[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(k1, q0, s[u], 0, 0, 0);
                s[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(k0, q1, s[u], 0, 0, 0);
                s[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(k0, q0, s[u], 0, 0, 0);
            }
        }
    };
    // ---- online softmax per query block (lane-local + one exchange with lane ^ 32) ----------------
    auto do_softmax = [&](int g) {
        const int kbase = g * 32;
#pragma unroll
        for (int u = 0; u < QB; ++u) {
            const l_f* tp = tbq[u] + kbase;
#pragma unroll
            for (int r = 0; r < 16; ++r) s[u][r] = fmaf(g_i[u], tp[(r & 3) + 8 * (r >> 2)], s[u][r]);
            if (kbase + 32 > Tk) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (kbase + (r & 3) + 8 * (r >> 2) + 4 * lh >= Tk) s[u][r] = -__builtin_inff();
            }
            float mx = fmaxf(fmaxf(s[u][0], s[u][1]), fmaxf(s[u][2], s[u][3]));
#pragma unroll
            for (int r = 4; r < 16; r += 4) mx = fmaxf(mx, fmaxf(fmaxf(s[u][r], s[u][r + 1]), fmaxf(s[u][r + 2], s[u][r + 3])));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float m_new = fmaxf(m_run[u], mx);
            const float alpha = __builtin_amdgcn_exp2f((m_run[u] - m_new) * (1.0f / 256.0f));
            const float mneg = fmaf(m_new, -1.0f / 256.0f, 14.0f);
            float ps = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[u][r] = __builtin_amdgcn_exp2f(fmaf(s[u][r], 1.0f / 256.0f, mneg)); ps += s[u][r]; }
            ps += __shfl_xor(ps, 32, 64);
            l_run[u] = l_run[u] * alpha + ps;
            m_run[u] = m_new;
            if (__builtin_amdgcn_ballot_w64(alpha != 1.f)) {
#pragma unroll
                for (int d = 0; d < 2; ++d)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[u][d][r] *= alpha;
            }
        }
    };
    // ---- O^T += V^T . P^T: every transposed V fragment is read once -------------------------------
    auto do_PV = [&](int boff, int sub) {
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            f16x8 b0[QB], b1[QB];
#pragma unroll
            for (int u = 0; u < QB; ++u) {
                g2_u32x2 h0, l0, h1, l1;
                f16x2_split4((f32x4){s[u][8 * st + 0], s[u][8 * st + 1], s[u][8 * st + 2], s[u][8 * st + 3]}, 1.0f, h0, l0);
                f16x2_split4((f32x4){s[u][8 * st + 4], s[u][8 * st + 5], s[u][8 * st + 6], s[u][8 * st + 7]}, 1.0f, h1, l1);
                b0[u] = __builtin_bit_cast(f16x8, (au32x4){h0[0], h0[1], h1[0], h1[1]});
                b1[u] = __builtin_bit_cast(f16x8, (au32x4){l0[0], l0[1], l1[0], l1[1]});
            }
            const lc* vrow = v_tr + boff + (sub * 32 + 16 * st) * VP2;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                typedef __attribute__((address_space(3))) short4v l_s4;
                const lc* vp = vrow + dt * 64;
                const au32x2 x0 = __builtin_bit_cast(au32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((l_s4*)(vp)));
                const au32x2 x1 = __builtin_bit_cast(au32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((l_s4*)(vp + 8 * VP2)));
                const au32x2 y0 = __builtin_bit_cast(au32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((l_s4*)(vp + 128)));
                const au32x2 y1 = __builtin_bit_cast(au32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((l_s4*)(vp + 8 * VP2 + 128)));
                const f16x8 v0 = __builtin_bit_cast(f16x8, (au32x4){x0[0], x0[1], x1[0], x1[1]});
                const f16x8 v1 = __builtin_bit_cast(f16x8, (au32x4){y0[0], y0[1], y1[0], y1[1]});
#pragma unroll
                for (int u = 0; u < QB; ++u) {
                    f32x16 c = o[u][dt];
                    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(v1, b0[u], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(v0, b1[u], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(v0, b0[u], c, 0, 0, 0);
                    o[u][dt] = c;
                }
            }
        }
    };

    // tiles 0 and 1 staged, tile 2 on its way in registers; every wave opens with the S of step 0
    gload(0);
    stage_tile(0);
    if (ntiles > 1) { gload(KT); stage_tile(KVB); }
    if (ntiles > 2) gload(2 * KT);
    __syncthreads();
    do_S(0, 0);
    int bc = 0;                                   // byte offset of tile t's buffer (t % 3)
    for (int t = 0; t < ntiles; ++t) {
        // tile t + 2 goes into the buffer whose last readers (tile t - 1) passed the previous barrier; its global loads were issued
        // an iteration ago, those of tile t + 3 start now
        const int bn = bc == 2 * KVB ? 0 : bc + KVB, bp = bc == 0 ? 2 * KVB : bc - KVB;       // tile t + 1, tile t + 2 (= t - 1)
        if (t + 2 < ntiles) stage_tile(bp);
        if (t + 3 < ntiles) gload((t + 3) * KT);
        do_softmax(2 * t); do_PV(bc, 0);
        if (2 * t + 1 < nsteps) { do_S(bc, 1); do_softmax(2 * t + 1); do_PV(bc, 1); }
        // the barrier sits BEFORE the next step's S for waves 0-3 and BEHIND it for waves 4-7 (one arrival per wave and tile either way)
        if (!late) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); }
        if (2 * t + 2 < nsteps) do_S(bn, 0);
        if (late) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); }
        bc = bn;
    }
#pragma unroll
    for (int u = 0; u < QB; ++u) {
        if (!qvalid[u]) continue;
        const float inv = 0.0625f / l_run[u];
        float* op = out + ((long)b * T + qi[u]) * E + head * HD;
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                f32x4 v = {o[u][d][r4 * 4 + 0] * inv, o[u][d][r4 * 4 + 1] * inv, o[u][d][r4 * 4 + 2] * inv, o[u][d][r4 * 4 + 3] * inv};
                if (out_split) {
                    g2_u32x2 hi, lo;
                    f16x2_split4(v, KN_F16X2_A_SCALE, hi, lo);
                    const int c = head * HD + d * 32 + r4 * 8 + lh * 4;
                    char* ob = (char*)(out + ((long)b * T + qi[u]) * E) + (c >> 5) * 128 + (c & 31) * 2;
                    *(g2_u32x2*)ob = hi;
                    *(g2_u32x2*)(ob + 64) = lo;
                } else
                    *(f32x4*)(op + d * 32 + r4 * 8 + lh * 4) = v;
            }
    }
}


}  // namespace

extern "C" int knnsvc_wavlm_attention(const float* qkv, const float* gate, const float* table, int32_t batches,
                                      int32_t T, int32_t heads, float* out, int32_t out_f16x2, int32_t kv_f16x2, const int32_t* kv_len,
                                      void* stream) {
    KN_REQUIRE(qkv && gate && table && out, "wavlm_attention: null pointer");
    KN_REQUIRE(batches > 0 && T > 0 && heads > 0 && heads <= 65535 && batches <= 65535, "wavlm_attention: bad sizes");
    KN_REQUIRE(((uintptr_t)qkv & 15) == 0 && ((uintptr_t)out & 15) == 0, "wavlm_attention: 16-byte alignment");
    static int env_mode = -1;      // KNNSVC_ATTENTION = f16x2 (default) | bf16x3 | fp32 (exact-f32 MFMA kernel)
    if (env_mode < 0) {
        const char* e = getenv("KNNSVC_ATTENTION");
        env_mode = !e ? 2 : (e[0] == 'b') ? 3 : (e[0] == 'f' && e[1] == 'p') ? 0 : 2;
    }
    // bit 2 of the flags: the caller's range analysis says Q / K / V may leave the f16x2 kernel's fixed-scale range
    const int mode = (out_f16x2 & 4) && env_mode == 2 ? 3 : env_mode;
    out_f16x2 &= 1;
    if (mode == 2) {
        const size_t l2 = (size_t)KT * KP2 + (size_t)KT * VP2 + (size_t)(2 * T - 1 + 64) * 4;
        KN_REQUIRE(l2 <= 160 * 1024, "wavlm_attention: T too long for the LDS bias table (T <= ~16000)");
        static size_t attr2 = 0;
        if (l2 > attr2) {
            if (hipFuncSetAttribute((const void*)attention2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l2) != hipSuccess)
                return knnsvc_fail(KNNSVC_EHIP, "wavlm_attention: hipFuncSetAttribute failed");
            attr2 = l2;
        }
        // 64 queries per wave once the 128-query grid is at least two rounds of the 768 resident workgroups (bench batch, 21 x 1500:
        // 0.86 -> 0.78 ms); below that the larger workgroups leave CUs idle (3 x 777: 0.054 vs 0.063 ms).  Same results either
        // way, bit for bit.  KNNSVC_ATT_QB=1 / 2 forces one of them.
        const char* eq = getenv("KNNSVC_ATT_QB");          // read per call: the parity test flips it
        const int qb = !eq ? 0 : (eq[0] == '1' ? 1 : 2);
        const long blocks128 = (long)((T + 127) / 128) * heads * batches;
        if ((qb == 2 || (qb == 0 && blocks128 >= 1536)) && T > 128) {
            // eight waves per workgroup (three K / V buffers, one barrier per tile, SIMD neighbours out of phase) once the 512-query
            // grid is at least two rounds of one workgroup per CU; KNNSVC_ATT_NW=4 / 8 forces one.  Same results, bit for bit.
            const char* ew = getenv("KNNSVC_ATT_NW");
            const long blocks512 = (long)((T + 511) / 512) * heads * batches;
            const size_t l2w = (size_t)3 * (KT * KP2 + KT * VP2) + (size_t)(2 * T - 1 + 64) * 4;
            if ((ew ? ew[0] == '8' : blocks512 >= 512) && l2w <= 160 * 1024) {
                static size_t attr2w = 0;
                if (l2w > attr2w) {
                    if (hipFuncSetAttribute((const void*)attention2w_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l2w) != hipSuccess)
                        return knnsvc_fail(KNNSVC_EHIP, "wavlm_attention: hipFuncSetAttribute failed");
                    attr2w = l2w;
                }
                dim3 gridw((unsigned)((T + 511) / 512), (unsigned)heads, (unsigned)batches);
                hipLaunchKernelGGL(attention2w_kernel<2>, gridw, dim3(512), l2w, (hipStream_t)stream, qkv, gate, table, kv_len, T, heads, out, out_f16x2, kv_f16x2);
                return knnsvc_check_launch("wavlm_attention2w");
            }
            static size_t attr2q = 0;
            if (l2 > attr2q) {
                if (hipFuncSetAttribute((const void*)attention2q_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l2) != hipSuccess)
                    return knnsvc_fail(KNNSVC_EHIP, "wavlm_attention: hipFuncSetAttribute failed");
                attr2q = l2;
            }
            dim3 gridq((unsigned)((T + 255) / 256), (unsigned)heads, (unsigned)batches);
            hipLaunchKernelGGL(attention2q_kernel<2>, gridq, dim3(256), l2, (hipStream_t)stream, qkv, gate, table, kv_len, T, heads, out, out_f16x2, kv_f16x2);
            return knnsvc_check_launch("wavlm_attention2q");
        }
        dim3 grid2((unsigned)((T + 127) / 128), (unsigned)heads, (unsigned)batches);
        hipLaunchKernelGGL(attention2_kernel, grid2, dim3(256), l2, (hipStream_t)stream, qkv, gate, table, kv_len, T, heads, out, out_f16x2, kv_f16x2);
        return knnsvc_check_launch("wavlm_attention2");
    }
    KN_REQUIRE(!out_f16x2 && !kv_f16x2, "wavlm_attention: split output / pre-split K,V are only implemented by the f16x2 kernel");
    if (mode == 3) {
        const size_t l3 = (size_t)KT * KP3 + (size_t)HD * VP3 + (size_t)(2 * T - 1 + 64) * 4;
        KN_REQUIRE(l3 <= 160 * 1024, "wavlm_attention: T too long for the LDS bias table (T <= ~13000)");
        static size_t attr3 = 0;
        if (l3 > attr3) {
            if (hipFuncSetAttribute((const void*)attention3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l3) != hipSuccess)
                return knnsvc_fail(KNNSVC_EHIP, "wavlm_attention: hipFuncSetAttribute failed");
            attr3 = l3;
        }
        dim3 grid3((unsigned)((T + 127) / 128), (unsigned)heads, (unsigned)batches);
        hipLaunchKernelGGL(attention3_kernel, grid3, dim3(256), l3, (hipStream_t)stream, qkv, gate, table, kv_len, T, heads, out);
        return knnsvc_check_launch("wavlm_attention3");
    }
    const size_t lds = (size_t)(KT * LDKK + KT * LDV + 2 * T - 1) * 4;
    KN_REQUIRE(lds <= 160 * 1024, "wavlm_attention: T too long for the LDS bias table (T <= ~16000)");
    static size_t attr = 0;
    if (lds > attr) {
        if (hipFuncSetAttribute((const void*)attention_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
            hipSuccess)
            return knnsvc_fail(KNNSVC_EHIP, "wavlm_attention: hipFuncSetAttribute failed");
        attr = lds;
    }
    dim3 grid((unsigned)((T + 127) / 128), (unsigned)heads, (unsigned)batches);
    hipLaunchKernelGGL(attention_kernel, grid, dim3(256), lds, (hipStream_t)stream, qkv, gate, table, kv_len, T, heads, out);
    return knnsvc_check_launch("wavlm_attention");
}
