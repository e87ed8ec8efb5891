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

namespace {

constexpr int HD = 64;          // head dim
constexpr int KT = 64;          // keys per LDS tile
constexpr int LDKK = 68;        // K tile pitch (floats): 16 consecutive rows hit 16 distinct 4-bank slots
constexpr int LDV = 64;

__global__ __launch_bounds__(256) void attention_kernel(const float* __restrict__ qkv, const float* __restrict__ gate,
                                                       const float* __restrict__ table, int T, int heads,
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
    const int ntiles = (T + KT - 1) / KT;
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
            if (kbase >= T) break;
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
                if (key < T && qvalid) v = s[r] + g_i * tb[key - qi + T - 1];
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

}  // namespace

extern "C" int knnsvc_wavlm_attention(const float* qkv, const float* gate, const float* table, int32_t batches,
                                      int32_t T, int32_t heads, float* out, void* stream) {
    KN_REQUIRE(qkv && gate && table && out, "wavlm_attention: null pointer");
    KN_REQUIRE(batches > 0 && T > 0 && heads > 0 && heads <= 65535 && batches <= 65535, "wavlm_attention: bad sizes");
    KN_REQUIRE(((uintptr_t)qkv & 15) == 0 && ((uintptr_t)out & 15) == 0, "wavlm_attention: 16-byte alignment");
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
    hipLaunchKernelGGL(attention_kernel, grid, dim3(256), lds, (hipStream_t)stream, qkv, gate, table, T, heads, out);
    return knnsvc_check_launch("wavlm_attention");
}
