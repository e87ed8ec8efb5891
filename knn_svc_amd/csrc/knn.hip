// Cosine-distance kNN: fp32-MFMA q.p^T tile GEMM fused with the reference's distance
// epilogue and a wavefront bitonic top-k.  Reference: lib_ongaku_test.py:148-175
// (fast_cosine_dist), ddsp_prematch_dataset.py:1195-1210 (20-row chunks + topk(32)).
//
// Grid: x = 128-row query tile, y = pool split.  A block walks its pool slice in tiles of
// 128 rows; after each tile's K loop the 128x128 distances go to LDS and every wave scans
// its 32 query rows against the row's current 32nd-best key.  Survivors are merged into the
// row's sorted list (LDS) by a 64-lane bitonic sort.  Per-split lists are merged by a second
// small kernel.  Keys are (order-preserving bits of the f32 distance) << 32 | pool index, so
// ties resolve to the lower pool index on every device count.
#include <stdlib.h>
#include "gemm_core.h"
#include "gemm2_core.h"

namespace {

using G = GemmTile<128, 128, 4, 2, 1, 2, 512>;      // 8 waves (two per SIMD inside the one block a CU can hold)
constexpr int NT = 512;
constexpr int KMAX = 32;
constexpr unsigned long long KEY_INF = 0xFFFFFFFFFFFFFFFFull;

__device__ __forceinline__ unsigned sortable(float d) {
    unsigned u = __float_as_uint(d);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float unsortable(unsigned s) {
    unsigned u = (s & 0x80000000u) ? (s & 0x7FFFFFFFu) : ~s;
    return __uint_as_float(u);
}

// ascending bitonic sort of one u64 per lane across the 64-lane wave
__device__ __forceinline__ unsigned long long wave_sort64(unsigned long long v, int lane) {
#pragma unroll
    for (int k = 2; k <= 64; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
            unsigned long long o = __shfl_xor(v, j, 64);
            const bool up = ((lane & k) == 0);          // ascending block
            const bool lower = ((lane & j) == 0);
            const bool take_min = (up == lower);
            const unsigned long long mn = v < o ? v : o, mx = v < o ? o : v;
            v = take_min ? mn : mx;
        }
    }
    return v;
}

__device__ __forceinline__ unsigned long long readlane64(unsigned long long v, int l) {      // l wave-uniform
    const unsigned lo = __builtin_amdgcn_readlane((unsigned)v, l), hi = __builtin_amdgcn_readlane((unsigned)(v >> 32), l);
    return ((unsigned long long)hi << 32) | lo;
}

// Sorted list of k <= 32 keys held one per lane (lanes >= k hold KEY_INF): insert one key that is known to be smaller
// than the current k-th.  One ballot + popcount finds the slot, one lane shift makes room — a few instructions where
// the bitonic merge costs 21 compare-exchange stages (keys are unique: the pool index sits in their low half).
__device__ __forceinline__ void list_insert(unsigned long long& mine, unsigned long long key, int lane, int k) {
    const int pos = __popcll(__ballot(mine < key));
    const unsigned long long up = __shfl_up(mine, 1, 64);
    if (lane == pos) mine = key;
    else if (lane > pos) mine = up;
    if (lane >= k) mine = KEY_INF;
}

// the reference's operation sequence on top of dot = sum_k q_k p_k :
//   cdist (mm route) : r = -2*dot + |q|^2 + |p|^2 ; cd = sqrt(max(r, 1e-30))
//   fast_cosine_dist : 1 - (((-cd*cd + qn*qn) + pn*pn) / 2) / (qn*pn)
__device__ __forceinline__ float ref_distance(float dot, float qsq, float psq, float qn, float pn) {
#pragma clang fp contract(off)      // every product below is rounded on its own, as in the reference
    float r = (-2.0f * dot + qsq) + psq;
    float cd = sqrtf(fmaxf(r, 1e-30f));
    float dp = ((-(cd * cd)) + qn * qn) + pn * pn;
    dp = dp / 2.0f;
    return 1.0f - dp / (qn * pn);
}

struct RowLoader {      // rows of a [rows][dim] row-major matrix, 16-byte vectors
    const float* base; long rows; int dim; long r0;
    __device__ __forceinline__ f32x4 operator()(int kt, int j, int) const {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        const long r = r0 + (threadIdx.x >> 3) + G::RP * j;
        const int k = kt * 32 + (threadIdx.x & 7) * 4;
        if (r < rows && k < dim) v = *(const f32x4*)(base + r * dim + k);
        return v;
    }
    __device__ __forceinline__ f32x4 finish(f32x4 v) const { return v; }
    __device__ __forceinline__ void begin(int) const {}
};

constexpr int LDD = 128;                                  // distance tile pitch
constexpr int LDS_STAGE = G::LDS_FLOATS;                  // 18432 floats (>= 128*128)
constexpr int KNN_LDS_BYTES = LDS_STAGE * 4 + 128 * KMAX * 8 /*lists*/ + 8 * 128 * 8 /*scratch*/ + 128 * 4 * 2;

__global__ __launch_bounds__(512) void knn_tile_kernel(
    const float* __restrict__ q, const float* __restrict__ qn, const float* __restrict__ qsq, long nq,
    const float* __restrict__ pool, const float* __restrict__ pn, const float* __restrict__ psq, long np,
    int dim, int k, long rows_per_split, long mask_lo, long mask_hi, unsigned long long* __restrict__ part, int* nan_flag) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* dist = lds;                                                       // [128][LDD] (aliases staging)
    unsigned long long* lists = (unsigned long long*)(lds + LDS_STAGE);      // [128][32]
    unsigned long long* scratch = lists + 128 * KMAX;                        // [8 waves][128]
    float* s_qn = (float*)(scratch + 8 * 128);
    float* s_qsq = s_qn + 128;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long q0 = (long)blockIdx.x * 128;
    const long p_begin = (long)blockIdx.y * rows_per_split;
    const long p_end = p_begin + rows_per_split < np ? p_begin + rows_per_split : np;

    for (int i = tid; i < 128 * KMAX; i += NT) lists[i] = KEY_INF;
    if (tid < 128) {
        const long r = q0 + tid;
        s_qn[tid] = r < nq ? qn[r] : 1.f;
        s_qsq[tid] = r < nq ? qsq[r] : 0.f;
    }
    __syncthreads();

    const int nk = (dim + 31) / 32;
    bool saw_nan = false;
    for (long p0 = p_begin; p0 < p_end; p0 += 128) {
        f32x16 acc[G::TM][G::TN];
#pragma unroll
        for (int i = 0; i < G::TM; ++i)
#pragma unroll
            for (int j = 0; j < G::TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        RowLoader al{q, nq, dim, q0};
        RowLoader bl{pool, p_end, dim, p0};
        G::mainloop(lds, nk, al, bl, acc, 0, 0);            // ends with __syncthreads(): staging LDS is free

        // distances -> LDS tile [query][pool]
#pragma unroll
        for (int j = 0; j < G::TN; ++j) {
            const int c = G::acc_col(wave, lane, j);
            const long p = p0 + c;
            const bool pv = p < p_end;
            const float v_pn = pv ? pn[p] : 1.f, v_psq = pv ? psq[p] : 0.f;
#pragma unroll
            for (int i = 0; i < G::TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = G::acc_row(wave, lane, i, r);
                    float d = ref_distance(acc[i][j][r], s_qsq[row], v_psq, s_qn[row], v_pn);
                    if (pv && (q0 + row) < nq && d != d) saw_nan = true;
                    if (p >= mask_lo && p < mask_hi) d = 1.f;          // dists[:, start:end] = 1 (ddsp_prematch_dataset.py:1607)
                    dist[row * LDD + c] = pv ? d : __builtin_inff();
                }
        }
        __syncthreads();

        // selection: wave w owns query rows 16w .. 16w+15
        unsigned long long* my_scratch = scratch + wave * 128;
        for (int rr = 0; rr < 16; ++rr) {
            const int row = wave * 16 + rr;
            if (q0 + row >= nq) break;
            unsigned long long* lst = lists + row * KMAX;
            unsigned long long thr = lst[k - 1];
            const float d0 = dist[row * LDD + lane], d1 = dist[row * LDD + 64 + lane];
            // NaN / +inf never enter: their sortable bits are >= those of +inf
            unsigned long long k0 = ((unsigned long long)sortable(d0) << 32) | (unsigned)(p0 + lane);
            unsigned long long k1 = ((unsigned long long)sortable(d1) << 32) | (unsigned)(p0 + 64 + lane);
            const bool f0 = (d0 < __builtin_inff()) && k0 < thr;
            const bool f1 = (d1 < __builtin_inff()) && k1 < thr;
            const unsigned long long b0 = __ballot(f0), b1 = __ballot(f1);
            if ((b0 | b1) == 0ull) continue;
            const unsigned long long lt = (1ull << lane) - 1ull;
            const int c0 = __popcll(b0), total = c0 + __popcll(b1);
            if (f0) my_scratch[__popcll(b0 & lt)] = k0;
            if (f1) my_scratch[c0 + __popcll(b1 & lt)] = k1;
            __builtin_amdgcn_wave_barrier();
            for (int base = 0; base < total; base += 32) {
                unsigned long long v;
                if (lane < 32) v = lane < k ? lst[lane] : KEY_INF;
                else v = (base + lane - 32) < total ? my_scratch[base + lane - 32] : KEY_INF;
                v = wave_sort64(v, lane);
                if (lane < k) lst[lane] = v;
                __builtin_amdgcn_wave_barrier();
            }
        }
        __syncthreads();      // dist tile is about to be overwritten by the next tile's staging
    }
    if (saw_nan) atomicOr(nan_flag, 1);

    // per-split lists -> workspace  part[split][q][k]
    for (int i = tid; i < 128 * k; i += NT) {
        const int row = i / k, e = i - row * k;
        if (q0 + row < nq) part[((long)blockIdx.y * nq + q0 + row) * k + e] = lists[row * KMAX + e];
    }
}

// Top-k from a precomputed dot-product matrix (the two-kernel route: q.p^T comes from the emulated-fp32 GEMM of
// conv_gemm.hip, 3-4x the rate of the fp32 MFMA tile above; this kernel replays the reference's distance formula on it
// and selects).  One block per query row, 4 waves; wave w scans pool columns [128 w + 512 t, +128) with the same
// threshold filter + 64-lane bitonic merge as knn_tile_kernel, wave 0 then folds the four sorted lists.
template <bool SCREEN>
__global__ __launch_bounds__(256) void knn_select_kernel(
    const float* __restrict__ dots, long ld, const float* __restrict__ qn, const float* __restrict__ qsq, long nq,
    const float* __restrict__ pn, const float* __restrict__ psq, long np, int k, long idx_offset,
    long mask_lo, long mask_hi, long* __restrict__ out_idx, float* __restrict__ out_dist, int* nan_flag) {
    __shared__ unsigned long long lists[4][KMAX];
    __shared__ unsigned long long scratch[4][128];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long row = blockIdx.x;
    const float* drow = dots + row * ld;
    const float v_qn = qn[row], v_qsq = qsq[row];
    unsigned long long* my_scratch = scratch[wave];
    // the wave's sorted top-k lives in registers, one key per lane (lanes >= k: KEY_INF).  After the first tiles a
    // surviving candidate is rare and single: it is inserted with list_insert; only bursts (the warm-up, when everything
    // beats an empty list) go through the scratch buffer and the 64-lane bitonic merge.
    unsigned long long mine = KEY_INF, thr = KEY_INF;
    bool saw_nan = false, has_thr = false;      // has_thr: the list holds k real entries, thr_d is its largest distance
    float thr_d = 0.f;
    const float v_rq = __builtin_amdgcn_rcpf(v_qn);
#ifndef KN_SELECT_U
#define KN_SELECT_U 2          // 2: 156 us at 1500 x 30000; 4: 164; 8: 200; 16: 220 (kernel time under rocprofv3) — the warm-up merges bound it, not load latency
#endif
    constexpr int U = KN_SELECT_U;     // sub-chunks of 128 columns per wave and iteration, all loads issued up front
    for (long base0 = (long)wave * (128 * U); base0 < np; base0 += 512 * U) {
        float dv[2 * U], sv[2 * U], nv[2 * U];
#pragma unroll
        for (int u = 0; u < 2 * U; ++u) {
            const long p = base0 + u * 64 + lane;
            const bool in = p < np;
            dv[u] = in ? drow[p] : 0.f; sv[u] = in ? psq[p] : 0.f; nv[u] = in ? pn[p] : 1.f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long base = base0 + u * 128;
            if (base >= np) break;
            const long p0 = base + lane, p1 = base + 64 + lane;
            // Screen: the reference formula costs a correctly rounded sqrt and two divisions per element, and almost no
            // element can beat the current k-th distance.  approx = 1 - dot / (|q||p|) in three instructions; the formula's
            // own rounding (r = -2 dot + |q|^2 + |p|^2 cancels to ~2 dot) keeps |reference - exact| below
            // ~3 eps (|q|/|p| + |p|/|q| + 2), the approximation adds a few eps: a margin of 64 eps (1 + ratio) is > 10x that.
            // Any NaN / inf in the inputs turns approx or the margin into NaN, and a NaN comparison sends the sub-chunk
            // down the exact path — so NaN detection and the result are those of evaluating the formula everywhere.
            if (SCREEN && has_thr) {
                const float c0s = v_rq * __builtin_amdgcn_rcpf(nv[2 * u]), c1s = v_rq * __builtin_amdgcn_rcpf(nv[2 * u + 1]);
                const float a0 = 1.0f - dv[2 * u] * c0s, a1 = 1.0f - dv[2 * u + 1] * c1s;
                const float g0 = fmaf((v_qsq + sv[2 * u]) * c0s, 3.8e-6f, 3.8e-6f), g1 = fmaf((v_qsq + sv[2 * u + 1]) * c1s, 3.8e-6f, 3.8e-6f);
                bool need = (p0 < np && !(a0 >= thr_d + g0)) || (p1 < np && !(a1 >= thr_d + g1));
                if (mask_lo < mask_hi) need = need || (p0 >= mask_lo && p0 < mask_hi) || (p1 >= mask_lo && p1 < mask_hi);
                if (__ballot(need) == 0ull) continue;
            }
            float d0 = __builtin_inff(), d1 = __builtin_inff();
            if (p0 < np) { d0 = ref_distance(dv[2 * u], v_qsq, sv[2 * u], v_qn, nv[2 * u]); if (d0 != d0) saw_nan = true; }
            if (p1 < np) { d1 = ref_distance(dv[2 * u + 1], v_qsq, sv[2 * u + 1], v_qn, nv[2 * u + 1]); if (d1 != d1) saw_nan = true; }
            // self-matching (ddsp_prematch_dataset.py:1606-1607): the query's own utterance competes at distance exactly 1,
            // after the NaN check (fast_cosine_dist exits on NaN before the caller overwrites anything)
            if (p0 >= mask_lo && p0 < mask_hi) d0 = 1.f;
            if (p1 >= mask_lo && p1 < mask_hi) d1 = 1.f;
            // NaN / +inf never enter: their sortable bits are >= those of +inf
            const unsigned long long k0 = ((unsigned long long)sortable(d0) << 32) | (unsigned)p0;
            const unsigned long long k1 = ((unsigned long long)sortable(d1) << 32) | (unsigned)p1;
            const bool f0 = (d0 < __builtin_inff()) && k0 < thr;
            const bool f1 = (d1 < __builtin_inff()) && k1 < thr;
            unsigned long long b0 = __ballot(f0), b1 = __ballot(f1);
            if ((b0 | b1) == 0ull) continue;
            const int c0 = __popcll(b0), total = c0 + __popcll(b1);
            if (total <= 8) {
                while (b0) {
                    const int l = __builtin_ctzll(b0); b0 &= b0 - 1;
                    const unsigned long long key = readlane64(k0, l);
                    if (key < thr) { list_insert(mine, key, lane, k); thr = readlane64(mine, k - 1); }
                }
                while (b1) {
                    const int l = __builtin_ctzll(b1); b1 &= b1 - 1;
                    const unsigned long long key = readlane64(k1, l);
                    if (key < thr) { list_insert(mine, key, lane, k); thr = readlane64(mine, k - 1); }
                }
            } else {
                const unsigned long long lt = (1ull << lane) - 1ull;
                if (f0) my_scratch[__popcll(b0 & lt)] = k0;
                if (f1) my_scratch[c0 + __popcll(b1 & lt)] = k1;
                __builtin_amdgcn_wave_barrier();
                for (int b = 0; b < total; b += 32) {
                    unsigned long long v = mine;                       // lanes < 32: the list (KEY_INF beyond k)
                    if (lane >= 32) v = (b + lane - 32) < total ? my_scratch[b + lane - 32] : KEY_INF;
                    v = wave_sort64(v, lane);
                    mine = lane < k ? v : KEY_INF;
                }
                __builtin_amdgcn_wave_barrier();
                thr = readlane64(mine, k - 1);
            }
            has_thr = thr != KEY_INF;
            thr_d = unsortable((unsigned)(thr >> 32));
        }
    }
    if (saw_nan) atomicOr(nan_flag, 1);
    if (lane < KMAX) lists[wave][lane] = mine;
    __syncthreads();
    if (wave == 0) {
        unsigned long long best = lane < k ? lists[0][lane] : KEY_INF;
        for (int s = 1; s < 4; ++s) {
            unsigned long long v = best;
            if (lane >= 32) v = (lane - 32) < k ? lists[s][lane - 32] : KEY_INF;
            else if (lane >= k) v = KEY_INF;
            best = wave_sort64(v, lane);
        }
        if (lane < k) {
            out_idx[row * k + lane] = (long)(unsigned)(best & 0xFFFFFFFFull) + idx_offset;
            out_dist[row * k + lane] = unsortable((unsigned)(best >> 32));
        }
    }
}

// one wave per query row: fold `parts` sorted key lists into one
__global__ __launch_bounds__(256) void knn_merge_keys_kernel(const unsigned long long* __restrict__ part,
                                                            int parts, long nq, int k, long idx_offset,
                                                            long* __restrict__ out_idx, float* __restrict__ out_dist) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= nq) return;
    unsigned long long best = lane < k ? part[row * k + lane] : KEY_INF;
    for (int s = 1; s < parts; ++s) {
        unsigned long long v = best;
        if (lane >= 32) v = (lane - 32) < k ? part[((long)s * nq + row) * k + lane - 32] : KEY_INF;
        else if (lane >= k) v = KEY_INF;
        best = wave_sort64(v, lane);
    }
    if (lane < k) {
        out_idx[row * k + lane] = (long)(unsigned)(best & 0xFFFFFFFFull) + idx_offset;
        out_dist[row * k + lane] = unsortable((unsigned)(best >> 32));
    }
}

// merge of (dist, global idx) lists coming from other devices
__global__ __launch_bounds__(256) void knn_merge_pairs_kernel(const float* __restrict__ pd, const long* __restrict__ pi,
                                                             int parts, long nq, int k,
                                                             long* __restrict__ out_idx, float* __restrict__ out_dist) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= nq) return;
    auto load = [&](int s, int e) -> unsigned long long {
        const long o = ((long)s * nq + row) * k + e;
        const float d = pd[o];
        if (!(d < __builtin_inff())) return KEY_INF;
        return ((unsigned long long)sortable(d) << 32) | (unsigned)pi[o];
    };
    unsigned long long best = lane < k ? load(0, lane) : KEY_INF;
    best = wave_sort64(best, lane);
    for (int s = 1; s < parts; ++s) {
        unsigned long long v = best;
        if (lane >= 32) v = (lane - 32) < k ? load(s, lane - 32) : KEY_INF;
        else if (lane >= k) v = KEY_INF;
        best = wave_sort64(v, lane);
    }
    if (lane < k) {
        out_idx[row * k + lane] = (long)(unsigned)(best & 0xFFFFFFFFull);
        out_dist[row * k + lane] = unsortable((unsigned)(best >> 32));
    }
}

__global__ __launch_bounds__(256) void row_norms_kernel(const float* __restrict__ x, long rows, int dim, int ldx,
                                                       float* __restrict__ norm, float* __restrict__ sq, float* __restrict__ max_slot) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + row * (long)ldx;
    double s = 0.0;
    for (int c = lane; c < dim; c += 64) { const double v = xr[c]; s += v * v; }
    s = wave_sum_d(s);
    if (lane == 0) {
        const float nr = (float)sqrt(s);
        if (sq) sq[row] = (float)s;
        if (norm) norm[row] = nr;
        // range slot: max row norm >= max |x| (bit-pattern max: a NaN row makes the slot NaN); relaxed pre-check keeps
        // the same-address atomics to the few rows that raise the maximum
        if (max_slot) {       // a range slot is 64 stripes of one cache line each (conv_gemm.hip): one fire-and-forget atomic per row
            const unsigned b = __float_as_uint(nr) & 0x7FFFFFFFu;
            if (b) atomicMax((unsigned*)max_slot + (blockIdx.x & 63) * 32, b);
        }
    }
}

// -------------------------------------------------------------------------------------------------
// Fused route for LARGE query sets (BASELINE cfg 5: 24 000+ query frames per search): no [Nq, Np] dot matrix.
//   pass 1 (host, existing kernels): exact top-k against a SAMPLE of the pool -> thr[row] = a distance the row's k-th best
//           cannot exceed;
//   pass 2 (knn_screen_kernel): q.p^T on the 256x256 / 128x128-wave-tile f16x2 main loop (Gemm2QuadR, the fastest loop of
//           the library at K = 1024: no output to store here), each accumulator element screened IN REGISTERS against its
//           row's thr with the same conservative margin as knn_select_kernel; the few survivors (pool index, dot) are
//           compacted per row into a candidate buffer — LDS-staged per tile, one global atomic per (tile, row);
//   pass 3 (knn_refine_kernel): the reference's distance formula on the candidates only, top-k with the same keys.
// The candidate set is a superset of the true top-k (thr is an exact k-th distance over a subset; the margin covers the
// screen's rounding), so the result is identical to evaluating the formula on every pair.  Traffic: 8 B per survivor
// (~0.4 % of the pairs at a 1/22 sample) instead of 8 B per pair.
// -------------------------------------------------------------------------------------------------
typedef unsigned kn_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t knn_rsrc(const void* p, int bytes) {
    const unsigned long long u = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
    bytes = __builtin_amdgcn_readfirstlane(bytes);
    return __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), 0, bytes, 0x00020000);
}
__device__ __forceinline__ float knn_pick_scale(const float* slot) {      // == kn_pick_scale(kn_slot_max(slot)) of conv_gemm.hip
    unsigned m = 0;
#pragma unroll
    for (int i = 0; i < 64; ++i) { const unsigned v = __float_as_uint(slot[i * 32]) & 0x7FFFFFFFu; m = v > m ? v : m; }
    unsigned e = (m >> 23) & 0xFFu;
    e = e < 87u ? 87u : e;
    return __uint_as_float((268u - e) << 23);
}

// Gemm2QuadS (v_mfma_f32_16x16x32_f16), the loop of the encoder's large GEMMs.  The dot-matrix route runs its GEMM on the SAME
// loop whatever its size (knnsvc_conv_desc.fixed_tile = 2), so both routes — and pool shards of any size — sum over K in the same
// grouping and give the same dot-product BITS: the distances, and with them the order of near-ties, do not depend on the route
// (tests: fused == dot-matrix, sharded == unsharded, bit for bit).  Round 2 kept both on the 32x32x16 grouping (Gemm2QuadR here,
// the 128x128 kernel there) and paid ~15 % of the matrix rate for it.
using QG = Gemm2QuadS;
constexpr int SCR_LIST = 3072;                       // entries of the tile list in LDS (flushed to the rows' global slots when full)
constexpr int SCR_QD = 40;                           // entries of a lane's queue (drained before a column's 32 elements could overflow it)

// Persistent: block b walks tiles b, b + gridDim.x, ... (gridDim.x a multiple of 8: a tile keeps its XCD).  The host caps the
// grid (`max_blocks`): inside a stream pipeline the search then leaves CUs to the single-workgroup recurrences and the generator
// of the other items instead of occupying every CU with a 128 KB-LDS block (VERDICT r2 #5b).
__global__ __launch_bounds__(256, 1) void knn_screen_kernel(
    const float* __restrict__ q2, const float* __restrict__ q_absmax, const float* __restrict__ qn, const float* __restrict__ qsq, long nq,
    const unsigned short* __restrict__ p2, const float* __restrict__ p_absmax, const float* __restrict__ pn, const float* __restrict__ psq,
    long np, int dim, const float* __restrict__ thr, const long* __restrict__ thr_idx, long mask_lo, long mask_hi,
    int* __restrict__ cand_count, unsigned* __restrict__ cand, int cap, int* __restrict__ overflow, int ntiles) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int gx = (int)((nq + 255) / 256), gy = (int)((np + 255) / 256);
    const float out_scale = 1.0f / (knn_pick_scale(q_absmax) * knn_pick_scale(p_absmax));
  for (int vt = blockIdx.x; vt < ntiles; vt += gridDim.x) {
    int mt, nt;
    if (!quad_order_decode(vt, gx, gy, mt, nt)) continue;          // XCD-aware order, padding ids (gemm2_core.h)
    const int m0 = mt * 256, n0 = nt * 256;

    typename QG::acc_t acc[QG::TM][QG::TN];
#pragma unroll
    for (int i = 0; i < QG::TM; ++i)
#pragma unroll
        for (int j = 0; j < QG::TN; ++j)
#pragma unroll
            for (int r = 0; r < QG::NR; ++r) acc[i][j][r] = 0.f;
    const int M = (int)nq, row_bytes = dim * 4;
    auto row_off = [&](int m) -> int { return m < M ? m * row_bytes : QG::OOB_OFF; };
    auto step = [&](int kt) -> int { return kt * 128; };
    QG::mainloop(lds, dim / 32, row_off, step, knn_rsrc(q2, (int)(nq * row_bytes)), knn_rsrc(p2, (int)(np * (long)(dim / 32) * 128)),
                 (int)np, dim, m0, n0, acc);
    __syncthreads();                                   // the operand stages are free: row data, counters and the survivor list

    // ---- screening epilogue.  Round 2's form tested every accumulator element under a branch that held two LDS atomics: with
    // ~1 survivor per (row, tile) some lane of a wave took the slow path at ~40 % of its 256 elements — 40-80 us per tile next to a
    // 64 us main loop — and its only test was the conservative one, which lets every near-tie through: on audio with silence
    // (thousands of near-identical pool frames) rows overflowed their candidate buffers and the whole search fell back.  Now:
    // (1) a branch-light COARSE pass over the registers, 3 VALU per element: knn_select_kernel's conservative screen
    //        1 - dot/(|q||p|) < thr + eps (1 + (|q|^2 + |p|^2)/(|q||p|))
    //     rearranged to  acc > A1[row] * |p| - (A2[row] + B2[col])  (eps' = 4.5e-6 instead of 3.8e-6 absorbs this form's rounding);
    // (2) on its few survivors the EXACT test: the reference's distance from this very dot product (ref_distance, the value the
    //     refine pass would compute) as a (distance bits, index) key against key32[row], the row's k-th best key over the sample.
    //     A pool row can only be among the true k best if its key is <= that — ties included, they are ordered by index — so what
    //     passes is exactly the set {key <= key32}: its size is the RANK of the sample's k-th best in the whole chunk (about
    //     k * stride), however many near-identical rows the pool holds;
    // (3) survivors go to the lane's private LDS queue (lane-interleaved: conflict-free), no atomics; the queues are drained
    //     into the tile list, and the list is flushed to the rows' global candidate slots (one global atomic per (flush, row))
    //     as often as needed — a tile of silence against silence legitimately yields tens of thousands of candidates.
    float* s_a1 = lds;                                 // [256] |q| (1 - thr - eps') / out_scale   (+huge for rows past nq)
    float* s_a2 = s_a1 + 256;                          // [256] eps' |q|^2 / out_scale
    float* s_qn = s_a2 + 256;                          // [256] |q|, |q|^2: the exact formula's row operands
    float* s_qsq = s_qn + 256;
    unsigned* s_khi = (unsigned*)(s_qsq + 256);        // [256] key32: distance bits, pool index
    unsigned* s_klo = s_khi + 256;
    float* s_pn = (float*)(s_klo + 256);               // [256] |p|, |p|^2 of the tile's pool rows: the exact formula's column operands
    float* s_psq = s_pn + 256;
    int* s_cnt = (int*)(s_psq + 256);                  // [256] entries per row in the current list
    int* s_base = s_cnt + 256;                         // [256] their first slot in the row's global candidate list
    int* s_n = s_base + 256;                           // [1] entries in the tile list
    unsigned* s_list = (unsigned*)(s_n + 4);           // [SCR_LIST][3]: row << 16 | position in row, pool index, dot bits
    unsigned* s_queue = s_list + SCR_LIST * 3;         // [SCR_QD][256][2]: lane queues: row << 8 | column, dot bits
    // (laundered: derived from a plain threadIdx.x, the epilogue's per-lane rows / columns / LDS addresses are loop-invariant,
    //  get hoisted out of the tile loop and then live — and spill — across the main loop, which has no register to spare)
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63, wave = tid >> 6;
    constexpr float EPS = 4.5e-6f;
    const float inv_os = 1.0f / out_scale;             // powers of two: exact
    {
        const long r = (long)m0 + tid;
        const bool v = r < nq;
        const float t = v ? thr[r] : 0.f, n_ = v ? qn[r] : 0.f, sq_ = v ? qsq[r] : 0.f;
        s_a1[tid] = v ? n_ * ((1.0f - t) - EPS) * inv_os : 3.0e38f;        // rows past nq: nothing survives
        s_a2[tid] = v ? EPS * sq_ * inv_os : 0.f;
        s_qn[tid] = n_; s_qsq[tid] = sq_;
        // a NaN / +inf threshold (a NaN query row) keeps every pair: the refine pass then reports it
        const bool open = !(t < __builtin_inff());
        s_khi[tid] = open ? 0xFFFFFFFFu : sortable(t);
        s_klo[tid] = open ? 0xFFFFFFFFu : (unsigned)(v ? thr_idx[r] : 0);
        s_cnt[tid] = 0;
        if (tid == 0) s_n[0] = 0;
        const long pc = (long)n0 + tid;
        s_pn[tid] = pc < np ? pn[pc] : 0.f;
        s_psq[tid] = pc < np ? psq[pc] : 0.f;
    }
    __syncthreads();
    const bool masked = mask_lo < mask_hi;
    float ra1[QG::TM][QG::NR], ra2[QG::TM][QG::NR];
#pragma unroll
    for (int i = 0; i < QG::TM; ++i)
#pragma unroll
        for (int r = 0; r < QG::NR; ++r) {
            const int row = QG::acc_row(wave, lane, i, r);
            ra1[i][r] = s_a1[row]; ra2[i][r] = s_a2[row];
        }
    int qcnt = 0;
    bool capped = false;
    // all lanes: queues -> tile list -> global slots, repeated until every queue is empty (block-uniform control flow)
    auto drain_and_flush = [&]() __attribute__((always_inline)) {
        int e = 0;
        for (;;) {
            for (; e < qcnt; ++e) {
                const unsigned* qe = s_queue + ((e * 256 + tid) << 1);
                const unsigned code = qe[0];
                const int row = (int)(code >> 8), col = (int)(code & 255u);
                // the EXACT test (2): the distance the refine pass would compute for this pair, as a key against key32[row]
                const float dot = __uint_as_float(qe[1]);
                const long p = (long)n0 + col;
                float d = ref_distance(dot, s_qsq[row], s_psq[col], s_qn[row], s_pn[col]);
                const bool isnan = d != d;
                if (masked && p >= mask_lo && p < mask_hi) d = 1.f;      // a masked pool row competes at exactly 1
                const unsigned hi = sortable(d), khi = s_khi[row];
                if (!(isnan || hi < khi || (hi == khi && (unsigned)p <= s_klo[row]))) continue;
                const int pos = atomicAdd(&s_n[0], 1);
                if (pos >= SCR_LIST) break;                          // list full: this entry and the rest after the flush
                const int lp = atomicAdd(&s_cnt[row], 1);
                s_list[pos * 3] = ((unsigned)row << 16) | (unsigned)lp;
                s_list[pos * 3 + 1] = (unsigned)p;
                s_list[pos * 3 + 2] = qe[1];
            }
            __syncthreads();
            { const int c = s_cnt[tid]; s_base[tid] = c ? atomicAdd(&cand_count[m0 + tid], c) : 0; }
            __syncthreads();
            const int n = s_n[0] < SCR_LIST ? s_n[0] : SCR_LIST;
            for (int x = tid; x < n; x += 256) {
                const unsigned rl = s_list[x * 3];
                const int row = (int)(rl >> 16), slot = s_base[row] + (int)(rl & 0xFFFFu);
                if (slot < cap) {
                    unsigned* dst = cand + ((long)(m0 + row) * cap + slot) * 2;
                    dst[0] = s_list[x * 3 + 1]; dst[1] = s_list[x * 3 + 2];
                } else capped = true;
            }
            __syncthreads();
            s_cnt[tid] = 0;
            if (tid == 0) s_n[0] = 0;
            if (!__syncthreads_or(e < qcnt)) break;
        }
        qcnt = 0;
    };
    auto push = [&](int row, int col, float accv) __attribute__((always_inline)) {
        unsigned* e = s_queue + ((qcnt * 256 + tid) << 1);           // qcnt < SCR_QD: the queues are drained whenever a lane
        e[0] = ((unsigned)row << 8) | (unsigned)col;                 // comes within one column's 32 elements of the depth
        e[1] = __float_as_uint(accv * out_scale);                    // the dot product (out_scale is a power of two: exact)
        ++qcnt;
    };
#pragma unroll
    for (int j = 0; j < QG::TN; ++j) {
        const int col = QG::acc_col(wave, lane, j);
        const long p = (long)n0 + col;
        const bool pv = p < np;
        const float v_pn = pv ? pn[p] : 0.f;
        const float v_b2 = pv ? EPS * psq[p] * inv_os : -__builtin_inff();       // columns past np: the bound becomes +inf
        const bool in_mask = masked && pv && p >= mask_lo && p < mask_hi;
#pragma unroll
        for (int i = 0; i < QG::TM; ++i)
#pragma unroll
            for (int r = 0; r < QG::NR; ++r) {
                const float bound = fmaf(ra1[i][r], v_pn, -(ra2[i][r] + v_b2));
                // (NaN anywhere: the comparison fails and the pair goes to the exact test, which keeps NaN distances;
                //  a masked pool row competes at exactly 1, whatever its dot product)
                if ((!(acc[i][j][r] <= bound) || in_mask) && pv) {
                    const int row = QG::acc_row(wave, lane, i, r);
                    if ((long)m0 + row < nq) push(row, col, acc[i][j][r]);
                }
            }
        if (__syncthreads_or(qcnt > SCR_QD - QG::TM * QG::NR)) drain_and_flush();
    }
    drain_and_flush();
    if (capped) atomicOr(overflow, 1);
    __syncthreads();                                   // the list is consumed: the stages take the next tile's operands
  }
}

// one wave per query row: exact distances of its candidates, ascending top-k (same keys / order as knn_select_kernel)
__global__ __launch_bounds__(256) void knn_refine_kernel(const int* __restrict__ cand_count, const unsigned* __restrict__ cand, int cap,
                                                        const float* __restrict__ qn, const float* __restrict__ qsq, long nq,
                                                        const float* __restrict__ pn, const float* __restrict__ psq, long np, int k, long idx_offset,
                                                        long mask_lo, long mask_hi, long* __restrict__ out_idx, float* __restrict__ out_dist,
                                                        int* nan_flag) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= nq) return;
    int cnt = cand_count[row];
    cnt = cnt < cap ? cnt : cap;
    const float v_qn = qn[row], v_qsq = qsq[row];
    const unsigned* cr = cand + row * (long)cap * 2;
    unsigned long long best = KEY_INF;                                  // lanes < k: the sorted list
    bool saw_nan = false;
    for (int base = 0; base < cnt; base += 32) {
        unsigned long long v = lane < 32 ? best : KEY_INF;
        // (p < np: when the screen overflowed, reserved slots may never have been written — the caller discards the result then,
        //  but nothing may be read through a garbage index)
        if (lane >= 32 && base + lane - 32 < cnt && (long)cr[(base + lane - 32) * 2] < np) {
            const unsigned p = cr[(base + lane - 32) * 2];
            const float dot = __uint_as_float(cr[(base + lane - 32) * 2 + 1]);
            float d = ref_distance(dot, v_qsq, psq[p], v_qn, pn[p]);
            if (d != d) saw_nan = true;
            if ((long)p >= mask_lo && (long)p < mask_hi) d = 1.f;
            if (d < __builtin_inff()) v = ((unsigned long long)sortable(d) << 32) | p;      // NaN / +inf never enter
        }
        v = wave_sort64(v, lane);
        best = lane < k ? v : KEY_INF;
    }
    if (__ballot(saw_nan)) { if (lane == 0) atomicOr(nan_flag, 1); }
    if (lane < k) {
        out_idx[row * k + lane] = (long)(unsigned)(best & 0xFFFFFFFFull) + idx_offset;
        out_dist[row * k + lane] = unsortable((unsigned)(best >> 32));
    }
}

int split_count(long nq, long np) {
    // The kernel needs 109 KB of LDS, so one block is resident per CU: aim for ONE wave of <= 256 blocks and let
    // every block walk as many 128-row pool tiles as possible — the first tile of a block pays a full top-32
    // build for each of its 128 query rows (about as long as the tile's MFMA work), later tiles only filter.
    const long qtiles = cdiv64(nq, 128), ptiles = cdiv64(np, 128);
    long s = 256 / qtiles;
    if (s < 1) s = 1;
    if (s > ptiles) s = ptiles;
    if (s > 4096) s = 4096;
    return (int)s;
}

}  // namespace

extern "C" int knnsvc_row_norms(const float* x, int64_t rows, int32_t dim, int32_t ldx, float* norm, float* sq,
                                float* max_slot, void* stream) {
    KN_REQUIRE(x && rows >= 0 && dim > 0 && ldx >= dim, "row_norms: bad arguments");
    if (rows == 0) return KNNSVC_OK;
    hipLaunchKernelGGL(row_norms_kernel, dim3((unsigned)cdiv64(rows, 4)), dim3(256), 0, (hipStream_t)stream,
                       x, (long)rows, dim, ldx, norm, sq, max_slot);
    return knnsvc_check_launch("row_norms");
}

extern "C" size_t knnsvc_knn_workspace_bytes(int64_t nq, int64_t np, int32_t k) {
    if (nq <= 0 || np <= 0 || k <= 0) return 0;
    return (size_t)split_count(nq, np) * (size_t)nq * (size_t)k * 8;
}

extern "C" int knnsvc_knn_topk(const float* q, const float* q_norm, const float* q_sq, int64_t nq,
                               const float* pool, const float* p_norm, const float* p_sq, int64_t np,
                               int32_t dim, int32_t k, int64_t idx_offset, int64_t mask_lo, int64_t mask_hi,
                               int64_t* out_idx, float* out_dist,
                               void* workspace, size_t workspace_bytes, int32_t* nan_flag, void* stream) {
    KN_REQUIRE(q && q_norm && q_sq && pool && p_norm && p_sq && out_idx && out_dist && nan_flag, "knn_topk: null pointer");
    KN_REQUIRE(nq > 0 && np > 0, "knn_topk: empty query or pool");
    KN_REQUIRE(k >= 1 && k <= KMAX, "knn_topk: k must be in 1..32");
    KN_REQUIRE(np >= k, "knn_topk: pool smaller than k (the reference's topk would raise)");
    KN_REQUIRE(dim > 0 && dim % 4 == 0, "knn_topk: dim must be a multiple of 4");
    KN_REQUIRE(((uintptr_t)q & 15) == 0 && ((uintptr_t)pool & 15) == 0, "knn_topk: q/pool must be 16-byte aligned");
    KN_REQUIRE(np < (1ll << 32), "knn_topk: pool rows must fit 32 bits");
    const int S = split_count(nq, np);
    const size_t need = (size_t)S * nq * k * 8;
    if (workspace_bytes < need || !workspace)
        return knnsvc_fail(KNNSVC_EWORKSPACE, "knn_topk: workspace %zu < %zu bytes", workspace_bytes, need);
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute((const void*)knn_tile_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                KNN_LDS_BYTES) != hipSuccess)
            return knnsvc_fail(KNNSVC_EHIP, "knn_topk: hipFuncSetAttribute failed");
        attr = true;
    }
    const long ptiles = cdiv64(np, 128);
    const long rows_per_split = cdiv64(ptiles, S) * 128;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((unsigned)cdiv64(nq, 128), (unsigned)S);
    hipLaunchKernelGGL(knn_tile_kernel, grid, dim3(NT), KNN_LDS_BYTES, st, q, q_norm, q_sq, (long)nq, pool, p_norm,
                       p_sq, (long)np, dim, k, rows_per_split, (long)mask_lo, (long)mask_hi, (unsigned long long*)workspace, nan_flag);
    int rc = knnsvc_check_launch("knn_tile");
    if (rc) return rc;
    hipLaunchKernelGGL(knn_merge_keys_kernel, dim3((unsigned)cdiv64(nq, 4)), dim3(256), 0, st,
                       (const unsigned long long*)workspace, S, (long)nq, k, (long)idx_offset, (long*)out_idx, out_dist);
    return knnsvc_check_launch("knn_merge_keys");
}

extern "C" int knnsvc_knn_merge(const float* part_dist, const int64_t* part_idx, int32_t parts, int64_t nq, int32_t k,
                                int64_t* out_idx, float* out_dist, void* stream) {
    KN_REQUIRE(part_dist && part_idx && out_idx && out_dist, "knn_merge: null pointer");
    KN_REQUIRE(parts >= 1 && nq > 0 && k >= 1 && k <= KMAX, "knn_merge: bad sizes");
    hipLaunchKernelGGL(knn_merge_pairs_kernel, dim3((unsigned)cdiv64(nq, 4)), dim3(256), 0, (hipStream_t)stream,
                       part_dist, (const long*)part_idx, parts, (long)nq, k, (long*)out_idx, out_dist);
    return knnsvc_check_launch("knn_merge_pairs");
}

extern "C" int knnsvc_knn_select(const float* dots, int64_t ld, const float* q_norm, const float* q_sq, int64_t nq,
                                 const float* p_norm, const float* p_sq, int64_t np, int32_t k, int64_t idx_offset,
                                 int64_t mask_lo, int64_t mask_hi,
                                 int64_t* out_idx, float* out_dist, int32_t* nan_flag, void* stream) {
    KN_REQUIRE(dots && q_norm && q_sq && p_norm && p_sq && out_idx && out_dist && nan_flag, "knn_select: null pointer");
    KN_REQUIRE(nq > 0 && np > 0 && ld >= np, "knn_select: empty query or pool, or ld < np");
    KN_REQUIRE(k >= 1 && k <= KMAX, "knn_select: k must be in 1..32");
    KN_REQUIRE(np >= k, "knn_select: pool smaller than k (the reference's topk would raise)");
    KN_REQUIRE(np < (1ll << 32) && nq <= 0x7FFFFFFFll, "knn_select: pool rows must fit 32 bits");
    // KNNSVC_KNN_SCREEN=0 evaluates the reference formula on every element (A/B and equivalence tests); read per call
    const char* e = getenv("KNNSVC_KNN_SCREEN");
    if (e && e[0] == '0')
        hipLaunchKernelGGL(knn_select_kernel<false>, dim3((unsigned)nq), dim3(256), 0, (hipStream_t)stream, dots, (long)ld, q_norm, q_sq,
                           (long)nq, p_norm, p_sq, (long)np, k, (long)idx_offset, (long)mask_lo, (long)mask_hi, (long*)out_idx, out_dist, nan_flag);
    else
        hipLaunchKernelGGL(knn_select_kernel<true>, dim3((unsigned)nq), dim3(256), 0, (hipStream_t)stream, dots, (long)ld, q_norm, q_sq,
                           (long)nq, p_norm, p_sq, (long)np, k, (long)idx_offset, (long)mask_lo, (long)mask_hi, (long*)out_idx, out_dist, nan_flag);
    return knnsvc_check_launch("knn_select");
}


extern "C" int knnsvc_knn_screen(const void* q_f16x2, const float* q_absmax, const float* q_norm, const float* q_sq, int64_t nq,
                                 const void* p_f16x2, const float* p_absmax, const float* p_norm, const float* p_sq, int64_t np,
                                 int32_t dim, const float* thr, const int64_t* thr_idx, int64_t mask_lo, int64_t mask_hi, int32_t* cand_count,
                                 void* cand, int32_t cap, int32_t* overflow_flag, int32_t max_blocks, void* stream) {
    KN_REQUIRE(q_f16x2 && q_absmax && q_norm && q_sq && p_f16x2 && p_absmax && p_norm && p_sq && thr && thr_idx && cand_count && cand && overflow_flag,
               "knn_screen: null pointer");
    KN_REQUIRE(nq > 0 && np > 0 && dim >= 32 && dim % 32 == 0 && cap > 0, "knn_screen: bad sizes (dim must be a multiple of 32)");
    KN_REQUIRE(nq * (long)dim * 4 < (1L << 30) && np * (long)dim * 4 < (1L << 30), "knn_screen: operands must stay below 1 GiB (chunk the call)");
    KN_REQUIRE(((uintptr_t)q_f16x2 & 15) == 0 && ((uintptr_t)p_f16x2 & 15) == 0, "knn_screen: 16-byte alignment");
    KN_REQUIRE(max_blocks >= 0, "knn_screen: max_blocks must be >= 0 (0 = one block per CU)");
    static_assert(QG::LDS_BYTES >= (256 * 10 + 4) * 4 + SCR_LIST * 12 + SCR_QD * 256 * 8, "epilogue state fits the operand stages");
    static_assert(SCR_QD > QG::TM * QG::NR, "a lane queue holds more than one column's elements");
    static bool attr = false;
    static int cus = 0;
    if (!attr) {
        if (hipFuncSetAttribute((const void*)knn_screen_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, QG::LDS_BYTES) != hipSuccess)
            return knnsvc_fail(KNNSVC_EHIP, "knn_screen: hipFuncSetAttribute failed");
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        attr = true;
    }
    const long ntiles = quad_order_ids(cdiv64(nq, 256), cdiv64(np, 256));
    KN_REQUIRE(ntiles < (1L << 31), "knn_screen: too many tiles (chunk the call)");
    long blocks = max_blocks > 0 && max_blocks < cus ? max_blocks : cus;
    blocks = blocks / 8 * 8 > 0 ? blocks / 8 * 8 : 8;
    if (blocks > ntiles) blocks = ntiles;              // (ntiles is a multiple of 8)
    hipLaunchKernelGGL(knn_screen_kernel, dim3((unsigned)blocks), dim3(256), QG::LDS_BYTES, (hipStream_t)stream,
                       (const float*)q_f16x2, q_absmax, q_norm, q_sq, (long)nq, (const unsigned short*)p_f16x2, p_absmax, p_norm, p_sq,
                       (long)np, dim, thr, (const long*)thr_idx, (long)mask_lo, (long)mask_hi, cand_count, (unsigned*)cand, cap, overflow_flag, (int)ntiles);
    return knnsvc_check_launch("knn_screen");
}

extern "C" int knnsvc_knn_refine(const int32_t* cand_count, const void* cand, int32_t cap, const float* q_norm, const float* q_sq,
                                 int64_t nq, const float* p_norm, const float* p_sq, int64_t np, int32_t k, int64_t idx_offset,
                                 int64_t mask_lo, int64_t mask_hi, int64_t* out_idx, float* out_dist, int32_t* nan_flag, void* stream) {
    KN_REQUIRE(cand_count && cand && q_norm && q_sq && p_norm && p_sq && out_idx && out_dist && nan_flag, "knn_refine: null pointer");
    KN_REQUIRE(nq > 0 && np > 0 && cap > 0 && k >= 1 && k <= KMAX, "knn_refine: bad sizes");
    hipLaunchKernelGGL(knn_refine_kernel, dim3((unsigned)cdiv64(nq, 4)), dim3(256), 0, (hipStream_t)stream, cand_count,
                       (const unsigned*)cand, cap, q_norm, q_sq, (long)nq, p_norm, p_sq, (long)np, k, (long)idx_offset, (long)mask_lo, (long)mask_hi,
                       (long*)out_idx, out_dist, nan_flag);
    return knnsvc_check_launch("knn_refine");
}
