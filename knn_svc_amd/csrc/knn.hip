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
//
// Round 5 — exact re-score.  Whatever produced the dot products (the f16x2 matrix-core loop: 96 fp32 roundings along K; the fp32
// MFMA tile), they only PICK candidates now: every selection kernel keeps a WIDE list per row — up to 64 keys, one per lane: the
// k best plus everything within KNN_GUARD of the k-th — and knn_rescore_kernel recomputes q.p for those rows in fp64, rounds ONCE
// to fp32, replays ref_distance and sorts by (distance bits, index).  The order a caller sees is then that of the reference's
// formula on correctly rounded dot products: it differs from the reference's own order only where the reference's BLAS
// rounding (<= 6.8e-7 on fixture g3c) decides, and no longer where this path's (1.22e-6) did.
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

// ---- wide lists: 64 keys in ascending order, one per lane (unused lanes: KEY_INF).  The k-th key + KNN_GUARD bounds what a list
// keeps; when more than 64 keys lie inside the guard band (thousands of identical pool rows) the 64 best by (distance, index) stay.
constexpr int KW = 64;
constexpr float KNN_GUARD = 4.0e-6f;      // > 3 x the largest |screening distance - exact distance| measured (1.22e-6, g3c): 2 x is the least that keeps every exact top-k member listed
__device__ __forceinline__ unsigned long long guard_key(unsigned long long kth) {
    if (kth == KEY_INF) return KEY_INF;
    const float d = unsortable((unsigned)(kth >> 32)) + KNN_GUARD;
    return ((unsigned long long)sortable(d) << 32) | 0xFFFFFFFFull;
}
__device__ __forceinline__ unsigned long long wide_thr(unsigned long long mine, int k) {       // a candidate can matter iff key <= this
    const unsigned long long g = guard_key(readlane64(mine, k - 1)), w = readlane64(mine, KW - 1);
    return g < w ? g : w;
}
// mine (ascending) U c (ascending) -> the 64 smallest, ascending: min(a_i, c_{63-i}) is bitonic and holds them; six
// compare-exchange stages sort it
__device__ __forceinline__ void wide_merge_sorted(unsigned long long& mine, unsigned long long c, int lane) {
    const unsigned long long r = __shfl(c, 63 - lane, 64);
    unsigned long long m = mine < r ? mine : r;
#pragma unroll
    for (int j = 32; j > 0; j >>= 1) {
        const unsigned long long o = __shfl_xor(m, j, 64);
        const unsigned long long mn = m < o ? m : o, mx = m < o ? o : m;
        m = ((lane & j) == 0) ? mn : mx;
    }
    mine = m;
}
__device__ __forceinline__ void wide_merge(unsigned long long& mine, unsigned long long cand, int lane) {      // cand: any order
    wide_merge_sorted(mine, wave_sort64(cand, lane), lane);
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
constexpr int KNN_LDS_BYTES = LDS_STAGE * 4 + 128 * KW * 8 /*wide lists*/ + 128 * 4 * 2;

__global__ __launch_bounds__(512) void knn_tile_kernel(
    const float* __restrict__ q, const float* __restrict__ qn, const float* __restrict__ qsq, long nq,
    const float* __restrict__ pool, const float* __restrict__ pn, const float* __restrict__ psq, long np,
    int dim, int k, long rows_per_split, long mask_lo, long mask_hi, unsigned long long* __restrict__ part, int* nan_flag) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* dist = lds;                                                       // [128][LDD] (aliases staging)
    unsigned long long* lists = (unsigned long long*)(lds + LDS_STAGE);      // [128][KW]: wide lists
    float* s_qn = (float*)(lists + 128 * KW);
    float* s_qsq = s_qn + 128;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long q0 = (long)blockIdx.x * 128;
    const long p_begin = (long)blockIdx.y * rows_per_split;
    const long p_end = p_begin + rows_per_split < np ? p_begin + rows_per_split : np;

    for (int i = tid; i < 128 * KW; i += NT) lists[i] = KEY_INF;
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

        // selection: wave w owns query rows 16w .. 16w+15; the row's wide list lives in LDS, one key per lane
        for (int rr = 0; rr < 16; ++rr) {
            const int row = wave * 16 + rr;
            if (q0 + row >= nq) break;
            unsigned long long* lst = lists + row * KW;
            unsigned long long mine = lst[lane];
            const unsigned long long thr = wide_thr(mine, k);
            const float d0 = dist[row * LDD + lane], d1 = dist[row * LDD + 64 + lane];
            // NaN / +inf never enter: their sortable bits are >= those of +inf
            const unsigned long long k0 = ((unsigned long long)sortable(d0) << 32) | (unsigned)(p0 + lane);
            const unsigned long long k1 = ((unsigned long long)sortable(d1) << 32) | (unsigned)(p0 + 64 + lane);
            const bool f0 = (d0 < __builtin_inff()) && k0 <= thr;
            const bool f1 = (d1 < __builtin_inff()) && k1 <= thr;
            const unsigned long long b0 = __ballot(f0), b1 = __ballot(f1);
            if ((b0 | b1) == 0ull) continue;
            if (b0) wide_merge(mine, f0 ? k0 : KEY_INF, lane);
            if (b1) wide_merge(mine, f1 ? k1 : KEY_INF, lane);
            lst[lane] = mine;
        }
        __syncthreads();      // dist tile is about to be overwritten by the next tile's staging
    }
    if (saw_nan) atomicOr(nan_flag, 1);

    // per-split wide lists -> workspace  part[split][q][KW]
    for (int i = tid; i < 128 * KW; i += NT) {
        const int row = i / KW, e = i - row * KW;
        if (q0 + row < nq) part[((long)blockIdx.y * nq + q0 + row) * KW + e] = lists[row * KW + e];
    }
}

// Top-k from a precomputed dot-product matrix (the two-kernel route: q.p^T comes from the emulated-fp32 GEMM of
// conv_gemm.hip, 3-4x the rate of the fp32 MFMA tile above; this kernel replays the reference's distance formula on it
// and selects).  One block per query row, 4 waves; wave w scans pool columns [128 w + 512 t, +128) with the same
// threshold filter + 64-lane bitonic merge as knn_tile_kernel, wave 0 then folds the four sorted lists.
template <bool SCREEN>
__global__ __launch_bounds__(256) void knn_select_kernel(
    const float* __restrict__ dots, long ld, const float* __restrict__ qn, const float* __restrict__ qsq, long nq,
    const float* __restrict__ pn, const float* __restrict__ psq, long np, int k,
    long mask_lo, long mask_hi, unsigned long long* __restrict__ wide_out, int* nan_flag) {
    __shared__ unsigned long long lists[4][KW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long row = blockIdx.x;
    const float* drow = dots + row * ld;
    const float v_qn = qn[row], v_qsq = qsq[row];
    // the wave's WIDE list lives in registers, one key per lane: the k best so far and everything within KNN_GUARD of the k-th
    // (wide_thr).  After the first tiles a surviving candidate is rare and single: it is inserted with list_insert; bursts (the
    // warm-up, when everything beats an empty list) go through wide_merge (one 64-lane sort + six merge stages per 64 columns).
    unsigned long long mine = KEY_INF, thr = KEY_INF;
    bool saw_nan = false, has_thr = false;      // has_thr: the list holds k real entries, thr_d bounds what can still matter
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
            // element can beat the current threshold.  approx = 1 - dot / (|q||p|) in three instructions; the formula's
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
            const bool f0 = (d0 < __builtin_inff()) && k0 <= thr;
            const bool f1 = (d1 < __builtin_inff()) && k1 <= thr;
            unsigned long long b0 = __ballot(f0), b1 = __ballot(f1);
            if ((b0 | b1) == 0ull) continue;
            const int total = __popcll(b0) + __popcll(b1);
            if (total <= 8) {
                while (b0) {
                    const int l = __builtin_ctzll(b0); b0 &= b0 - 1;
                    const unsigned long long key = readlane64(k0, l);
                    if (key <= thr) { list_insert(mine, key, lane, KW); thr = wide_thr(mine, k); }
                }
                while (b1) {
                    const int l = __builtin_ctzll(b1); b1 &= b1 - 1;
                    const unsigned long long key = readlane64(k1, l);
                    if (key <= thr) { list_insert(mine, key, lane, KW); thr = wide_thr(mine, k); }
                }
            } else {
                if (b0) wide_merge(mine, f0 ? k0 : KEY_INF, lane);
                if (b1) wide_merge(mine, f1 ? k1 : KEY_INF, lane);
                thr = wide_thr(mine, k);
            }
            has_thr = thr != KEY_INF;
            thr_d = unsortable((unsigned)(thr >> 32));
        }
    }
    if (saw_nan) atomicOr(nan_flag, 1);
    lists[wave][lane] = mine;
    __syncthreads();
    if (wave == 0) {
        unsigned long long best = lists[0][lane];
#pragma unroll
        for (int s = 1; s < 4; ++s) wide_merge_sorted(best, lists[s][lane], lane);
        if (best > wide_thr(best, k)) best = KEY_INF;           // beyond the guard band of the final k-th: cannot matter
        wide_out[row * KW + lane] = best;
    }
}

// one wave per query row: fold `parts` wide lists into one
__global__ __launch_bounds__(256) void knn_merge_keys_kernel(const unsigned long long* __restrict__ part,
                                                            int parts, long nq, int k, unsigned long long* __restrict__ wide_out) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= nq) return;
    unsigned long long best = part[row * KW + lane];
    for (int s = 1; s < parts; ++s) wide_merge_sorted(best, part[((long)s * nq + row) * KW + lane], lane);
    if (best > wide_thr(best, k)) best = KEY_INF;
    wide_out[row * KW + lane] = best;
}

// Exact re-score (round 5): one workgroup per query row, its wide list in -> ascending top-k out.  For every listed pool row the
// dot product is recomputed in fp64 from the fp32 operands (products exact, sums at 2^-53: the correctly rounded fp32 dot up
// to double-rounding ties), rounded once to fp32 and put through ref_distance: the distance the reference's formula gives on an
// exact matrix product.  Keys (distance bits, index) are re-sorted; ties keep the lower index.  exact = 0 (KNNSVC_KNN_RESCORE=0,
// A/B aid) passes the screening distances through.  The pass is a gather of ~33 rows of 4 KB per query row (200 MB at the
// north-star point): eight waves share the row's entries, four entries per wave in flight (DIM = 1024: every load of a
// group is issued before the first is used).
constexpr int RS_WAVES = 8, RS_FLIGHT = 4;
template <int DIM>
__global__ __launch_bounds__(RS_WAVES * 64) void knn_rescore_kernel(const unsigned long long* __restrict__ wide, long nq, int k,
                                                         const float* __restrict__ q, const float* __restrict__ qn,
                                                         const float* __restrict__ qsq, const float* __restrict__ pool,
                                                         const float* __restrict__ pn, const float* __restrict__ psq, int dim_rt,
                                                         long idx_offset, long mask_lo, long mask_hi, int exact,
                                                         long* __restrict__ out_idx, float* __restrict__ out_dist) {
    __shared__ unsigned long long s_keys[KW];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int dim = DIM ? DIM : dim_rt;
    // workgroup b runs on XCD b % 8: each XCD takes one contiguous eighth of the query rows, so the rows in flight on one L2 are
    // neighbours in time — on real (temporally smooth) features they list mostly the same pool rows
    const long rpx = (nq + 7) / 8;
    const long row = (long)(blockIdx.x & 7) * rpx + (blockIdx.x >> 3);
    if ((blockIdx.x >> 3) >= rpx || row >= nq) return;
    const unsigned long long key = wide[row * KW + lane];
    const int n = __popcll(__ballot(key != KEY_INF));            // a sorted list: its entries are the first n lanes
    if (exact) {
        const float* qr = q + row * (long)dim;
        const float v_qn = qn[row], v_qsq = qsq[row];
        const int idx = (int)(unsigned)(key & 0xFFFFFFFFull);
        for (int e0 = wave; e0 < n; e0 += RS_WAVES * RS_FLIGHT) {
            long pp[RS_FLIGHT];
            const float* rp[RS_FLIGHT];
            double acc[RS_FLIGHT][4];
#pragma unroll
            for (int f = 0; f < RS_FLIGHT; ++f) {
                const int e = e0 + f * RS_WAVES;
                pp[f] = (long)(unsigned)__builtin_amdgcn_readlane(idx, e < n ? e : e0);      // (past the list: the group's first row again)
                rp[f] = pool + pp[f] * dim;
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[f][t] = 0.0;
            }
#pragma unroll
            for (int c0 = 0; c0 < (DIM ? DIM : 1 << 30); c0 += 256) {
                const int c = c0 + lane * 4;
                if (!DIM && c0 >= dim) break;
                const bool in = DIM ? true : c < dim;
                const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                const f32x4 vq = in ? *(const f32x4*)(qr + c) : z;
                f32x4 vp[RS_FLIGHT];
#pragma unroll
                for (int f = 0; f < RS_FLIGHT; ++f) vp[f] = in ? *(const f32x4*)(rp[f] + c) : z;
#pragma unroll
                for (int f = 0; f < RS_FLIGHT; ++f)
#pragma unroll
                    for (int t = 0; t < 4; ++t) acc[f][t] = fma((double)vq[t], (double)vp[f][t], acc[f][t]);
            }
            double dots[RS_FLIGHT];
#pragma unroll
            for (int f = 0; f < RS_FLIGHT; ++f) dots[f] = wave_sum_d((acc[f][0] + acc[f][1]) + (acc[f][2] + acc[f][3]));
#pragma unroll
            for (int f = 0; f < RS_FLIGHT; ++f) {
                const int e = e0 + f * RS_WAVES;
                if (e >= n) break;                                       // (wave-uniform)
                const unsigned long long old = readlane64(key, e);
                if (lane == f) {
                    float d = ref_distance((float)dots[f], v_qsq, psq[pp[f]], v_qn, pn[pp[f]]);
                    if (pp[f] >= mask_lo && pp[f] < mask_hi) d = 1.f;
                    // (a listed pair has a finite screening distance; should the exact one not be finite — it cannot, from the same
                    //  finite operands — the pair keeps its screening key)
                    s_keys[e] = d < __builtin_inff() ? (((unsigned long long)sortable(d) << 32) | (unsigned)pp[f]) : old;
                }
            }
        }
        __syncthreads();
    }
    if (wave == 0) {
        unsigned long long best = key;
        if (exact) best = wave_sort64(lane < n ? s_keys[lane] : KEY_INF, lane);
        if (lane < k) {
            out_idx[row * k + lane] = (best == KEY_INF ? 0l : (long)(unsigned)(best & 0xFFFFFFFFull)) + idx_offset;      // unfilled (NaN row): a valid row, NaN distance
            out_dist[row * k + lane] = unsortable((unsigned)(best >> 32));
        }
    }
}
static void launch_rescore(const unsigned long long* wide, long nq, int k, const float* q, const float* qn, const float* qsq, const float* pool,
                           const float* pn, const float* psq, int dim, long idx_offset, long mask_lo, long mask_hi, int exact, long* out_idx,
                           float* out_dist, hipStream_t st) {
    if (dim == 1024)
        hipLaunchKernelGGL(knn_rescore_kernel<1024>, dim3((unsigned)(8 * ((nq + 7) / 8))), dim3(RS_WAVES * 64), 0, st, wide, nq, k, q, qn, qsq, pool, pn, psq, dim,
                           idx_offset, mask_lo, mask_hi, exact, out_idx, out_dist);
    else
        hipLaunchKernelGGL(knn_rescore_kernel<0>, dim3((unsigned)(8 * ((nq + 7) / 8))), dim3(RS_WAVES * 64), 0, st, wide, nq, k, q, qn, qsq, pool, pn, psq, dim,
                           idx_offset, mask_lo, mask_hi, exact, out_idx, out_dist);
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
        out_idx[row * k + lane] = best == KEY_INF ? 0l : (long)(unsigned)(best & 0xFFFFFFFFull);
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
    if ((dim & 3) == 0 && (ldx & 3) == 0 && ((uintptr_t)x & 15) == 0) {
        // 16-byte loads, four independent fp64 chains per lane (the scalar walk was one dependent chain of 4-byte loads: 3.6 TB/s)
        double s4[4] = {0.0, 0.0, 0.0, 0.0};
        for (int c = lane * 4; c < dim; c += 256) {
            const f32x4 v = *(const f32x4*)(xr + c);
#pragma unroll
            for (int e = 0; e < 4; ++e) s4[e] = fma((double)v[e], (double)v[e], s4[e]);
        }
        s = (s4[0] + s4[1]) + (s4[2] + s4[3]);
    } else {
        for (int c = lane; c < dim; c += 64) { const double v = xr[c]; s += v * v; }
    }
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
// Fused route (default from 256 query frames on: the north-star point, dataset mode, serving, cfg 5): no [Nq, Np] dot matrix.
// The pool's columns are walked in EPOCHS of growing size; every epoch is one persistent launch of knn_screen_kernel (q.p^T on
// the Gemm2QuadS main loop, each accumulator element screened IN REGISTERS against its row's threshold, the reference's distance
// formula evaluated on the few survivors, (pool index, distance bits) compacted per row into a candidate buffer) followed by
// knn_refine_kernel (the row's top-k so far + the new candidates -> top-k, and the k-th KEY as the next epoch's threshold).
//   epoch 0 (COLD = true): no threshold exists yet, so each tile bounds its rows itself.  Per row and column half (one wave: 16
//           lanes x 8 columns) the 16th largest of the 32 maxima over groups of 4 columns of dot / |p| is bracketed by bisection:
//           at least 16 + 16 distinct columns of the tile lie at or above the smaller of the two halves' bounds, so the row's true
//           top-k cannot lie beyond it (plus the screen's rounding margin).  About 40 of a tile's 256 columns pass per row;
//   epoch e > 0: the threshold is the exact k-th (distance, index) KEY over everything seen so far: exactly {key <= key_k}
//           passes — its size is the rank of that key among the new columns (about k x new / seen), however many
//           near-identical rows the pool holds.
// Round 3 took its threshold from a separate pass over a strided sample of the pool (a GEMM + a selection: 0.15 ms whatever the
// size, 1 ms at 24 000 x 180 000) and lost to the dot-matrix route below 2048 query frames; here the first epoch IS part of the
// search.  The candidate set of every epoch is a superset of the true top-k among its columns, and the distances are computed by
// ref_distance from the very dot-product bits of the dot-matrix route, so the result is identical to it — indices and distance
// bits (tests: fused == dot matrix, sharded == unsharded, grouped == ungrouped).
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

// Persistent: block b walks tiles b, b + gridDim.x, ... in the XCD-aware order of gemm2_core.h (for grids that order pads —
// quad_order_ids a multiple of 8 — and gridDim.x a multiple of 8, a block's tiles stay on its XCD; small grids use the plain order
// and make no such promise).  The host caps the grid (`max_blocks`): inside a stream pipeline the search then leaves CUs to the
// single-workgroup recurrences and the generator of the other items instead of occupying every CU with a 128 KB-LDS block.
// p2 / pn / psq point at the epoch's first pool row, np = its rows, p_base = that row's index in the chunk (candidate indices, the
// mask range and thr_idx live in the chunk's index space).
__device__ __forceinline__ float knn_row_fold_min(float v) {       // all 16 lanes of a DPP row get the row's minimum
#define KN_F(CTRL) v = fminf(v, __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true)));
    KN_F(0xB1) KN_F(0x4E) KN_F(0x141) KN_F(0x140)
#undef KN_F
    return v;
}
__device__ __forceinline__ float knn_row_fold_max(float v) {
#define KN_F(CTRL) v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true)));
    KN_F(0xB1) KN_F(0x4E) KN_F(0x141) KN_F(0x140)
#undef KN_F
    return v;
}
__device__ __forceinline__ int knn_row_fold_add(int v) {
#define KN_F(CTRL) v += __builtin_amdgcn_mov_dpp(v, CTRL, 0xF, 0xF, true);
    KN_F(0xB1) KN_F(0x4E) KN_F(0x141) KN_F(0x140)
#undef KN_F
    return v;
}
#ifdef KN_KNN_PROF        // timing aid (tools/knn_prof.py): per block and tile, 10 ns ticks: start / main loop done / cold bound done / coarse pass
__device__ long long kn_knn_prof_buf[4096 * 8];        // + drains done / tile done, survivors pushed by thread 0's wave, drains
#define KN_KP(SLOT, VAL) { if (threadIdx.x == 0 && kp_tile < 4096) kn_knn_prof_buf[kp_tile * 8 + (SLOT)] = (long long)(VAL); }
#define KN_KP_T(SLOT) KN_KP(SLOT, __builtin_amdgcn_s_memrealtime())
#else
#define KN_KP(SLOT, VAL)
#define KN_KP_T(SLOT)
#endif
constexpr int XCH_MAX = 88;                          // half-tile bounds a row reads in the first epoch's exchange (2 x 44 column tiles)
constexpr int COLD_STEPS_XCH = 2;                    // ... when the tiles also exchange bounds (the tile's own bound is the backstop then)
constexpr int COLD_STEPS = 5;                        // bisection steps of the cold bound (each halves the bracket of the 16th largest)

template <bool COLD, bool XCH>
__global__ __launch_bounds__(256, 1) void knn_screen_kernel(
    const float* __restrict__ q2, const float* __restrict__ q_absmax, const float* __restrict__ qn, const float* __restrict__ qsq, long nq,
    const unsigned short* __restrict__ p2, const float* __restrict__ p_absmax, const float* __restrict__ pn, const float* __restrict__ psq,
    long np, int dim, const float* __restrict__ thr, const long* __restrict__ thr_idx, long mask_lo, long mask_hi, long p_base,
    int* __restrict__ cand_count, unsigned* __restrict__ cand, int cap, unsigned* __restrict__ cold_ws, int* __restrict__ overflow,
    int ntiles) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int gx = (int)((nq + 255) / 256), gy = (int)((np + 255) / 256);
    const float out_scale = 1.0f / (knn_pick_scale(q_absmax) * knn_pick_scale(p_absmax));
  for (int vt = blockIdx.x; vt < ntiles; vt += gridDim.x) {
    int mt, nt;
    if (!quad_order_decode(vt, gx, gy, mt, nt)) continue;          // XCD-aware order, padding ids (gemm2_core.h)
    const int m0 = mt * 256, n0 = nt * 256;
#ifdef KN_KNN_PROF
    const int kp_tile = vt;
    int kp_drains = 0, kp_pushed = 0;
#endif
    KN_KP_T(0)

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
    KN_KP_T(1)

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
    unsigned* s_khi = (unsigned*)(s_qsq + 256);        // [256] the row's threshold KEY: distance bits, pool index
    unsigned* s_klo = s_khi + 256;
    float* s_pn = (float*)(s_klo + 256);               // [256] |p|, |p|^2 of the tile's pool rows: the exact formula's column operands
    float* s_psq = s_pn + 256;
    int* s_cnt = (int*)(s_psq + 256);                  // [256] entries per row in the current list
    int* s_base = s_cnt + 256;                         // [256] their first slot in the row's global candidate list
    int* s_n = s_base + 256;                           // [1] entries in the tile list
    unsigned* s_list = (unsigned*)(s_n + 4);           // [SCR_LIST][3]: row << 16 | position in row, pool index, distance bits
    unsigned* s_queue = s_list + SCR_LIST * 3;         // [SCR_QD][256][2]: lane queues: row << 8 | column, dot bits
    float* s_T = (float*)s_queue;                      // COLD: [2][256] per column half, [8] |p| range (the queues are still empty)
    // (laundered: derived from a plain threadIdx.x, the epilogue's per-lane rows / columns / LDS addresses are loop-invariant,
    //  get hoisted out of the tile loop and then live — and spill — across the main loop, which has no register to spare)
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63, wave = tid >> 6;
    constexpr float EPS = 4.5e-6f;
    const float inv_os = 1.0f / out_scale;             // powers of two: exact
    const bool masked = mask_lo < mask_hi;
    const long pc0 = p_base + n0;                      // chunk index of the tile's first pool row
    // `t` = a distance the row's k-th best cannot exceed -> the coarse test's row constants and the exact test's key
    auto set_row = [&](float t, unsigned klo_v) __attribute__((always_inline)) {
        const long r = (long)m0 + tid;
        const bool v = r < nq;
        const float n_ = s_qn[tid], sq_ = s_qsq[tid];
        s_a1[tid] = v ? n_ * ((1.0f - t) - EPS) * inv_os : 3.0e38f;        // rows past nq: nothing survives
        s_a2[tid] = v ? EPS * sq_ * inv_os : 0.f;
        // a NaN / +inf threshold (a NaN query row, a tile that cannot bound the row) keeps every pair: the refine pass sees them all
        const bool open = !(t < __builtin_inff());
        s_khi[tid] = open ? 0xFFFFFFFFu : sortable(t);
        s_klo[tid] = open ? 0xFFFFFFFFu : klo_v;
    };
    {
        const long r = (long)m0 + tid;
        const bool v = r < nq;
        s_qn[tid] = v ? qn[r] : 0.f; s_qsq[tid] = v ? qsq[r] : 0.f;
        s_cnt[tid] = 0;
        if (tid == 0) { s_n[0] = 0; s_n[1] = 0; }
        const long pc = (long)n0 + tid;
        s_pn[tid] = pc < np ? pn[pc] : 0.f;
        s_psq[tid] = pc < np ? psq[pc] : 0.f;
        if (!COLD) set_row(v ? thr[r] : 0.f, (unsigned)(v ? thr_idx[r] : 0));
    }
    __syncthreads();
    if (COLD) {
        // ---- no threshold yet: the tile bounds its rows itself.  sim = acc / |p| orders a row's columns by approximate cosine
        // (|q| and the operand scales are row constants).  Each lane takes the maxima of its two groups of 4 columns per row; a
        // row's 16 lanes of THIS wave then bracket the 16th largest of their 32 group maxima by bisection (DPP row folds, no LDS):
        // lo always has >= 16 group maxima — 16 distinct columns — at or above it.  Both column halves together: 32 distinct
        // columns with sim >= min(lo_0, lo_1), so the row's k-th best (k <= 32) approximate distance is at most
        // t = 1 - that / |q|, its exact one at most t + g (g: knn_select_kernel's margin between the formula and the approximation,
        // bounded over the tile's |p| range; doubled for this form's own roundings).  Masked columns and columns past np are
        // no witnesses (sim = -inf); they are still pushed by the coarse pass below when they have to be.
        float rp[QG::TN], bias[QG::TN];
        float pmin = __builtin_inff(), pmax = 0.f;
#pragma unroll
        for (int j = 0; j < QG::TN; ++j) {
            const int col = QG::acc_col(wave, lane, j);
            const long p = pc0 + col;
            const float v_pn = s_pn[col];
            const bool ok = (long)n0 + col < np && v_pn > 0.f && !(masked && p >= mask_lo && p < mask_hi);
            rp[j] = ok ? __builtin_amdgcn_rcpf(v_pn) : 0.f;
            bias[j] = ok ? 0.f : -__builtin_inff();
            if (ok) { pmin = fminf(pmin, v_pn); pmax = fmaxf(pmax, v_pn); }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { pmin = fminf(pmin, __shfl_xor(pmin, o, 64)); pmax = fmaxf(pmax, __shfl_xor(pmax, o, 64)); }
        if (lane == 0) { s_T[512 + wave * 2] = pmin; s_T[512 + wave * 2 + 1] = pmax; }
#pragma unroll
        for (int i = 0; i < QG::TM; ++i)
#pragma unroll
            for (int r = 0; r < QG::NR; ++r) {
                float ga = -__builtin_inff(), gb = -__builtin_inff();
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    ga = fmaxf(ga, fmaf(acc[i][j][r], rp[j], bias[j]));
                    gb = fmaxf(gb, fmaf(acc[i][j + 4][r], rp[j + 4], bias[j + 4]));
                }
                // bracket of the 16th largest T* of the row's 32 group maxima: u = the smallest of the 16 lane maxima (16 values at or
                // above it), v = the largest of the lanes' SMALLER group maxima — only lane maxima lie above v, so either fewer than 16
                // values do (T* <= v) or all 16 lane maxima do (T* = u): T* in [u, max(u, v)].  Both ends sit inside the bulk of the
                // row's values; the row maximum (an outlier whenever the tile holds a true neighbour) would stretch the bracket
                // over the whole range and leave the bisection's resolution coarser than the bulk is wide (everything passes)
                float lo = knn_row_fold_min(fmaxf(ga, gb));
                float hi = fmaxf(lo, knn_row_fold_max(fminf(ga, gb)));
#pragma unroll
                for (int it = 0; it < (XCH ? COLD_STEPS_XCH : COLD_STEPS); ++it) {
                    const float mid = 0.5f * lo + 0.5f * hi;
                    const int c = knn_row_fold_add((ga >= mid ? 1 : 0) + (gb >= mid ? 1 : 0));
                    if (c >= 16) lo = mid; else hi = mid;
                }
                const float hb = knn_row_fold_max(fmaxf(ga, gb));                 // this half's best column of the row
                if ((lane & 15) == 0) {
                    const int row = QG::acc_row(wave, lane, i, r);
                    s_T[(wave & 1) * 256 + row] = lo;
                    s_T[520 + (wave & 1) * 256 + row] = hb;
                }
            }
        __syncthreads();
        KN_KP_T(7)
        {
            const float T = fminf(s_T[tid], s_T[256 + tid]);                       // (-inf: a half without 16 witnesses)
            const float lo_p = fminf(fminf(s_T[512], s_T[514]), fminf(s_T[516], s_T[518]));
            const float hi_p = fmaxf(fmaxf(s_T[513], s_T[515]), fmaxf(s_T[517], s_T[519]));
            const float n_ = s_qn[tid];
            // approximate distance of the weakest witness, then the margin  3.8e-6 (1 + |q|/|p| + |p|/|q|)  at its largest, twice
            const float t = 1.0f - T * out_scale / n_;
            const float g = 7.6e-6f * (1.0f + n_ / lo_p + hi_p / n_);
            const float te = t + g;
            const bool fin = n_ > 0.f && T > -__builtin_inff() && te < __builtin_inff();
            float tb = fin ? te : __builtin_inff();
            // ---- exchange between the tiles of this row tile (they run side by side in the first epoch).  One tile's 256 columns
            // cannot bound a row below their own 32nd best — the 12 % quantile of the pool; the tiles TOGETHER can: every column
            // half publishes the bound of its single best column per row, and once 32 such bounds of distinct columns are in, the
            // 32nd smallest of them bounds the row's k-th best over the whole epoch (the 0.5 % quantile for neighbours spread over
            // the pool; neighbours packed into one tile are what the tile's own bound handles).  Only tightness depends on who
            // has published when: a word that is still zero reads as "no bound".  The wait for the other tiles is bounded (8 us).
            if (cold_ws && XCH) {
                unsigned* best = cold_ws + nq;                                     // [2 gy][nq], zero-filled by the caller
                unsigned* arrived = cold_ws + nq * (long)(1 + 2 * gy) + (long)mt * 32;     // [gx] counters, one cache line each
                const long r = (long)m0 + tid;
                if (r < nq && n_ > 0.f) {
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const float B = s_T[520 + h * 256 + tid];
                        const float b = (1.0f - B * out_scale / n_) + g;
                        if (B > -__builtin_inff() && b < __builtin_inff())
                            __hip_atomic_store(&best[(long)(2 * nt + h) * nq + r], ~sortable(b), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
                // (the bounds went out as device-scope stores; this waits for their acknowledgement — no L2 write-back as a device-scope
                //  fence would do: a reader that sees the counter before a bound only gets a weaker bound)
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __syncthreads();
                if (tid == 0) {
                    atomicAdd(arrived, 1u);
                    const long long t_end = (long long)__builtin_amdgcn_s_memrealtime() + 800;        // 100 MHz ticks
                    while (__hip_atomic_load(arrived, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)gy &&
                           (long long)__builtin_amdgcn_s_memrealtime() < t_end)
                        __builtin_amdgcn_s_sleep(24);
                }
                __syncthreads();
                KN_KP_T(5)
                if (r < nq && n_ > 0.f) {
                    float v[XCH_MAX];
                    unsigned w[XCH_MAX];
                    int n_pub = 0;
                    float vmin = __builtin_inff(), vmax = -__builtin_inff();
                    // (plain loads through this XCD's L2: the words were stored through to memory, and nothing of this launch has read
                    //  their lines on this XCD earlier than the tiles that wait here together — a stale zero would only mean a weaker
                    //  bound.  All of them in flight at once, planes past 2 gy re-read plane 0: behind a branch each, the compiler
                    //  waited for every load before issuing the next — 24 us.)
#pragma unroll
                    for (int x = 0; x < XCH_MAX; ++x) w[x] = best[(long)(x < 2 * gy ? x : 0) * nq + r];
#pragma unroll
                    for (int x = 0; x < XCH_MAX; ++x) {
                        const bool pub = w[x] != 0u && x < 2 * gy;
                        v[x] = pub ? unsortable(~w[x]) : __builtin_inff();
                        if (pub) { ++n_pub; vmin = fminf(vmin, v[x]); vmax = fmaxf(vmax, v[x]); }
                    }
#ifdef KN_KNN_PROF_LOADS
                    KN_KP_T(6)
#endif
                    if (n_pub >= 32) {              // 32nd smallest, from above: `hi` always has >= 32 published bounds at or below it
                        float lo = vmin, hi = vmax;
#pragma unroll 1
                        for (int it = 0; it < 10; ++it) {
                            const float mid = 0.5f * lo + 0.5f * hi;
                            int c = 0;
#pragma unroll
                            for (int x = 0; x < XCH_MAX; ++x) c += v[x] <= mid ? 1 : 0;
                            if (c >= 32) hi = mid; else lo = mid;
                        }
                        tb = fminf(tb, hi);
                    }
                }
            }
#ifndef KN_KNN_PROF_LOADS
            KN_KP_T(6)
#endif
            tb += KNN_GUARD;                 // the wide lists keep everything within the guard band of the k-th (exact re-score)
            set_row(tb, 0xFFFFFFFFu);
            // the best bound found for the row (stored inverted: atomicMax over a zero-filled word): the refine pass starts from it
            // and drops the survivors of tiles that had to work with weaker bounds unseen
            if (tb < __builtin_inff() && cold_ws && (long)m0 + tid < nq) atomicMax(&cold_ws[m0 + tid], ~sortable(tb));
        }
        __syncthreads();
    }
    KN_KP_T(2)
    // The coarse test per element:  acc > a1[row] |p| - (a2[row] + b2[col]).  a2 = eps' |q|^2 / out_scale is the margin's row part: the
    // LARGEST one of the wave's 128 rows serves them all (a looser bound for the rows with smaller norms, by a fraction of an
    // already tiny margin — a few more survivors, never fewer), so (a2 + b2) is one value per column and an element costs one
    // fma and one compare instead of an add, an fma and a compare (the pass runs at one wave per SIMD: issue-bound).
    float ra1[QG::TM][QG::NR];
#pragma unroll
    for (int i = 0; i < QG::TM; ++i)
#pragma unroll
        for (int r = 0; r < QG::NR; ++r) ra1[i][r] = s_a1[QG::acc_row(wave, lane, i, r)];
    float a2max = fmaxf(s_a2[(wave / QG::WN) * 128 + lane], s_a2[(wave / QG::WN) * 128 + 64 + lane]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) a2max = fmaxf(a2max, __shfl_xor(a2max, o, 64));
    const int rows_left = (int)((nq - m0) < 256 ? (nq - m0) : 256);       // rows of this tile that exist (32-bit compares below)
    int qcnt = 0;
    bool capped = false;
    // all lanes: queues -> tile list -> global slots, repeated until every queue is empty (block-uniform control flow)
    auto drain_and_flush = [&]() __attribute__((always_inline)) {
        int e = 0;
        for (;;) {
            // one queue entry per lane and trip; the wave's survivors of a trip take their places in the tile list with ONE LDS
            // atomic (a cold tile lets ~10 000 pairs through: one atomic each on a single address serialised the whole block —
            // 125 us per tile; warm tiles have a handful)
            bool stalled = false;
            for (;;) {
                const bool have = e < qcnt && !stalled;
                if (__ballot(have) == 0ull) break;
                bool pass = false;
                unsigned hi = 0, p32 = 0;
                int row = 0;
                if (have) {
                    const unsigned* qe = s_queue + ((e * 256 + tid) << 1);
                    const unsigned code = qe[0];
                    row = (int)(code >> 8);
                    const int col = (int)(code & 255u);
                    // the EXACT test: the reference's distance of this pair — from this very dot product — as a key against the row's
                    const float dot = __uint_as_float(qe[1]);
                    const long p = pc0 + col;
                    float d = ref_distance(dot, s_qsq[row], s_psq[col], s_qn[row], s_pn[col]);
                    const bool isnan = d != d;
                    if (masked && p >= mask_lo && p < mask_hi) d = 1.f;      // a masked pool row competes at exactly 1
                    hi = isnan ? 0xFFFFFFFFu : sortable(d);                   // (NaN: kept, the refine pass reports it)
                    p32 = (unsigned)p;
                    const unsigned khi = s_khi[row];
                    pass = hi < khi || (hi == khi && p32 <= s_klo[row]) || isnan;
                }
                const unsigned long long bm = __ballot(pass);
                if (bm) {
                    const int first = __builtin_ctzll(bm);
                    int base = 0;
                    if (lane == first) base = atomicAdd(&s_n[0], __popcll(bm));
                    base = __builtin_amdgcn_readlane(base, first);
                    const int pos = base + __popcll(bm & ((1ull << lane) - 1ull));
                    if (pass) {
                        if (pos < SCR_LIST) {
                            const int lp = atomicAdd(&s_cnt[row], 1);
                            s_list[pos * 3] = ((unsigned)row << 16) | (unsigned)lp;
                            s_list[pos * 3 + 1] = p32;
                            s_list[pos * 3 + 2] = hi;
                        } else stalled = true;                        // list full: this entry and the rest after the flush
                    }
                }
                if (have && !stalled) ++e;
            }
            __syncthreads();
            if (tid == 0) s_n[1] = 0;                                // (everybody has read the drain request by now)
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
#ifdef KN_KNN_PROF
        ++kp_pushed;
#endif
    };
#pragma unroll
    for (int j = 0; j < QG::TN; ++j) {
        const int col = QG::acc_col(wave, lane, j);
        const long p = pc0 + col;
        const bool pv = (long)n0 + col < np;
        const float v_pn = s_pn[col];
        const bool in_mask = masked && pv && p >= mask_lo && p < mask_hi;
        // columns past np: the bound becomes +inf (nothing passes); a masked column: NaN (every row passes: it competes at exactly 1
        // whatever its dot product) — both without a term in the per-element test
        const float v_nb2 = !pv ? __builtin_inff() : in_mask ? __builtin_nanf("") : -(a2max + EPS * s_psq[col] * inv_os);
#pragma unroll
        for (int i = 0; i < QG::TM; ++i) {
            // (NaN anywhere: the comparison fails and the pair goes to the exact test, which keeps NaN distances; a masked pool row
            //  competes at exactly 1, whatever its dot product.)  The four rows a lane holds of one 16 x 16 tile share ONE branch: the
            //  four compare masks are OR-ed on the scalar unit — a survivor is rare, the exec-mask round trip per element was not
            bool c[QG::NR];
#pragma unroll
            for (int r = 0; r < QG::NR; ++r) c[r] = !(acc[i][j][r] <= fmaf(ra1[i][r], v_pn, v_nb2));
            if ((c[0] || c[1] || c[2] || c[3]) && pv) {
#pragma unroll
                for (int r = 0; r < QG::NR; ++r) {
                    const int row = QG::acc_row(wave, lane, i, r);
                    if (c[r] && row < rows_left) push(row, col, acc[i][j][r]);
                }
            }
        }
        // (a rolled loop over the groups with the elements behind a switch would hold the drain code once instead of eight times —
        //  tried: the register allocator answers with 600-700 spilled registers)
#ifdef KN_WHATIF_NOCHECK
        continue;
#endif
        // "does any lane's queue come within one column group of its depth?" with ONE barrier: a wave that needs the drain raises a
        // flag in LDS (plain store, rare), everybody reads it behind the barrier; drain_and_flush lowers it again behind its own first
        // barrier.  (__syncthreads_or is three barriers and an LDS atomic: 24 barriers per tile for a question whose answer is no.)
        if (__ballot(qcnt > SCR_QD - QG::TM * QG::NR) != 0ull && lane == 0) s_n[1] = 1;
        __syncthreads();
        if (__builtin_expect(s_n[1] != 0, 0)) {                     // (unlikely: laid out behind the hot path)
#ifdef KN_KNN_PROF
            ++kp_drains;
#endif
            drain_and_flush();
        }
    }
    KN_KP_T(3)
    drain_and_flush();
    KN_KP_T(4)
#ifdef KN_KNN_PROF
    if (!COLD) { KN_KP(5, kp_pushed) KN_KP(6, kp_drains) }
#endif
    if (capped) atomicOr(overflow, 2);             // bit 1 of the search's flag: the candidate buffer overflowed
    __syncthreads();                                   // the list is consumed: the stages take the next tile's operands
  }
}

// One wave per query row: the row's WIDE list so far (an earlier epoch's output) + the new candidates -> wide list (the k best and
// everything within KNN_GUARD of the k-th, ascending, same keys as knn_select_kernel); the list's threshold key (wide_thr) goes
// out as the next epoch's threshold and the row's candidate count is reset.  A candidate is (pool index, distance bits): the
// screen evaluated the reference's formula; 0xFFFFFFFF marks a NaN distance.  Candidates are filtered against the threshold 256
// at a time; a survivor is rare once the list is warm and is inserted by ballot + lane shift, bursts go through wide_merge.
__global__ __launch_bounds__(256) void knn_refine_kernel(int* __restrict__ cand_count, const unsigned* __restrict__ cand, int cap, long nq,
                                                        int k, const unsigned* __restrict__ row_bound, int has_prev,
                                                        unsigned long long* __restrict__ wide, float* __restrict__ thr_out,
                                                        long* __restrict__ thr_idx_out, int final_pass, int* flags) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long row = (long)blockIdx.x * 4 + wave;
    if (row >= nq) return;
    int cnt = cand_count[row];
    cnt = cnt < cap ? cnt : cap;
    const unsigned long long* cr = (const unsigned long long*)(cand + row * (long)cap * 2);      // little endian: index | bits << 32
    unsigned long long mine = has_prev ? wide[row * KW + lane] : KEY_INF;      // (sorted, its unused tail is KEY_INF)
    bool saw_nan = false;
    // first epoch: nothing that matters lies beyond the best bound a tile derived for the row (guard band included; ties at the
    // bound pass: index = max)
    unsigned long long bound_key = KEY_INF;
    if (row_bound) { const unsigned b = ~row_bound[row]; if (b != 0xFFFFFFFFu) bound_key = ((unsigned long long)b << 32) | 0xFFFFFFFFull; }
#define KN_THR() ({ const unsigned long long t_ = wide_thr(mine, k); t_ < bound_key ? t_ : bound_key; })
    unsigned long long thr = KN_THR();
    constexpr unsigned INF_BITS = 0xFF800000u;                          // sortable(+inf): NaN / +inf never enter
    for (int base = 0; base < cnt; base += 256) {
        unsigned long long kv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int x = base + u * 64 + lane;
            kv[u] = KEY_INF;
            if (x < cnt) {
                const unsigned long long raw = cr[x];
                const unsigned p = (unsigned)raw, hi = (unsigned)(raw >> 32);
                if (hi == 0xFFFFFFFFu) saw_nan = true;
                else if (hi < INF_BITS) kv[u] = ((unsigned long long)hi << 32) | p;
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (base + u * 64 >= cnt) break;
            const bool f = kv[u] <= thr && kv[u] != KEY_INF;
            unsigned long long b = __ballot(f);
            if (b == 0ull) continue;
            if (__popcll(b) <= 8) {
                while (b) {
                    const int l = __builtin_ctzll(b); b &= b - 1;
                    const unsigned long long key = readlane64(kv[u], l);
                    if (key <= thr) { list_insert(mine, key, lane, KW); thr = KN_THR(); }
                }
            } else {
                wide_merge(mine, f ? kv[u] : KEY_INF, lane);
                thr = KN_THR();
            }
        }
    }
#undef KN_THR
    thr = wide_thr(mine, k);
    if (mine > thr) mine = KEY_INF;                                     // beyond the guard band of the k-th: cannot matter any more
    const bool nan_row = __ballot(saw_nan) != 0ull;
    const bool is_short = readlane64(mine, k - 1) == KEY_INF;
    if (lane == 0) {
        cand_count[row] = 0;                                            // the next epoch starts an empty list
        int fl = nan_row ? 1 : 0;
        // fewer than k entries at the end without a NaN in sight: the candidate set was NOT a superset of the top-k — cannot happen
        // by construction; if it ever does, the caller repeats the search on the dot-matrix route instead of using a short list
        if (final_pass && is_short && !nan_row) fl |= 2;
        if (fl) atomicOr(flags, fl);
        if (thr_out) {
            thr_out[row] = thr == KEY_INF ? __builtin_inff() : unsortable((unsigned)(thr >> 32));
            thr_idx_out[row] = (long)(unsigned)(thr & 0xFFFFFFFFull);
        }
    }
    wide[row * KW + lane] = mine;
}

#ifdef KN_KNN_PROF
}  // namespace
extern "C" int knnsvc_debug_knn_prof(long long* host, int n_tiles) {
    if (n_tiles > 4096) n_tiles = 4096;
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(kn_knn_prof_buf), (size_t)n_tiles * 8 * sizeof(long long)) == hipSuccess ? 0 : 3;
}
namespace {
#endif
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
    return ((size_t)split_count(nq, np) + 1) * (size_t)nq * (size_t)KW * 8;       // per-split wide lists + the merged one
}

extern "C" int knnsvc_knn_topk(const float* q, const float* q_norm, const float* q_sq, int64_t nq,
                               const float* pool, const float* p_norm, const float* p_sq, int64_t np,
                               int32_t dim, int32_t k, int64_t idx_offset, int64_t mask_lo, int64_t mask_hi,
                               int64_t* out_idx, float* out_dist,
                               void* workspace, size_t workspace_bytes, int32_t* nan_flag, int32_t rescore, void* stream) {
    KN_REQUIRE(q && q_norm && q_sq && pool && p_norm && p_sq && out_idx && out_dist && nan_flag, "knn_topk: null pointer");
    KN_REQUIRE(nq > 0 && np > 0, "knn_topk: empty query or pool");
    KN_REQUIRE(k >= 1 && k <= KMAX, "knn_topk: k must be in 1..32");
    KN_REQUIRE(np >= k, "knn_topk: pool smaller than k (the reference's topk would raise)");
    KN_REQUIRE(dim > 0 && dim % 4 == 0, "knn_topk: dim must be a multiple of 4");
    KN_REQUIRE(((uintptr_t)q & 15) == 0 && ((uintptr_t)pool & 15) == 0, "knn_topk: q/pool must be 16-byte aligned");
    KN_REQUIRE(np < (1ll << 32), "knn_topk: pool rows must fit 32 bits");
    const int S = split_count(nq, np);
    const size_t need = ((size_t)S + 1) * nq * KW * 8;
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
    unsigned long long* wide = (unsigned long long*)workspace + (size_t)S * nq * KW;
    hipLaunchKernelGGL(knn_merge_keys_kernel, dim3((unsigned)cdiv64(nq, 4)), dim3(256), 0, st,
                       (const unsigned long long*)workspace, S, (long)nq, k, wide);
    rc = knnsvc_check_launch("knn_merge_keys");
    if (rc) return rc;
    launch_rescore((const unsigned long long*)wide, (long)nq, k, q, q_norm, q_sq, pool, p_norm, p_sq, dim, (long)idx_offset, (long)mask_lo,
                   (long)mask_hi, rescore ? 1 : 0, (long*)out_idx, out_dist, st);
    return knnsvc_check_launch("knn_rescore");
}

extern "C" int knnsvc_knn_rescore(const void* wide, int64_t nq, int32_t k, const float* q, const float* q_norm, const float* q_sq,
                                  const float* pool, const float* p_norm, const float* p_sq, int64_t np, int32_t dim,
                                  int64_t idx_offset, int64_t mask_lo, int64_t mask_hi, int32_t exact,
                                  int64_t* out_idx, float* out_dist, void* stream) {
    KN_REQUIRE(wide && out_idx && out_dist, "knn_rescore: null pointer");
    KN_REQUIRE(nq > 0 && nq <= 0x7FFFFFFFll && k >= 1 && k <= KMAX, "knn_rescore: bad sizes");
    if (exact) {
        KN_REQUIRE(q && q_norm && q_sq && pool && p_norm && p_sq && np > 0, "knn_rescore: the exact pass needs the fp32 operands and their norms");
        KN_REQUIRE(dim > 0 && dim % 4 == 0 && ((uintptr_t)q & 15) == 0 && ((uintptr_t)pool & 15) == 0, "knn_rescore: dim % 4 == 0 and 16-byte aligned rows");
    }
    launch_rescore((const unsigned long long*)wide, (long)nq, k, q, q_norm, q_sq, pool, p_norm, p_sq, dim, (long)idx_offset, (long)mask_lo,
                   (long)mask_hi, exact ? 1 : 0, (long*)out_idx, out_dist, (hipStream_t)stream);
    return knnsvc_check_launch("knn_rescore");
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
                                 const float* p_norm, const float* p_sq, int64_t np, int32_t k,
                                 int64_t mask_lo, int64_t mask_hi,
                                 void* wide_out, int32_t* nan_flag, void* stream) {
    KN_REQUIRE(dots && q_norm && q_sq && p_norm && p_sq && wide_out && nan_flag, "knn_select: null pointer");
    KN_REQUIRE(nq > 0 && np > 0 && ld >= np, "knn_select: empty query or pool, or ld < np");
    KN_REQUIRE(k >= 1 && k <= KMAX, "knn_select: k must be in 1..32");
    KN_REQUIRE(np >= k, "knn_select: pool smaller than k (the reference's topk would raise)");
    KN_REQUIRE(np < (1ll << 32) && nq <= 0x7FFFFFFFll, "knn_select: pool rows must fit 32 bits");
    // KNNSVC_KNN_SCREEN=0 evaluates the reference formula on every element (A/B and equivalence tests); read per call
    const char* e = getenv("KNNSVC_KNN_SCREEN");
    if (e && e[0] == '0')
        hipLaunchKernelGGL(knn_select_kernel<false>, dim3((unsigned)nq), dim3(256), 0, (hipStream_t)stream, dots, (long)ld, q_norm, q_sq,
                           (long)nq, p_norm, p_sq, (long)np, k, (long)mask_lo, (long)mask_hi, (unsigned long long*)wide_out, nan_flag);
    else
        hipLaunchKernelGGL(knn_select_kernel<true>, dim3((unsigned)nq), dim3(256), 0, (hipStream_t)stream, dots, (long)ld, q_norm, q_sq,
                           (long)nq, p_norm, p_sq, (long)np, k, (long)mask_lo, (long)mask_hi, (unsigned long long*)wide_out, nan_flag);
    return knnsvc_check_launch("knn_select");
}


extern "C" int knnsvc_knn_screen(const void* q_f16x2, const float* q_absmax, const float* q_norm, const float* q_sq, int64_t nq,
                                 const void* p_f16x2, const float* p_absmax, const float* p_norm, const float* p_sq, int64_t np,
                                 int32_t dim, const float* thr, const int64_t* thr_idx, int64_t mask_lo, int64_t mask_hi, int64_t p_base,
                                 int32_t* cand_count, void* cand, int32_t cap, uint32_t* cold_ws, int32_t* overflow_flag, int32_t max_blocks,
                                 void* stream) {
    KN_REQUIRE(q_f16x2 && q_absmax && q_norm && q_sq && p_f16x2 && p_absmax && p_norm && p_sq && cand_count && cand && overflow_flag,
               "knn_screen: null pointer");
    KN_REQUIRE((thr == nullptr) == (thr_idx == nullptr), "knn_screen: thr and thr_idx come together (both null: the first epoch)");
    KN_REQUIRE(nq > 0 && np > 0 && dim >= 32 && dim % 32 == 0 && cap > 0 && p_base >= 0, "knn_screen: bad sizes (dim must be a multiple of 32)");
    KN_REQUIRE(nq * (long)dim * 4 < (1L << 30) && np * (long)dim * 4 < (1L << 30), "knn_screen: operands must stay below 1 GiB (chunk the call)");
    KN_REQUIRE(p_base + np < (1ll << 32), "knn_screen: pool rows must fit 32 bits");
    KN_REQUIRE(((uintptr_t)q_f16x2 & 15) == 0 && ((uintptr_t)p_f16x2 & 15) == 0, "knn_screen: 16-byte alignment");
    KN_REQUIRE(max_blocks >= 0, "knn_screen: max_blocks must be >= 0 (0 = one block per CU)");
    static_assert(QG::LDS_BYTES >= (256 * 10 + 4) * 4 + SCR_LIST * 12 + SCR_QD * 256 * 8, "epilogue state fits the operand stages");
    static_assert(SCR_QD > QG::TM * QG::NR, "a lane queue holds more than one column's elements");
    static_assert(SCR_QD * 256 * 8 >= (520 + 512) * 4, "the cold pass's per-half bounds fit the (empty) queue area");
    static bool attr = false;
    static int cus = 0;
    if (!attr) {
        if (hipFuncSetAttribute((const void*)knn_screen_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, QG::LDS_BYTES) != hipSuccess ||
            hipFuncSetAttribute((const void*)knn_screen_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, QG::LDS_BYTES) != hipSuccess ||
            hipFuncSetAttribute((const void*)knn_screen_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, QG::LDS_BYTES) != hipSuccess)
            return knnsvc_fail(KNNSVC_EHIP, "knn_screen: hipFuncSetAttribute failed");
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        attr = true;
    }
    const long ntiles = quad_order_ids(cdiv64(nq, 256), cdiv64(np, 256));
    KN_REQUIRE(ntiles < (1L << 31), "knn_screen: too many tiles (chunk the call)");
    long blocks = max_blocks > 0 && max_blocks < cus ? max_blocks : cus;
    if (blocks >= 8) blocks = blocks / 8 * 8;          // a multiple of 8: with the padded (XCD-aware) tile order a block's tiles keep their XCD
    if (blocks > ntiles) blocks = ntiles;
    const long gy = cdiv64(np, 256);
    // the exchange pays from ~20 column tiles on (>= 32 bounds needed) — and only when every tile of the epoch is resident at once: under
    // a grid cap (max_blocks: the lanes of the stream pipeline) the tiles of a row tile run one after another in the same block, and
    // thread 0 would spin its full 8 us per tile for arrivals that cannot come
    const bool xch = !thr && cold_ws && 2 * gy >= 40 && 2 * gy <= XCH_MAX && blocks >= cdiv64(nq, 256) * gy;
#define KN_LAUNCH(COLD_, XCH_)                                                                                                       \
    hipLaunchKernelGGL((knn_screen_kernel<COLD_, XCH_>), dim3((unsigned)blocks), dim3(256), QG::LDS_BYTES, (hipStream_t)stream,       \
                       (const float*)q_f16x2, q_absmax, q_norm, q_sq, (long)nq, (const unsigned short*)p_f16x2, p_absmax, p_norm, p_sq, \
                       (long)np, dim, thr, (const long*)thr_idx, (long)mask_lo, (long)mask_hi, (long)p_base, cand_count,              \
                       (unsigned*)cand, cap, (unsigned*)cold_ws, overflow_flag, (int)ntiles)
    if (thr) KN_LAUNCH(false, false);
    else if (xch) KN_LAUNCH(true, true);
    else KN_LAUNCH(true, false);
#undef KN_LAUNCH
    return knnsvc_check_launch("knn_screen");
}

extern "C" int knnsvc_knn_refine(int32_t* cand_count, const void* cand, int32_t cap, int64_t nq, int32_t k,
                                 const uint32_t* row_bound, int32_t has_prev, void* wide, float* thr_out,
                                 int64_t* thr_idx_out, int32_t final_pass, int32_t* flags, void* stream) {
    KN_REQUIRE(cand_count && cand && wide && flags, "knn_refine: null pointer");
    KN_REQUIRE((thr_out == nullptr) == (thr_idx_out == nullptr), "knn_refine: pointer pairs come together");
    KN_REQUIRE(nq > 0 && cap > 0 && k >= 1 && k <= KMAX, "knn_refine: bad sizes");
    hipLaunchKernelGGL(knn_refine_kernel, dim3((unsigned)cdiv64(nq, 4)), dim3(256), 0, (hipStream_t)stream, cand_count,
                       (const unsigned*)cand, cap, (long)nq, k, (const unsigned*)row_bound, has_prev ? 1 : 0, (unsigned long long*)wide,
                       thr_out, (long*)thr_idx_out, final_pass, flags);
    return knnsvc_check_launch("knn_refine");
}
