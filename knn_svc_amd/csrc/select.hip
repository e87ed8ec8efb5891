// Neighbour post-processing kernels: median log-f0 (radix select), f0 shift, stable f0
// re-rank, and the frame-sequential concatenation-cost re-selection.
// Reference: ddsp_prematch_dataset.py:1224-1233, 954-1016; lib_ongaku_test.py:270-369.
#include "common.h"
#include <cstdlib>

namespace {

__device__ __forceinline__ unsigned f2s(float d) {
    unsigned u = __float_as_uint(d);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
// f32 log / exp / log2 evaluated in f64 and rounded once: correctly rounded f32 results.  The
// reference's values come from the host libm (SLEEF, <= 1 ulp); the shifted f0 feeds a phase
// integrator in the synthesiser, so every ulp here is worth ~1e-4 cycles after 30 s.
__device__ __forceinline__ float log_rn(float x) { return (float)log((double)x); }
__device__ __forceinline__ float exp_rn(float x) { return (float)exp((double)x); }
__device__ __forceinline__ float log2_rn(float x) { return (float)log2((double)x); }

__device__ __forceinline__ float s2f(unsigned s) {
    return __uint_as_float((s & 0x80000000u) ? (s & 0x7FFFFFFFu) : ~s);
}

// lower median (torch.median) of log(f0) over f0 != 0, by 4-pass byte radix select in one block
__global__ __launch_bounds__(1024) void log_f0_median_kernel(const float* __restrict__ f0, long n,
                                                            float* __restrict__ result, float* __restrict__ ws) {
    __shared__ unsigned hist[256];
    __shared__ unsigned s_prefix, s_rank, s_count;
    const int tid = threadIdx.x;
    if (tid == 0) s_count = 0;
    __syncthreads();
    unsigned local = 0;
    for (long i = tid; i < n; i += 1024) {
        const float f = f0[i];
        if (f != 0.f) { ws[i] = log_rn(f); ++local; } else ws[i] = __builtin_nanf("");
    }
    atomicAdd(&s_count, local);
    __syncthreads();
    const unsigned nv = s_count;
    if (nv == 0) { if (tid == 0) { result[0] = __builtin_nanf(""); result[1] = 0.f; } return; }
    if (tid == 0) { s_prefix = 0; s_rank = (nv - 1) / 2; }
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = 24 - 8 * pass;
        if (tid < 256) hist[tid] = 0;
        __syncthreads();
        const unsigned prefix = s_prefix;
        const unsigned himask = pass == 0 ? 0u : (0xFFFFFFFFu << (shift + 8));
        for (long i = tid; i < n; i += 1024) {
            const float v = ws[i];
            if (v == v) {
                const unsigned s = f2s(v);
                if ((s & himask) == prefix) atomicAdd(&hist[(s >> shift) & 255u], 1u);
            }
        }
        __syncthreads();
        if (tid == 0) {
            unsigned r = s_rank, b = 0;
            for (; b < 256; ++b) { if (r < hist[b]) break; r -= hist[b]; }
            s_rank = r;
            s_prefix = prefix | (b << shift);
        }
        __syncthreads();
    }
    if (tid == 0) { result[0] = s2f(s_prefix); result[1] = (float)nv; }
}

__global__ void shift_f0_kernel(const float* __restrict__ f0, long n, const float* __restrict__ qmed,
                                const float* __restrict__ pmed, float* __restrict__ out) {
#pragma clang fp contract(off)
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float f = f0[i];
    out[i] = f != 0.f ? exp_rn((log_rn(f) + pmed[0]) - qmed[0]) : f;
}

// half a wave (32 lanes) per query row; rank by counting gives torch.sort(stable=True) order
__global__ __launch_bounds__(256) void f0_rerank_kernel(const long* __restrict__ nn, long nq, int k,
                                                       const float* __restrict__ sf0, const float* __restrict__ pf0,
                                                       long* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= nq) return;
    const bool act = lane < k;
    const long id = act ? nn[row * k + lane] : 0;
    const float key = act ? fabsf(log2_rn(pf0[id] + 1e-5f) - log2_rn(sf0[row] + 1e-5f)) : __builtin_inff();
    int rank = 0;
    for (int j = 0; j < k; ++j) {
        const float kj = __shfl(key, j, 64);
        rank += (kj < key || (kj == key && j < lane)) ? 1 : 0;
    }
    if (act) out[row * k + rank] = id;
}

// ---------------------------------------------------------------------------------------------
// knn_with_concat_cost.  One block (4 waves) walks the frames of one sequence.  Candidate rows
// live in a two-slot LDS ring so that the previous frame's selection is still resident when the
// concat costs are formed.  Distances follow fast_cosine_dist on torch.cdist's direct route
// (both sides <= 25 rows): cd = sqrt(sum (x-y)^2); d = 1 - (((-cd^2 + |x|^2) + |y|^2)/2)/(|x||y|).
// ---------------------------------------------------------------------------------------------
constexpr int KC = 4;        // neighbours kept per frame
constexpr int NC = 2 * KC;   // candidates per frame

__device__ __forceinline__ float cos_from_cd(float ss, float xn, float yn) {
#pragma clang fp contract(off)
    const float cd = sqrtf(ss);
    float dp = ((-(cd * cd)) + xn * xn) + yn * yn;
    dp = dp / 2.0f;
    return 1.0f - dp / (xn * yn);
}

__device__ __forceinline__ float sqdiff(const float* __restrict__ x, const float* __restrict__ y, int dim, int lane) {
    float s = 0.f;
    for (int c = lane * 4; c < dim; c += 256) {
        const f32x4 a = *(const f32x4*)(x + c), b = *(const f32x4*)(y + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float d = a[e] - b[e]; s += d * d; }
    }
    return wave_sum(s);
}

__global__ __launch_bounds__(256) void concat_reselect_kernel(
    const long* __restrict__ idx_in, const float* __restrict__ q, const float* __restrict__ qn, long nq,
    const float* __restrict__ pool, const float* __restrict__ pn, long np, int dim,
    const float* __restrict__ sf0, const float* __restrict__ pf0, int use_f0, float concat_weight,
    long* __restrict__ idx_out) {
#pragma clang fp contract(off)
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* ring = sm;                               // [2][NC][dim]
    float* qrow = ring + 2 * NC * dim;              // [2][dim]
    __shared__ float s_match[NC], s_cc[KC][NC], s_base;
    __shared__ long s_cand[2][NC];
    __shared__ int s_prev_slot[KC];                 // slots (in the previous ring half) of the kept rows
    __shared__ long s_prev_idx[KC];
    __shared__ float s_w;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // frame 0 keeps its neighbours
    if (tid < KC) {
        const long id = idx_in[tid];
        idx_out[tid] = id;
        s_prev_idx[tid] = id; s_prev_slot[tid] = tid; s_cand[0][tid] = id;
    }
    if (tid == 0) s_w = concat_weight;
    __syncthreads();
    for (int r = 0; r < KC; ++r)
        for (int c = tid * 4; c < dim; c += 1024) *(f32x4*)&ring[(0 * NC + r) * dim + c] = *(const f32x4*)(pool + s_cand[0][r] * (long)dim + c);
    for (int c = tid * 4; c < dim; c += 1024) *(f32x4*)&qrow[c] = *(const f32x4*)(q + c);
    __syncthreads();

    for (long i = 1; i < nq; ++i) {
        const int cur = (int)(i & 1), prv = cur ^ 1;
        if (tid < NC) {
            long id;
            if (tid < KC) id = idx_in[i * KC + tid];
            else { id = s_prev_idx[tid - KC] + 1; if (id >= np) id = np - 1; }
            s_cand[cur][tid] = id;
        }
        __syncthreads();
        // stage the 8 candidate rows and q[i]
        for (int r = 0; r < NC; ++r) {
            const float* src = pool + s_cand[cur][r] * (long)dim;
            for (int c = tid * 4; c < dim; c += 1024) *(f32x4*)&ring[(cur * NC + r) * dim + c] = *(const f32x4*)(src + c);
        }
        for (int c = tid * 4; c < dim; c += 1024) *(f32x4*)&qrow[cur * dim + c] = *(const f32x4*)(q + i * (long)dim + c);
        __syncthreads();
        // 8 match + 32 concat + 1 baseline squared distances, spread over the 4 waves
        for (int job = wave; job < NC + KC * NC + 1; job += 4) {
            if (job < NC) {
                const float ss = sqdiff(&qrow[cur * dim], &ring[(cur * NC + job) * dim], dim, lane);
                if (lane == 0) s_match[job] = cos_from_cd(ss, qn[i], pn[s_cand[cur][job]]);
            } else if (job < NC + KC * NC) {
                const int a = (job - NC) / NC, b = (job - NC) % NC;
                const float ss = sqdiff(&ring[(prv * NC + s_prev_slot[a]) * dim], &ring[(cur * NC + b) * dim], dim, lane);
                if (lane == 0) s_cc[a][b] = cos_from_cd(ss, pn[s_prev_idx[a]], pn[s_cand[cur][b]]);
            } else {
                const float ss = sqdiff(&qrow[prv * dim], &qrow[cur * dim], dim, lane);
                if (lane == 0) s_base = cos_from_cd(ss, qn[i - 1], qn[i]) * 2.0f;
            }
        }
        __syncthreads();
        if (wave == 0) {
            float total = __builtin_inff();
            const float base = s_base;
            float w = s_w;
            if (use_f0 && !(base < 0.08f)) w = 0.f;            // sticky: stays 0 for every later frame
            if (lane < NC) {
                float c4[KC];
#pragma unroll
                for (int a = 0; a < KC; ++a) {
                    float c = s_cc[a][lane];
                    if (use_f0) { if (base < 0.08f && c < 5.0f * base) c = 0.f; }
                    else if (c > base) c = 1.5f * c - base;
                    c4[a] = c;
                }
                // lower median of four = second smallest
                float lo01 = fminf(c4[0], c4[1]), hi01 = fmaxf(c4[0], c4[1]);
                float lo23 = fminf(c4[2], c4[3]), hi23 = fmaxf(c4[2], c4[3]);
                const float med = fminf(fmaxf(lo01, lo23), fminf(hi01, hi23));
                total = w * med + s_match[lane];
                if (use_f0) {
                    const float lp = log2_rn(pf0[s_cand[cur][lane]] + 1e-5f), lq = log2_rn(sf0[i] + 1e-5f);
                    total = total + fabsf(lp - lq);
                }
            }
            // A NaN cost (NaN features: the reference has exited in fast_cosine_dist by then, here the flag is read only after
            // everything is enqueued) must still rank: as +inf, ties by candidate number.  Unranked, all eight candidates took
            // rank 0, the kept-row broadcast read slot 64, and the next frame gathered pool rows through garbage indices.
            if (!(total == total)) total = __builtin_inff();
            int rank = 0;
#pragma unroll
            for (int j = 0; j < NC; ++j) {          // v_readlane (an SGPR broadcast), not a ds_bpermute round trip per candidate
                const float tj = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, total), j));
                rank += (tj < total || (tj == total && j < lane)) ? 1 : 0;
            }
            if (lane < NC && rank < KC) {
                const long id = s_cand[cur][lane];
                s_prev_idx[rank] = id; s_prev_slot[rank] = lane;
                idx_out[i * KC + rank] = id;
            }
            if (lane == 0) s_w = w;
        }
        __syncthreads();
    }
}



// buffer resource over a whole array (< 4 GiB: the dispatcher sends larger pools to the generic kernel)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t sel_rsrc(const void* p, unsigned long long bytes) {
    const unsigned long long u = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
    const unsigned nb = __builtin_amdgcn_readfirstlane((unsigned)bytes);
    return __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), 0, (int)nb, 0x00020000);
}

// ---------------------------------------------------------------------------------------------
// Pipelined walk (feature dim <= 1024).  Everything frame i needs is in LDS when the frame starts: its 4 kNN rows and q[i] were
// fetched during frame i-1, and the "previous selection + 1" rows come from a speculative fetch of the successors of ALL eight
// candidates of frame i-1 (4 of them are used).  Round 3's version of this walk (8 waves, one candidate each) took 10 900 cycles
// per frame; its cycle counters (tools/concat_prof.py) said where they went: 4 500 in the distances (one candidate per wave re-read the query and the four
// previous rows: 196 KB of LDS reads per frame; 3 VALU per element and 12 per wave-wide sum, two waves per SIMD; then ONE lane
// worked six sqrt / divide chains), 1 600 issuing the prefetch, 1 700 + 1 500 in the per-lane bookkeeping and the selection on
// wave 0 (everyone else at the barrier, or running the same selection on the same SIMD), 1 200 copying rows between LDS regions.
//   * nine waves.  Waves 0-7 compute distances; waves 0-3 fetch and stage the eight speculative successor rows and run the
//     selection (one copy per SIMD), waves 4-7 fetch and stage the rows that do not depend on it (next kNN rows, next query);
//     the ninth wave keeps the books (norm / f0 / id prefetch of the 13 new rows, log2 f0, next frame's tables and load
//     offsets, idx_out) — nothing scalar is left on a wave the others wait for;
//   * distances: wave (p, h) owns candidates 2p, 2p + 1 over column half h — seven rows of half length per wave (112 KB per
//     frame), fused multiply-adds, and ONE transposed fold for its twelve sums (two select-and-add DPP steps leave value
//     4 m + (lane & 3) in register m, two row rotations finish the 16 lanes of a row: 33 VALU instead of 132); the 2 x 4
//     partial sums per distance (column half x row of lanes) meet in LDS and are added in a fixed order;
//   * after the barrier lane ref * 8 + cand of waves 0-3 evaluates one cos_from_cd, four ds_bpermutes bring a candidate's
//     concat costs to its lane; wave 0 publishes the kept slots and ranks, the book-keeper polls an LDS word for them
//     and writes next frame's tables beside the staging (two barriers per frame);
//   * the previous selection is not copied: its rows stay where they are (A[prv] / S[prv]) and are addressed through the kept
//     slot numbers; of the eight speculative successor rows only the four kept ones are written to LDS (S[nxt][rank]), so
//     candidate c always sits at a fixed offset;
//   * tables (id, norm, log2 f0 per candidate; norms of the reference rows; byte offsets of the rows to prefetch; kept slots)
//     are READ only before the first barrier of a frame and WRITTEN only after it, each by the book-keeping lane that already
//     holds the value (a DPP row shift hands the rank of candidate c to the lane holding its successor).
//   LDS rows: A[2][4] kNN rows | S[2][4] successors of the kept rows | Q[2] queries  (18 rows, 72 KB at D = 1024).
// Candidate order, tie rule, sticky weight and NaN handling are those of concat_reselect_kernel above; sums of squares are
// taken in a different order than there and with fused multiply-adds (equal up to fp32 rounding of the accumulation).
// ---------------------------------------------------------------------------------------------
constexpr int LT = 576;        // 8 distance waves + the book-keeper

#define KN_DPPF(V, CTRL) __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, (V)), (CTRL), 0xF, 0xF, true))

template <bool use_f0>
__global__ __launch_bounds__(LT) void concat_reselect_pipe_kernel(
    const long* __restrict__ idx_in, const float* __restrict__ q, const float* __restrict__ qn, long nq,
    const float* __restrict__ pool, const float* __restrict__ pn, long np, int dim,
    const float* __restrict__ sf0, const float* __restrict__ pf0, float concat_weight, long* __restrict__ idx_out) {
#pragma clang fp contract(off)
    extern __shared__ __attribute__((aligned(16))) float sm[];
    typedef __attribute__((address_space(3))) float lf;
    typedef __attribute__((address_space(3))) f32x4 lf4;
    lf* L = (lf*)sm;
    const int D = dim;
    const int offA = 0, offS = 8 * D, offQ = 24 * D;                       // floats: A[2][4] | S[2][8] | Q[2]
    __shared__ int s_id[2][NC];                                            // candidate tables by frame parity
    __shared__ float s_pn[2][NC], s_lf0[2][NC];
    __shared__ float s_ref[2][8];          // [parity][0] |q|, [1..4] norms of the previous selection (by rank), [5] log2 source f0
    __shared__ unsigned s_off[2][16];      // byte offsets of the 13 rows fetched DURING a frame of this parity (for the next frame)
    __shared__ int s_rank[NC];             // last selection: rank by candidate slot (-1: dropped)
    __shared__ int s_poff[KC], s_soff[KC]; // LDS offsets (floats) of the rows kept last (by rank) and of their successors
    __shared__ int s_done;                 // frame whose selection wave 0 has published (the book-keeper polls it)
    __shared__ __attribute__((aligned(16))) float s_part[48][8];   // [ref * 8 + cand][column half * 4 + row of 16 lanes]; [40] = |q[i-1] - q[i]|^2, [41..47] unused

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool bk = wave == 8;             // the book-keeper: no rows, no distances
    const bool sel = wave < 4;             // runs the selection, stages the successor rows
    const int col = (tid & 255) * 4;       // 256 threads (waves 0-3, waves 4-7) cover one 1024-float row with float4
    const bool colok = col < D && !bk;
    const unsigned row_bytes = (unsigned)D * 4u;

    // ---- frame 0 keeps its neighbours; tables, load offsets and rows of frame 1 ------------------------
    if (tid < 4) {
        const long id0 = idx_in[tid];
        idx_out[tid] = id0;
        s_id[0][tid] = (int)id0;
        s_ref[1][1 + tid] = pn[id0];
        s_poff[tid] = offA + tid * D; s_soff[tid] = offS + (NC + tid) * D;
        if (tid == 0) s_done = 0;
        if (nq > 1) {
            long sid = id0 + 1; if (sid >= np) sid = np - 1;
            s_id[1][KC + tid] = (int)sid; s_pn[1][KC + tid] = pn[sid]; s_lf0[1][KC + tid] = use_f0 ? log2_rn(pf0[sid] + 1e-5f) : 0.f;
            const long a = idx_in[KC + tid];
            s_id[1][tid] = (int)a; s_pn[1][tid] = pn[a]; s_lf0[1][tid] = use_f0 ? log2_rn(pf0[a] + 1e-5f) : 0.f;
            long sa = a + 1; if (sa >= np) sa = np - 1;
            long ss = sid + 1; if (ss >= np) ss = np - 1;
            s_off[1][5 + tid] = (unsigned)sa * row_bytes; s_off[1][5 + KC + tid] = (unsigned)ss * row_bytes;
            s_off[1][tid] = nq > 2 ? (unsigned)idx_in[2 * KC + tid] * row_bytes : 0u;
        }
    }
    if (tid == 0) {
        s_ref[0][0] = qn[0];
        if (nq > 1) { s_ref[1][0] = qn[1]; s_ref[1][5] = use_f0 ? log2_rn(sf0[1] + 1e-5f) : 0.f; s_off[1][4] = nq > 2 ? 2u * row_bytes : 0u; }
    }
    __syncthreads();
    if (colok) {
        const int half = tid >> 8;
        for (int r = half; r < 4; r += 2) {
            *(lf4*)&L[offA + r * D + col] = *(const f32x4*)(pool + (long)s_id[0][r] * D + col);
            if (nq > 1) {
                *(lf4*)&L[offA + (4 + r) * D + col] = *(const f32x4*)(pool + (long)s_id[1][r] * D + col);
                *(lf4*)&L[offS + (NC + r) * D + col] = *(const f32x4*)(pool + (long)s_id[1][KC + r] * D + col);
            }
        }
        if (half == 0) *(lf4*)&L[offQ + col] = *(const f32x4*)(q + col);
        else if (nq > 1) *(lf4*)&L[offQ + D + col] = *(const f32x4*)(q + (long)D + col);
    }
    // book-keeper lanes 0..3: ids of the kNN rows of frame i + 1, carried from frame to frame (loaded two frames ahead)
    long id_a = (bk && lane < 4 && nq > 2) ? idx_in[2 * KC + lane] : 0;
    __syncthreads();

    const __amdgpu_buffer_rsrc_t p_rsrc = sel_rsrc(pool, (unsigned long long)np * row_bytes), q_rsrc = sel_rsrc(q, (unsigned long long)nq * row_bytes);
    const int dp = (wave >> 1) & 3, dh = wave & 1;         // distance work: candidate pair, column half
    const int hcols = ((D / 4 + 1) / 2) * 4;               // columns per half (float4 granules, first half takes the odd one)
    const bool odd1 = lane & 1, odd2 = lane & 2;
    float w = concat_weight;                               // sticky: once 0 (f0 variant) it stays 0
#ifdef KN_CONCAT_PROF
    unsigned long long pf[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tq = 0;
#define KN_TICK(K) { const unsigned long long now = __builtin_readcyclecounter(); pf[K] += now - tq; tq = now; }
#else
#define KN_TICK(K)
#endif
    for (long i = 1; i < nq; ++i) {
#ifdef KN_CONCAT_PROF
        tq = __builtin_readcyclecounter();
#endif
        const int cur = (int)(i & 1), nxt = cur ^ 1;
        const bool more = (i + 1 < nq);
        f32x4 pre[8];
        int rid = 0; long my_id = 0, nn_id = 0; float my_pn = 0.f, my_f0 = 0.f; int my_kind = -1;
        float rpn = 0.f, rlf0 = 0.f, xref = 0.f, qn_prv = 0.f, qn_cur = 0.f, lsf0 = 0.f;
        const int c8 = lane & 7, refi = lane >> 3;
        int oP[KC] = {0, 0, 0, 0}, oS[KC] = {0, 0, 0, 0};      // LDS offsets of the previous selection / of candidates 4..7 (wave-uniform)
        if (bk) {
            // ---- book-keeper: norm / f0 of the 13 rows being fetched, the kNN ids two frames ahead; lanes 0-3 kNN rows of
            //      frame i + 1, lane 4 its query, lanes 5-12 successors of this frame's candidates --------------------
            rid = s_id[cur][c8]; rpn = s_pn[cur][c8];
            const int l = lane < 13 ? lane : 12;
            long sid = (long)s_id[cur][l >= 5 ? l - 5 : 0] + 1; sid = sid >= np ? np - 1 : sid;
            const long row = l < 4 ? (more ? id_a : 0) : (l == 4 ? (more ? i + 1 : 0) : sid);
            const float* pnp = l == 4 ? qn : pn;
            const float* pfp = use_f0 ? (l == 4 ? sf0 : pf0) : pnp;                         // any valid address when f0 is off
            my_pn = pnp[row]; my_f0 = pfp[row];
            nn_id = idx_in[(i + 2 < nq ? (i + 2) * KC : 0) + (l & 3)];
            my_kind = lane < 13 ? (l < 4 ? (more ? 0 : -1) : (l == 4 ? (more ? 1 : -1) : 2)) : -1;
            my_id = (my_kind == 0 || my_kind == 2) ? row : 0;
            if (use_f0 && my_kind >= 0) my_f0 = log2_rn(my_f0 + 1e-5f);      // one evaluation per new row, behind the others' distances
            else my_f0 = 0.f;
            KN_TICK(0)
            KN_TICK(1)
        } else {
            // ---- row prefetch for frame i+1: waves 0-3 the 4 kNN rows and q[i+1] (rows 0-4 of the offset list) before their
            //      distances, waves 4-7 the successors of all 8 candidates (rows 5-12) in the middle of theirs.  Every load is
            //      unconditional (rows that are not needed read a harmless row); the byte offset of row r travels from lane r to
            //      an SGPR: buffer_load_dwordx4 v, col * 4, rsrc, soffset.
#ifdef KN_CONCAT_WHATIF_L2      // timing aid: every fetch hits the same 16 rows (cache hits); results are garbage
            const unsigned my_off = (unsigned)(lane & 15) * row_bytes + 0u * s_off[cur][lane & 15];
#else
            const unsigned my_off = s_off[cur][lane & 15];
#endif
            const int voff = (colok ? col : 0) * 4;
            if (sel) {
#pragma unroll
                for (int t = 0; t < 5; ++t)
                    pre[t] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(t == 4 ? q_rsrc : p_rsrc, voff, __builtin_amdgcn_readlane(my_off, t), 0));
            }
            // ---- table reads of this frame (tables are only written between the barriers) ------------------------
#pragma unroll
            for (int a = 0; a < KC; ++a) { oP[a] = __builtin_amdgcn_readfirstlane(s_poff[a]); oS[a] = __builtin_amdgcn_readfirstlane(s_soff[a]); }
            if (sel) {
                rpn = s_pn[cur][c8]; rlf0 = s_lf0[cur][c8];
                xref = s_ref[cur][refi < 5 ? refi : 0];
                qn_prv = s_ref[nxt][0]; qn_cur = s_ref[cur][0]; lsf0 = s_ref[cur][5];
            }
            KN_TICK(0)
            // ---- (b) squared distances: wave (dp, dh) = candidates 2 dp, 2 dp + 1 against q and the four previous rows,
            //      column half dh; the waves of pair 0 add |q[i-1] - q[i]|^2 -----------------------------------------
            const int oC0 = dp < 2 ? offA + (cur * 4 + 2 * dp) * D : oS[dp == 2 ? 0 : 2];
            const int oC1 = dp < 2 ? oC0 + D : oS[dp == 2 ? 1 : 3];
            const int oQ = offQ + cur * D, oQp = offQ + nxt * D;
            float v[12];                        // v[2 ref + cc]: candidate 2 dp + cc against reference ref; v[10] the baseline
#pragma unroll
            for (int j = 0; j < 12; ++j) v[j] = 0.f;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                if (k == 1 && !sel) {           // waves 4-7 fetch between their column steps: the texture path takes ~16 cycles per
#pragma unroll                                  // 1 KB request, and eight waves asking at once only wait for each other
                    for (int t = 0; t < NC; ++t)
                        pre[t] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(p_rsrc, voff, __builtin_amdgcn_readlane(my_off, 5 + t), 0));
                }
                const int cl = lane * 4 + 256 * k, c = dh * hcols + cl;
                if (cl < hcols && c < D) {
                    const f32x4 x0 = *(const lf4*)&L[oC0 + c], x1 = *(const lf4*)&L[oC1 + c];
                    const f32x4 qv = *(const lf4*)&L[oQ + c];
                    const f32x4 p0 = *(const lf4*)&L[oP[0] + c], p1 = *(const lf4*)&L[oP[1] + c];
                    const f32x4 p2 = *(const lf4*)&L[oP[2] + c], p3 = *(const lf4*)&L[oP[3] + c];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float d;
                        d = qv[e] - x0[e]; v[0] = fmaf(d, d, v[0]);
                        d = qv[e] - x1[e]; v[1] = fmaf(d, d, v[1]);
                        d = p0[e] - x0[e]; v[2] = fmaf(d, d, v[2]);
                        d = p0[e] - x1[e]; v[3] = fmaf(d, d, v[3]);
                        d = p1[e] - x0[e]; v[4] = fmaf(d, d, v[4]);
                        d = p1[e] - x1[e]; v[5] = fmaf(d, d, v[5]);
                        d = p2[e] - x0[e]; v[6] = fmaf(d, d, v[6]);
                        d = p2[e] - x1[e]; v[7] = fmaf(d, d, v[7]);
                        d = p3[e] - x0[e]; v[8] = fmaf(d, d, v[8]);
                        d = p3[e] - x1[e]; v[9] = fmaf(d, d, v[9]);
                    }
                    if (dp == 0) {
                        const f32x4 qp = *(const lf4*)&L[oQp + c];
#pragma unroll
                        for (int e = 0; e < 4; ++e) { const float d = qp[e] - qv[e]; v[10] = fmaf(d, d, v[10]); }
                    }
                }
            }
            // transposed fold: after step 1 even lanes hold v[2j], odd lanes v[2j+1] (summed over lane pairs); after step 2
            // register m holds v[4m + (lane & 3)] summed over the quad; two rotations by 4 and 8 lanes finish the row of 16
            float r1[6], r2[3];
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const float keep = odd1 ? v[2 * j + 1] : v[2 * j], send = odd1 ? v[2 * j] : v[2 * j + 1];
                r1[j] = keep + KN_DPPF(send, 0xB1);                       // quad_perm [1,0,3,2]
            }
#pragma unroll
            for (int m = 0; m < 3; ++m) {
                const float keep = odd2 ? r1[2 * m + 1] : r1[2 * m], send = odd2 ? r1[2 * m] : r1[2 * m + 1];
                r2[m] = keep + KN_DPPF(send, 0x4E);                       // quad_perm [2,3,0,1]
            }
#pragma unroll
            for (int m = 0; m < 3; ++m) {
                r2[m] += KN_DPPF(r2[m], 0x124);                           // row_ror:4
                r2[m] += KN_DPPF(r2[m], 0x128);                           // row_ror:8
            }
            if ((lane & 15) < 4) {
                const int j = lane & 3, rowq = lane >> 4;
#pragma unroll
                for (int m = 0; m < 3; ++m) {
                    const int idx = 4 * m + j;                            // = 2 ref + cc; 10: baseline; 11: nothing
                    const int e = idx < 10 ? (idx >> 1) * 8 + 2 * dp + (idx & 1) : (idx == 10 && dp == 0 ? 40 : 41 + dp);
                    s_part[e][dh * 4 + rowq] = r2[m];
                }
            }
            KN_TICK(1)
        }
        __syncthreads();
        KN_TICK(2)
        // ---- (c) waves 0-3 (one per SIMD): lane ref * 8 + cand evaluates one distance (lane 40: the baseline), candidates'
        //      lanes collect their costs, lower median over the previous selection, 4 smallest; wave 0 publishes ----------
        int my_slot = -1;                       // lane < 8: rank of candidate `lane` if kept
        if (sel) {
            const int pl = lane < 41 ? lane : 40;
            const f32x4 pa = *(const f32x4*)&s_part[pl][0], pb = *(const f32x4*)&s_part[pl][4];
            const float h0 = (pa[0] + pa[1]) + (pa[2] + pa[3]);
            const float h1 = (pb[0] + pb[1]) + (pb[2] + pb[3]);
            const float ss = h0 + h1;
            const float val = cos_from_cd(ss, lane < 40 ? xref : qn_prv, lane < 40 ? rpn : qn_cur);
            const float base = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, val), 40)) * 2.0f;
            float c4[KC];
#pragma unroll
            for (int a = 0; a < KC; ++a) c4[a] = __shfl(val, 8 * (a + 1) + c8, 64);
            if (use_f0 && !(base < 0.08f)) w = 0.f;
            float total = __builtin_inff();
            if (lane < NC) {
#pragma unroll
                for (int a = 0; a < KC; ++a) {
                    float c = c4[a];
                    if (use_f0) { if (base < 0.08f && c < 5.0f * base) c = 0.f; }
                    else if (c > base) c = 1.5f * c - base;
                    c4[a] = c;
                }
                const float lo01 = fminf(c4[0], c4[1]), hi01 = fmaxf(c4[0], c4[1]);
                const float lo23 = fminf(c4[2], c4[3]), hi23 = fmaxf(c4[2], c4[3]);
                const float med = fminf(fmaxf(lo01, lo23), fminf(hi01, hi23));
                total = w * med + val;
                if (use_f0) total = total + fabsf(rlf0 - lsf0);            // both already log2(f0 + 1e-5)
            }
            if (!(total == total)) total = __builtin_inff();       // NaN costs rank as +inf, ties by candidate number (see above)
            int rank = 0;
#pragma unroll
            for (int j = 0; j < NC; ++j) {
                const float tj = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, total), j));
                rank += (tj < total || (tj == total && j < lane)) ? 1 : 0;
            }
            if (lane < NC && rank < KC) my_slot = rank;
            if (wave == 0) {
                if (lane < NC) {
                    s_rank[lane] = my_slot;
                    if (my_slot >= 0) {             // where the kept row lives now, where its successor is being staged
                        const int o_mine = lane < KC ? offA + (cur * 4 + lane) * D : (lane == 4 ? oS[0] : (lane == 5 ? oS[1] : (lane == 6 ? oS[2] : oS[3])));
                        s_poff[my_slot] = o_mine; s_soff[my_slot] = offS + (nxt * NC + lane) * D;
                    }
                }
                if (lane == 0) __hip_atomic_store(&s_done, (int)i, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);   // LDS ops of a wave execute in order
            }
        }
        KN_TICK(3)
        // ---- (d) prefetched rows -> LDS: waves 4-7 beside the selection S[nxt][c] = successor of candidate c (all eight: which
        //      four are used is settled by the offsets wave 0 publishes), waves 0-3 after it A[nxt], Q[nxt] (beside the book-keeper)
        if (more && colok) {
            if (sel) {
#pragma unroll
                for (int t = 0; t < 4; ++t) *(lf4*)&L[offA + (nxt * 4 + t) * D + col] = pre[t];
                *(lf4*)&L[offQ + nxt * D + col] = pre[4];
            } else {
#pragma unroll
                for (int t = 0; t < NC; ++t) *(lf4*)&L[offS + (nxt * NC + t) * D + col] = pre[t];
            }
        }
        KN_TICK(4)
        // ---- the book-keeper writes next frame's tables as soon as wave 0 has published the selection (beside the staging) ----
        if (bk) {
            long nx = my_id + 1; if (nx >= np) nx = np - 1;                  // successor of a candidate of frame i + 1
            const bool more2 = i + 2 < nq;
            if (more) {                          // what does not depend on the selection goes first
                if (my_kind == 0) {
                    s_id[nxt][lane] = (int)my_id; s_pn[nxt][lane] = my_pn; s_lf0[nxt][lane] = my_f0;
                    s_off[nxt][5 + lane] = (unsigned)nx * row_bytes;
                    s_off[nxt][lane] = more2 ? (unsigned)nn_id * row_bytes : 0u;
                } else if (my_kind == 1) {
                    s_ref[nxt][0] = my_pn; s_ref[nxt][5] = my_f0;
                    s_off[nxt][4] = more2 ? (unsigned)(i + 2) * row_bytes : 0u;
                }
                if (lane < 4) id_a = nn_id;
            }
            while (__hip_atomic_load(&s_done, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != (int)i) __builtin_amdgcn_s_sleep(1);
            my_slot = lane < NC ? s_rank[lane] : -1;
            if (my_slot >= 0) idx_out[i * KC + my_slot] = (long)rid;
            if (more) {
                // rank of candidate c, handed to lane 5 + c (row_shr:5 inside the first row of 16 lanes)
                const int rk_succ = __builtin_amdgcn_update_dpp(-1, my_slot, 0x115, 0xF, 0xF, false);
                if (my_kind == 2 && rk_succ >= 0) {
                    s_id[nxt][KC + rk_succ] = (int)my_id; s_pn[nxt][KC + rk_succ] = my_pn; s_lf0[nxt][KC + rk_succ] = my_f0;
                    s_off[nxt][5 + KC + rk_succ] = (unsigned)nx * row_bytes;
                }
                if (my_slot >= 0) s_ref[nxt][1 + my_slot] = rpn;                 // norms of the rows kept now
            }
        }
        KN_TICK(5)
        __syncthreads();
        KN_TICK(6)
    }
#ifdef KN_CONCAT_PROF
    if (lane == 0 && (wave == 0 || wave == 3 || wave == 5 || wave == 8))
        printf("concat prof (cycles/frame, wave %d): loads + tables | books %.0f  distances %.0f  barrier1 %.0f  select %.0f  stage %.0f  tables (book-keeper: wait + write) %.0f  barrier2 %.0f\n",
               wave, (double)pf[0] / nq, (double)pf[1] / nq, (double)pf[2] / nq, (double)pf[3] / nq, (double)pf[4] / nq, (double)pf[5] / nq, (double)pf[6] / nq);
#endif
#undef KN_TICK
}

}  // namespace

extern "C" int knnsvc_log_f0_median(const float* f0, int64_t n, float* result, float* workspace, void* stream) {
    KN_REQUIRE(f0 && result && workspace && n > 0, "log_f0_median: bad arguments");
    hipLaunchKernelGGL(log_f0_median_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, f0, (long)n, result, workspace);
    return knnsvc_check_launch("log_f0_median");
}

extern "C" int knnsvc_shift_f0(const float* f0, int64_t n, const float* query_median, const float* pool_median,
                               float* shifted, void* stream) {
    KN_REQUIRE(f0 && query_median && pool_median && shifted && n > 0, "shift_f0: bad arguments");
    hipLaunchKernelGGL(shift_f0_kernel, dim3((unsigned)cdiv64(n, 256)), dim3(256), 0, (hipStream_t)stream, f0, (long)n,
                       query_median, pool_median, shifted);
    return knnsvc_check_launch("shift_f0");
}

extern "C" int knnsvc_f0_rerank(const int64_t* nn_idx, int64_t nq, int32_t k, const float* shifted_f0,
                                const float* pool_f0, int64_t* out_idx, void* stream) {
    KN_REQUIRE(nn_idx && shifted_f0 && pool_f0 && out_idx && nq > 0, "f0_rerank: bad arguments");
    KN_REQUIRE(k >= 1 && k <= 64, "f0_rerank: k must be in 1..64");
    hipLaunchKernelGGL(f0_rerank_kernel, dim3((unsigned)cdiv64(nq, 4)), dim3(256), 0, (hipStream_t)stream,
                       (const long*)nn_idx, (long)nq, k, shifted_f0, pool_f0, (long*)out_idx);
    return knnsvc_check_launch("f0_rerank");
}

extern "C" int knnsvc_concat_reselect(const int64_t* idx_in, const float* q, const float* q_norm, int64_t nq,
                                      const float* pool, const float* p_norm, int64_t np, int32_t dim,
                                      const float* shifted_f0, const float* pool_f0, int32_t use_f0,
                                      float concat_weight, int64_t* idx_out, void* stream) {
    KN_REQUIRE(idx_in && q && q_norm && pool && p_norm && idx_out, "concat_reselect: null pointer");
    KN_REQUIRE(nq > 0 && np > 0 && dim > 0 && dim % 4 == 0, "concat_reselect: bad sizes");
    KN_REQUIRE(!use_f0 || (shifted_f0 && pool_f0), "concat_reselect: f0 variant needs both f0 arrays");
    KN_REQUIRE(((uintptr_t)q & 15) == 0 && ((uintptr_t)pool & 15) == 0, "concat_reselect: 16-byte alignment");
    if (dim <= 1024 && (unsigned long long)np * dim * 4 < 0xFFFFFFFFull && (unsigned long long)nq * dim * 4 < 0xFFFFFFFFull) {
        // KNNSVC_CONCAT_OWN_CU=1: ask for (nearly) the whole LDS of the CU so that no other kernel's workgroup is placed next to this
        // one.  Debugging aid from round 3's determinism hunt (see the Makefile's note on -fno-slp-vectorize: with compiler-made
        // packed-fp32 math this kernel's sums were perturbed by MFMA-issuing neighbours on its CU; isolation removed the symptom
        // before the cause was found).
        size_t pl = (size_t)26 * dim * 4;
        { const char* e = getenv("KNNSVC_CONCAT_OWN_CU"); if (e && e[0] == '1' && pl < (size_t)158 * 1024) pl = (size_t)158 * 1024; }
        static size_t pattr[2] = {0, 0};
        if (pl > pattr[use_f0 ? 1 : 0]) {
            const void* fn = use_f0 ? (const void*)concat_reselect_pipe_kernel<true> : (const void*)concat_reselect_pipe_kernel<false>;
            if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl) != hipSuccess)
                return knnsvc_fail(KNNSVC_EHIP, "concat_reselect: hipFuncSetAttribute failed");
            pattr[use_f0 ? 1 : 0] = pl;
        }
        if (use_f0)
            hipLaunchKernelGGL(concat_reselect_pipe_kernel<true>, dim3(1), dim3(LT), pl, (hipStream_t)stream, (const long*)idx_in, q, q_norm,
                               (long)nq, pool, p_norm, (long)np, dim, shifted_f0, pool_f0, concat_weight, (long*)idx_out);
        else
            hipLaunchKernelGGL(concat_reselect_pipe_kernel<false>, dim3(1), dim3(LT), pl, (hipStream_t)stream, (const long*)idx_in, q, q_norm,
                               (long)nq, pool, p_norm, (long)np, dim, shifted_f0, pool_f0, concat_weight, (long*)idx_out);
        return knnsvc_check_launch("concat_reselect_pipe");
    }
    const size_t lds = (size_t)(2 * NC + 2) * dim * 4;
    KN_REQUIRE(lds <= 150 * 1024, "concat_reselect: feature dim too large for LDS");
    static size_t attr = 0;
    if (lds > attr) {
        if (hipFuncSetAttribute((const void*)concat_reselect_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds) != hipSuccess)
            return knnsvc_fail(KNNSVC_EHIP, "concat_reselect: hipFuncSetAttribute failed");
        attr = lds;
    }
    hipLaunchKernelGGL(concat_reselect_kernel, dim3(1), dim3(256), lds, (hipStream_t)stream, (const long*)idx_in, q,
                       q_norm, (long)nq, pool, p_norm, (long)np, dim, shifted_f0, pool_f0, use_f0, concat_weight,
                       (long*)idx_out);
    return knnsvc_check_launch("concat_reselect");
}
