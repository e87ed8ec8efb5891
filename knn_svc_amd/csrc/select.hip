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



// ---------------------------------------------------------------------------------------------
// Pipelined variant (feature dim <= 1024): 8 waves, one per candidate.  Everything frame i needs is
// already in LDS when the frame starts: its 4 kNN rows and q[i] were prefetched during frame i-1,
// and the "previous selection + 1" rows come from a speculative prefetch of the successors of ALL
// eight candidates of frame i-1 (4 of them are used).  The global loads of frame i+1 are issued
// at the top of frame i and land behind the distance reductions, so no frame waits on HBM/L2.
//   LDS rows: A[2][4] kNN rows | S[2][8] successor rows | P[4] previous selection | Q[2] queries.
// ---------------------------------------------------------------------------------------------
constexpr int CT = 512;

// buffer resource over a whole array (< 4 GiB: the dispatcher sends larger pools to the generic kernel)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t sel_rsrc(const void* p, unsigned long long bytes) {
    const unsigned long long u = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
    const unsigned nb = __builtin_amdgcn_readfirstlane((unsigned)bytes);
    return __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), 0, (int)nb, 0x00020000);
}

__global__ __launch_bounds__(CT) void concat_reselect_pipe_kernel(
    const long* __restrict__ idx_in, const float* __restrict__ q, const float* __restrict__ qn, long nq,
    const float* __restrict__ pool, const float* __restrict__ pn, long np, int dim,
    const float* __restrict__ sf0, const float* __restrict__ pf0, int use_f0, float concat_weight,
    long* __restrict__ idx_out) {
#pragma clang fp contract(off)
    extern __shared__ __attribute__((aligned(16))) float sm[];
    typedef __attribute__((address_space(3))) float lf;
    typedef __attribute__((address_space(3))) f32x4 lf4;
    lf* L = (lf*)sm;
    const int D = dim;
    const int offA = 0, offS = 8 * D, offP = 24 * D, offQ = 28 * D;      // floats
    __shared__ long s_idA[2][4], s_idS[2][8], s_idP[4];
    __shared__ float s_pnA[2][4], s_pnS[2][8], s_pnP[4], s_qn[2], s_f0A[2][4], s_f0S[2][8], s_sf0[2];
    __shared__ long s_idNext[2][4];        // kNN ids of frame f live in s_idNext[f & 1], loaded two frames ahead
    __shared__ int s_slot[2][4];           // slots (0..7) of the kept candidates of frame i, i&1
    __shared__ float s_match[NC], s_cc[KC][NC], s_base, s_wv[2];
    // candidate table, double buffered by frame parity: frame i reads [i&1] while wave 0 fills [(i+1)&1]
    __shared__ long s_candT[2][NC];
    __shared__ float s_cpnT[2][NC], s_cf0T[2][NC];
    __shared__ int s_coffT[2][NC];         // LDS float offset of each candidate row

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col = (tid & 255) * 4;       // 256 threads cover one 1024-float row with float4
    const int half = tid >> 8;             // two rows per pass
    const bool colok = col < D;

    // ---- frame 0 ---------------------------------------------------------------------------------
    if (tid < 4) {
        const long id = idx_in[tid];
        idx_out[tid] = id;
        s_idP[tid] = id; s_pnP[tid] = pn[id];
        long sid = id + 1; if (sid >= np) sid = np - 1;
        s_idS[0][tid] = sid; s_pnS[0][tid] = pn[sid]; s_f0S[0][tid] = use_f0 ? log2_rn(pf0[sid] + 1e-5f) : 0.f;
        s_slot[0][tid] = tid;
        if (nq > 2) s_idNext[0][tid] = idx_in[8 + tid];
        if (nq > 1) { const long a = idx_in[4 + tid]; s_idA[1][tid] = a; s_pnA[1][tid] = pn[a]; s_f0A[1][tid] = use_f0 ? log2_rn(pf0[a] + 1e-5f) : 0.f; }
    }
    if (tid == 0) {
        s_wv[1] = concat_weight; s_qn[0] = qn[0];
        if (nq > 1) { s_qn[1] = qn[1]; s_sf0[1] = use_f0 ? log2_rn(sf0[1] + 1e-5f) : 0.f; }
    }
    __syncthreads();
    if (tid < NC && nq > 1) {                // candidate table of frame 1
        if (tid < KC) { s_candT[1][tid] = s_idA[1][tid]; s_cpnT[1][tid] = s_pnA[1][tid]; s_cf0T[1][tid] = s_f0A[1][tid]; s_coffT[1][tid] = offA + (4 + tid) * D; }
        else { const int sl = tid - KC; s_candT[1][tid] = s_idS[0][sl]; s_cpnT[1][tid] = s_pnS[0][sl]; s_cf0T[1][tid] = s_f0S[0][sl]; s_coffT[1][tid] = offS + sl * D; }
    }
    if (colok) {
        for (int r = half; r < 4; r += 2) {
            *(lf4*)&L[offP + r * D + col] = *(const f32x4*)(pool + s_idP[r] * (long)D + col);
            *(lf4*)&L[offS + r * D + col] = *(const f32x4*)(pool + s_idS[0][r] * (long)D + col);
            if (nq > 1) *(lf4*)&L[offA + (4 + r) * D + col] = *(const f32x4*)(pool + s_idA[1][r] * (long)D + col);
        }
        if (half == 0) *(lf4*)&L[offQ + col] = *(const f32x4*)(q + col);
        else if (nq > 1) *(lf4*)&L[offQ + D + col] = *(const f32x4*)(q + (long)D + col);
    }
    __syncthreads();

    typedef unsigned g2u4 __attribute__((ext_vector_type(4)));
    const unsigned row_bytes = (unsigned)D * 4u;
    const int whalf = __builtin_amdgcn_readfirstlane(half);                    // waves 0-3: 0, waves 4-7: 1
    const __amdgpu_buffer_rsrc_t p_rsrc = sel_rsrc(pool, (unsigned long long)np * row_bytes), q_rsrc = sel_rsrc(q, (unsigned long long)nq * row_bytes);
#ifdef KN_CONCAT_PROF
    unsigned long long pf[7] = {0, 0, 0, 0, 0, 0, 0}, tq = 0;
#define KN_TICK(K) { const unsigned long long now = __builtin_readcyclecounter(); pf[K] += now - tq; tq = now; }
#else
#define KN_TICK(K)
#endif
    for (long i = 1; i < nq; ++i) {
#ifdef KN_CONCAT_PROF
        tq = __builtin_readcyclecounter();
#endif
        const int cur = (int)(i & 1), prv = cur ^ 1;
        const long* s_cand = s_candT[cur]; const float* s_cpn = s_cpnT[cur]; const float* s_cf0 = s_cf0T[cur];
        const int* s_coff = s_coffT[cur];
        // ---- (a) prefetch for frame i+1 into registers: 4 kNN rows, q[i+1], 8 successor rows --------
        const bool more = (i + 1 < nq);
        f32x4 pre[7];
        long my_id = 0, nn_id = 0; float my_pn = 0.f, my_f0 = 0.f; int my_kind = -1;   // per-thread scalar prefetch (tid < 13)
        {
            // row list r = 2 t + half: 0-3 -> A[i+1], 4 -> q[i+1], 5-12 -> successors of cand[0..7], 13 -> nothing.
            // Every load is UNCONDITIONAL (rows that are not needed read a harmless row and are dropped in (d)): with the loads
            // inside branches each one sat in its own basic block and waited for its predecessor — 3700 cycles per frame just to
            // issue seven loads (in-kernel cycle counters, tools/concat_prof.py), the largest single item of a 12 500-cycle frame.
            // Addresses cost no vector work either: lane r of every wave turns row r's id into a 32-bit byte offset once, the
            // offset travels to an SGPR (v_readlane) and the load is buffer_load_dwordx4 v, col * 4, rsrc, soffset.
            unsigned my_off = 0;
            if (lane < 4) my_off = more ? (unsigned)s_idNext[prv][lane] * row_bytes : 0u;
            else if (lane == 4) my_off = more ? (unsigned)(i + 1) * row_bytes : 0u;
            else if (lane < 13) { long sid = s_cand[lane - 5] + 1; sid = sid >= np ? np - 1 : sid; my_off = (unsigned)sid * row_bytes; }
            const int voff = (colok ? col : 0) * 4;
#pragma unroll
            for (int t = 0; t < 7; ++t) {
                const int r0 = 2 * t, r1 = 2 * t + 1 < 13 ? 2 * t + 1 : 12;             // this wave's row is r0 (half 0) or r1 (half 1)
                const unsigned o0 = __builtin_amdgcn_readlane(my_off, r0), o1 = __builtin_amdgcn_readlane(my_off, r1);
                const unsigned so = whalf ? o1 : o0;
#ifdef KN_CONCAT_NOLOAD        // what-if: no row loads at all (results are garbage)
                pre[t] = (f32x4){(float)so, 0.f, 0.f, 0.f};
                continue;
#endif
                if (t == 2) {                            // r0 = 4 is the query row (its own resource), r1 = 5 a pool row
                    const g2u4 vq = __builtin_amdgcn_raw_buffer_load_b128(q_rsrc, voff, o0, 0);
                    const g2u4 vp = __builtin_amdgcn_raw_buffer_load_b128(p_rsrc, voff, o1, 0);
                    pre[t] = __builtin_bit_cast(f32x4, whalf ? vp : vq);
                } else {
                    pre[t] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(p_rsrc, voff, so, 0));
                }
            }
            // The per-row scalars (norm, f0, the kNN ids two frames ahead) of lanes 0..12 of wave 0, branch-free for the same
            // reason: inside `if (tid < 4) .. else if ..` the compiler put s_waitcnt vmcnt(0) at the joins, and wave 0 sat out
            // the full latency of the row loads it had just issued — with every other wave waiting for it at the barrier.
            if (wave == 0) {                         // wave-uniform: only wave 0 keeps these values
                const int l = tid < 13 ? tid : 12;
                const long kid = more ? s_idNext[prv][l & 3] : 0;                              // kind 0: a kNN row of frame i + 1
                long sid = s_cand[l >= 5 ? l - 5 : 0] + 1; sid = sid >= np ? np - 1 : sid;      // kind 2: a successor row
                const long row = l < 4 ? kid : (l == 4 ? (more ? i + 1 : 0) : sid);
                const float* pnp = l == 4 ? qn : pn;
                const float* pfp = use_f0 ? (l == 4 ? sf0 : pf0) : pnp;                         // any valid address when f0 is off
                // issued here, first touched at the end of the frame: any use of the loaded values up here (even a select) makes
                // the compiler wait for them on the spot — behind the eight row loads, which return in order
                my_pn = pnp[row]; my_f0 = pfp[row];
                nn_id = idx_in[(i + 2 < nq ? (i + 2) * KC : 0) + (l & 3)];
                my_kind = tid < 13 ? (l < 4 ? (more ? 0 : -1) : (l == 4 ? (more ? 1 : -1) : 2)) : -1;
                my_id = (my_kind == 0 || my_kind == 2) ? row : 0;
            }
        }
        KN_TICK(0)
        // ---- (b) distances: wave b owns candidate b ---------------------------------------------------
        {
            const int coff = s_coff[wave];
            float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, a4 = 0.f, a5 = 0.f;
            for (int c = lane * 4; c < D; c += 256) {
                const f32x4 cv = *(const lf4*)&L[coff + c];
                const f32x4 qv = *(const lf4*)&L[offQ + cur * D + c];
                const f32x4 p0 = *(const lf4*)&L[offP + c], p1 = *(const lf4*)&L[offP + D + c];
                const f32x4 p2 = *(const lf4*)&L[offP + 2 * D + c], p3 = *(const lf4*)&L[offP + 3 * D + c];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float d;
                    d = qv[e] - cv[e]; a0 += d * d;
                    d = p0[e] - cv[e]; a1 += d * d;
                    d = p1[e] - cv[e]; a2 += d * d;
                    d = p2[e] - cv[e]; a3 += d * d;
                    d = p3[e] - cv[e]; a4 += d * d;
                }
                if (wave == 0) {
                    const f32x4 qp = *(const lf4*)&L[offQ + prv * D + c];
#pragma unroll
                    for (int e = 0; e < 4; ++e) { const float d = qp[e] - qv[e]; a5 += d * d; }
                }
            }
            a0 = wave_sum_dpp(a0); a1 = wave_sum_dpp(a1); a2 = wave_sum_dpp(a2); a3 = wave_sum_dpp(a3); a4 = wave_sum_dpp(a4);
            if (wave == 0) a5 = wave_sum_dpp(a5);
            if (lane == 0) {
                const float cn = s_cpn[wave];
                s_match[wave] = cos_from_cd(a0, s_qn[cur], cn);
                s_cc[0][wave] = cos_from_cd(a1, s_pnP[0], cn);
                s_cc[1][wave] = cos_from_cd(a2, s_pnP[1], cn);
                s_cc[2][wave] = cos_from_cd(a3, s_pnP[2], cn);
                s_cc[3][wave] = cos_from_cd(a4, s_pnP[3], cn);
                if (wave == 0) s_base = cos_from_cd(a5, s_qn[prv], s_qn[cur]) * 2.0f;
            }
        }
        KN_TICK(1)
        __syncthreads();
        KN_TICK(2)
        // ---- (c) costs, lower median over the previous selection, 4 smallest.  Every wave evaluates the
        // same 8-lane decision redundantly (it only reads LDS), so no barrier is needed before (d). ------
        int my_slot = -1;                       // lane < 8: rank of candidate `lane` if kept
        {
            float total = __builtin_inff();
            const float base = s_base;
            float w = s_wv[cur];
            if (use_f0 && !(base < 0.08f)) w = 0.f;
            if (lane < NC) {
                float c4[KC];
#pragma unroll
                for (int a = 0; a < KC; ++a) {
                    float c = s_cc[a][lane];
                    if (use_f0) { if (base < 0.08f && c < 5.0f * base) c = 0.f; }
                    else if (c > base) c = 1.5f * c - base;
                    c4[a] = c;
                }
                const float lo01 = fminf(c4[0], c4[1]), hi01 = fmaxf(c4[0], c4[1]);
                const float lo23 = fminf(c4[2], c4[3]), hi23 = fmaxf(c4[2], c4[3]);
                const float med = fminf(fmaxf(lo01, lo23), fminf(hi01, hi23));
                total = w * med + s_match[lane];
                if (use_f0) total = total + fabsf(s_cf0[lane] - s_sf0[cur]);      // both already log2(f0 + 1e-5)
            }
            if (!(total == total)) total = __builtin_inff();       // NaN costs rank as +inf, ties by candidate number (see above)
            int rank = 0;
#pragma unroll
            for (int j = 0; j < NC; ++j) {          // v_readlane (an SGPR broadcast), not a ds_bpermute round trip per candidate
                const float tj = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, total), j));
                rank += (tj < total || (tj == total && j < lane)) ? 1 : 0;
            }
            if (lane < NC && rank < KC) my_slot = rank;
            if (wave == 0) {
                if (my_slot >= 0) { s_slot[cur][my_slot] = lane; idx_out[i * KC + my_slot] = s_cand[lane]; }
                if (lane == 0) s_wv[prv] = w;
            }
        }
        // slot kept at rank r, broadcast inside the wave: lane holding rank r
        int kept[KC];
#pragma unroll
        for (int r = 0; r < KC; ++r) {
            const unsigned long long bal = __ballot(my_slot == r);
            kept[r] = (int)__builtin_ctzll(bal);
        }
        KN_TICK(3)
        // ---- (d) kept rows -> P, prefetched registers -> A[next], Q[next], S[cur] -----------------------
        f32x4 keep[2];
        if (colok) {
#pragma unroll
            for (int t = 0; t < 2; ++t) keep[t] = *(const lf4*)&L[s_coff[kept[2 * t + half]] + col];
        }
        if (colok) {                        // P is only read in (b); kept rows live in A[cur] / S[prv], not written here
#pragma unroll
            for (int t = 0; t < 2; ++t) *(lf4*)&L[offP + (2 * t + half) * D + col] = keep[t];
#pragma unroll
            for (int t = 0; t < 7; ++t) {
                const int r = 2 * t + half;
                if (r < 4) { if (more) *(lf4*)&L[offA + (prv * 4 + r) * D + col] = pre[t]; }
                else if (r == 4) { if (more) *(lf4*)&L[offQ + prv * D + col] = pre[t]; }
                else if (r < 13) *(lf4*)&L[offS + (cur * 8 + (r - 5)) * D + col] = pre[t];
            }
        }
        KN_TICK(4)
        if (my_kind < 0) { my_pn = 0.f; my_f0 = 0.f; }                   // lanes / rows without a row this frame: as if never loaded
        if (!use_f0) my_f0 = 0.f;
        if (use_f0 && my_kind >= 0) my_f0 = log2_rn(my_f0 + 1e-5f);      // one evaluation per new row, 13 lanes of wave 0
        // next frame's candidate table, straight from the prefetch registers of wave 0 (lanes 0..12):
        // slots 0-3 = its kNN rows, slots 4-7 = successors of the rows kept now
        {
            long nid = my_id; float npn = my_pn, nf0 = my_f0;
            int ksl = 0;
            if (tid >= KC && tid < NC) {
#pragma unroll
                for (int r = 0; r < KC; ++r) if (tid - KC == r) ksl = kept[r];
            }
            const int src = 5 + ksl;
            const long sid = __shfl(my_id, src, 64); const float spn = __shfl(my_pn, src, 64), sf = __shfl(my_f0, src, 64);
            long pid = 0; float ppn = 0.f;
            if (tid < KC) {
#pragma unroll
                for (int r = 0; r < KC; ++r) if (tid == r) { pid = s_cand[kept[r]]; ppn = s_cpn[kept[r]]; }
            }
            if (wave == 0) {
                // every lane of wave 0 has read the old table above; LDS ops of one wave execute in order
                if (tid < KC) { s_idP[tid] = pid; s_pnP[tid] = ppn; }
                if (tid < KC) { s_candT[prv][tid] = nid; s_cpnT[prv][tid] = npn; s_cf0T[prv][tid] = nf0; s_coffT[prv][tid] = offA + (prv * 4 + tid) * D; }
                else if (tid < NC) { s_candT[prv][tid] = sid; s_cpnT[prv][tid] = spn; s_cf0T[prv][tid] = sf; s_coffT[prv][tid] = offS + (cur * 8 + ksl) * D; }
            }
        }
        if (tid < 4 && i + 2 < nq) s_idNext[cur][tid] = nn_id;
        if (my_kind == 0) { s_idA[prv][tid] = my_id; s_pnA[prv][tid] = my_pn; s_f0A[prv][tid] = my_f0; }
        else if (my_kind == 1) { s_qn[prv] = my_pn; s_sf0[prv] = my_f0; }
        else if (my_kind == 2) { s_idS[cur][tid - 5] = my_id; s_pnS[cur][tid - 5] = my_pn; s_f0S[cur][tid - 5] = my_f0; }
        KN_TICK(5)
        __syncthreads();
        KN_TICK(6)
    }
#ifdef KN_CONCAT_PROF
    if (tid == 0) printf("concat prof (cycles/frame, wave 0): prefetch-issue %.0f  distances %.0f  barrier1 %.0f  select %.0f  stage(wait loads) %.0f  table %.0f  barrier2 %.0f\n",
                         (double)pf[0] / nq, (double)pf[1] / nq, (double)pf[2] / nq, (double)pf[3] / nq, (double)pf[4] / nq, (double)pf[5] / nq, (double)pf[6] / nq);
#endif
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
        size_t pl = (size_t)30 * dim * 4;
        { const char* e = getenv("KNNSVC_CONCAT_OWN_CU"); if (e && e[0] == '1' && pl < (size_t)158 * 1024) pl = (size_t)158 * 1024; }
        static size_t pattr = 0;
        if (pl > pattr) {
            if (hipFuncSetAttribute((const void*)concat_reselect_pipe_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)pl) != hipSuccess)
                return knnsvc_fail(KNNSVC_EHIP, "concat_reselect: hipFuncSetAttribute failed");
            pattr = pl;
        }
        hipLaunchKernelGGL(concat_reselect_pipe_kernel, dim3(1), dim3(CT), pl, (hipStream_t)stream, (const long*)idx_in, q,
                           q_norm, (long)nq, pool, p_norm, (long)np, dim, shifted_f0, pool_f0, use_f0, concat_weight,
                           (long*)idx_out);
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
