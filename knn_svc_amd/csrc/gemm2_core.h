// fp32 GEMM emulated on the fp16 matrix cores ("f16x2"): every fp32 operand x is pre-scaled by a
// power of two and split with round-to-nearest into two fp16 pieces  s x = hi + lo (+ <= 2^-22 |s x|),
// and a product a*b is accumulated in fp32 from three fp16 MFMAs  lo_a hi_b + hi_a lo_b + hi_a hi_b
// (the dropped lo_a lo_b term is below 2^-22 |a||b|).  Products of fp16 values are exact in fp32, so
// the result stays at fp32-GEMM accuracy (simulated against fp64 at K = 1024: 7.6e-8 rms relative
// representation error, vs 6.0e-8 for the bf16x3 split of gemm3_core.h and 3.4e-7 for an fp32 FMA
// chain) at HALF the matrix-core work of bf16x3: 3/16 of the fp32-MFMA cycles per product.
//
// Range: fp16 tops out at 65504 and goes subnormal below 6.1e-5 (v_mfma_*_f16 keeps subnormal inputs —
// tools/probe/mfma_f16_probe.hip), so both operands are moved into the middle of that range by exact
// power-of-two factors: weights by the per-tensor factor chosen when they are split (max |w| lands in
// [2^13, 2^14)), activations by a per-call factor (default 16: |x| < 4094 is representable and the
// absolute error floor is 2^-25 / 16 per element, i.e. activations whose rms is below ~0.01 start to
// lose relative accuracy — callers that know their range pass another power of two).  The epilogue
// multiplies the accumulator by 1 / (a_scale * w_scale), again exact.  An activation beyond the range
// becomes inf -> NaN in the output: loud, not silently wrong.
//
// Block = 256 threads, 4 waves (WM x WN), wave tile (32 TM) x (32 TN), K slab = 32.
// A arrives as fp32 from HBM and is split while it is staged (3 VALU per element); B (weights) is split
// once at load time into [n][K/32][2 planes][32] fp16.  LDS image per operand row: 2 planes x 64 B + 16 B
// pad = 144 B (36 dwords: the 16 rows of a ds_read_b128 lane group land on 16 distinct 4-bank slots).
#pragma once
#include "common.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef unsigned g2_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned g2_u32x2 __attribute__((ext_vector_type(2)));

constexpr float KN_F16X2_A_SCALE = 16.0f;

// 4 fp32 -> (hi, lo) planes of 4 fp16 each, as two dwords per plane
__device__ __forceinline__ void f16x2_split4(f32x4 v, float scale, g2_u32x2& hi, g2_u32x2& lo) {
    f32x2_t a = {v[0], v[1]}, b = {v[2], v[3]};
    a *= scale; b *= scale;
    const f16x2_t ha = __builtin_convertvector(a, f16x2_t), hb = __builtin_convertvector(b, f16x2_t);
    const f32x2_t ra = a - __builtin_convertvector(ha, f32x2_t), rb = b - __builtin_convertvector(hb, f32x2_t);
    const f16x2_t la = __builtin_convertvector(ra, f16x2_t), lb = __builtin_convertvector(rb, f16x2_t);
    hi = (g2_u32x2){__builtin_bit_cast(unsigned, ha), __builtin_bit_cast(unsigned, hb)};
    lo = (g2_u32x2){__builtin_bit_cast(unsigned, la), __builtin_bit_cast(unsigned, lb)};
}

// XCD-aware tile order of the 256x256-tile kernels.  Workgroup ids go round-robin over the 8 XCDs (each with its own L2), so one
// tile dimension is dealt over the XCDs in groups of 8 (padded: ids whose tile falls into the padding exit at once) and the other
// is walked in patches of CW = 4: an XCD's resident blocks then share operand panels through its L2.  Which dimension is padded
// matters when it is short: 1500 query rows are 6 row tiles — padded to 8, the workgroups of TWO XCDs would all be padding and a
// quarter of the chip would idle through the whole launch (the kNN at the north-star point); so the dimension whose padding
// wastes less is the one dealt over the XCDs (`swap`: the columns).  Host (grid size) and device (decode) use the same rule.
__host__ __device__ __forceinline__ bool quad_order_swap(long gx, long gy) {
    const long gx8 = (gx + 7) / 8 * 8, gy8 = (gy + 7) / 8 * 8;
    // rows unless their padding is more than 1/16 of the ids AND the columns' padding is smaller (long M: the padding is the tail
    // of the last group only, nothing idles for long)
    return (gx8 - gx) * 16 > gx8 && (gx8 - gx) * gy8 > (gy8 - gy) * gx8;
}
// Small grids where EITHER dimension would be padded by more than 1/16: plain order, no padding ids.  A padding id is not free — the
// workgroup needs its 128 KB of LDS before it can run and exit, workgroups are dispatched in order, and id i goes to XCD i mod 8: an
// XCD (32 CUs, one workgroup each) that is handed 42 ids runs two rounds even if a quarter of them are padding.  The feature
// extractor's last layers (1500 / 3000 rows per chunk = 6 / 12 row tiles x 2 column tiles x 21 chunks) ran 2 / 3 rounds that way;
// unpadded they are 252 / 504 ids = 1 / 2 rounds.  (L2 panel sharing is moot at these sizes — NOT for the long layers: the unpadded
// order on the 48 007-row layer saves a round of 32 and still loses, 3.97 vs 3.85 ms, same box.)
__host__ __device__ __forceinline__ bool quad_order_linear(long gx, long gy) {
    const long gx8 = (gx + 7) / 8 * 8, gy8 = (gy + 7) / 8 * 8;
    return (gx8 - gx) * 16 > gx8 && (gy8 - gy) * 16 > gy8;
}
__host__ __device__ __forceinline__ long quad_order_ids(long gx, long gy) {
    if (quad_order_linear(gx, gy)) return gx * gy;
    return quad_order_swap(gx, gy) ? (gy + 7) / 8 * 8 * gx : (gx + 7) / 8 * 8 * gy;
}
// rot: rotates which XCD gets which row of a group of 8 (a batched launch passes its batch index): every batch slice has the same
// padding rows, and unrotated they would starve the same XCDs in every slice — the feature extractor's last layers (1500 / 3000
// rows per chunk = 6 / 12 row tiles, padded to 8 / 16, x 21 chunks) ran 2 and 3 rounds of tiles on six / four busy XCDs instead of
// 1 and 2 on eight.
__device__ __forceinline__ bool quad_order_decode(int L, int gx, int gy, int& mt, int& nt, int rot = 0) {
    if (quad_order_linear(gx, gy)) { mt = L % gx; nt = L / gx; return true; }
    const bool sw = quad_order_swap(gx, gy);
    const int ga = sw ? gy : gx, gb = sw ? gx : gy;            // ga: dealt over the XCDs
    const int ga8 = (ga + 7) / 8 * 8;
    constexpr int CW = 4;
    const int full = (gb / CW) * CW * ga8;
    int c0, cw;
    if (L < full) { c0 = (L / (CW * ga8)) * CW; cw = CW; L -= (c0 / CW) * CW * ga8; }
    else { c0 = (gb / CW) * CW; cw = gb - c0; L -= full; }
    const int grp = L / (8 * cw), rem = L - grp * 8 * cw;
    const int at = grp * 8 + ((rem + rot) & 7), bt = c0 + (rem >> 3);
    mt = sw ? bt : at; nt = sw ? at : bt;
    return at < ga;
}

#ifdef KN_QUAD_PROF      // timing aid (tools/quad_prof.py): per block start / prologue done / main loop done / epilogue done, 10 ns ticks
__device__ long long kn_quad_prof_buf[8192 * 4];
#endif

template <int BM_, int BN_, int WM_, int WN_, int TM_, int TN_>
struct Gemm2Tile {
    typedef f32x16 acc_t;
    static constexpr int NR = 16;               // accumulator elements per MFMA tile and lane
    static constexpr int BM = BM_, BN = BN_, WM = WM_, WN = WN_, TM = TM_, TN = TN_;
    static constexpr int BK = 32, PITCH = 144, THREADS = 256;
    static_assert(WM * WN == 4 && WM * TM * 32 == BM && WN * TN * 32 == BN, "tile shape");
    static constexpr int A_F4 = BM / 32;                       // fp32 float4 per thread per slab
    static constexpr int B_PIECES = BN * 8;                    // 16-byte pieces of split weights per slab
    static constexpr int B_P = (B_PIECES + 255) / 256;         // per thread
    static constexpr int LDS_BYTES = (BM + BN) * PITCH;

    typedef __attribute__((address_space(3))) char lds_c;

    __device__ __forceinline__ static void split_store(lds_c* dst, f32x4 v, float a_scale) {
        g2_u32x2 hi, lo;
        f16x2_split4(v, a_scale, hi, lo);
        typedef __attribute__((address_space(3))) g2_u32x2 lds_u2;
        *(lds_u2*)(dst) = hi;
        *(lds_u2*)(dst + 64) = lo;
    }

    // aload: fp32 A loader (begin(kt), operator()(kt, j) -> f32x4 for row (tid>>3)+32j, k = (tid&7)*4, finish())
    // bload: split-weight loader (begin(kt), operator()(kt, j) -> 16-byte piece j of this thread)
    // A2 = true: the activations already are in the split layout ([rows][C/32][2][32] fp16, the same bytes per element
    // as fp32, written by the producing kernel with the default activation scale): the 16 bytes a thread loads are
    // one finished piece of its row's LDS image and staging is a plain copy.
    template <bool A2 = false, class ALoad, class BLoad, class RA, class RB>
    __device__ __forceinline__ static void mainloop(float* lds_generic, int nk, ALoad& aload, BLoad& bload,
                                                    f32x16 (&acc)[TM][TN], RA ra_desc, RB rb_desc, float a_scale) {
        lds_c* lds = (lds_c*)lds_generic;
        typedef __attribute__((address_space(3))) g2_u32x4 lds_u4;
        const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
        const int wm = wave / WN, wn = wave % WN;
        const int a_st = (tid >> 3) * PITCH + (tid & 7) * (A2 ? 16 : 8);   // staging address of piece j = 0
        const int li = lane & 31, lh = lane >> 5;
        const int a_frag = (wm * TM * 32 + li) * PITCH + lh * 16;
        const int b_frag = BM * PITCH + (wn * TN * 32 + li) * PITCH + lh * 16;

        f32x4 ra[A_F4];
        g2_u32x4 rb[B_P];
#ifdef KN_T_NOSTAGE_A
#define KN_STAGE2_A() _Pragma("unroll") for (int j = 0; j < A_F4; ++j) asm volatile("" ::"v"(ra[j][0]), "v"(ra[j][1]), "v"(ra[j][2]), "v"(ra[j][3]));
#else
#define KN_STAGE2_A()                                                                                         \
        _Pragma("unroll") for (int j = 0; j < A_F4; ++j) {                                                   \
            if constexpr (A2) *(lds_u4*)(lds + a_st + 32 * j * PITCH) = __builtin_bit_cast(g2_u32x4, ra[j]); \
            else split_store(lds + a_st + 32 * j * PITCH, aload.finish(ra[j]), a_scale);                     \
        }
#endif
#ifdef KN_T_NOSTAGE_B
#define KN_STAGE2_B() _Pragma("unroll") for (int j = 0; j < B_P; ++j) asm volatile("" ::"v"(rb[j][0]), "v"(rb[j][1]), "v"(rb[j][2]), "v"(rb[j][3]));
#else
#define KN_STAGE2_B()                                                                                         \
        _Pragma("unroll") for (int j = 0; j < B_P; ++j) {                                                    \
            const int q = tid + 256 * j;                                                                     \
            if (B_PIECES % 256 == 0 || q < B_PIECES)                                                         \
                *(lds_u4*)(lds + BM * PITCH + (q >> 3) * PITCH + (q & 7) * 16) = rb[j];                       \
        }
#endif
#define KN_STAGE2() { KN_STAGE2_A() KN_STAGE2_B() }
        aload.begin(0); bload.begin(0);
#pragma unroll
        for (int j = 0; j < A_F4; ++j) ra[j] = aload(0, j, ra_desc);
#pragma unroll
        for (int j = 0; j < B_P; ++j) rb[j] = bload(0, j, rb_desc);
        KN_STAGE2();
        __syncthreads();

        for (int kt = 0; kt < nk; ++kt) {
            const bool more = (kt + 1 < nk);
            if (more) {
                aload.begin(kt + 1); bload.begin(kt + 1);
#ifndef KN_T_NOLOAD_A
#pragma unroll
                for (int j = 0; j < A_F4; ++j) ra[j] = aload(kt + 1, j, ra_desc);
#endif
#ifndef KN_T_NOLOAD_B
#pragma unroll
                for (int j = 0; j < B_P; ++j) rb[j] = bload(kt + 1, j, rb_desc);
#endif
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                f16x8 fa[TM][2], fb[TN][2];
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int p = 0; p < 2; ++p)
                        fa[i][p] = __builtin_bit_cast(f16x8, *(const lds_u4*)(lds + a_frag + i * 32 * PITCH + p * 64 + ks * 32));
#pragma unroll
                for (int i = 0; i < TN; ++i)
#pragma unroll
                    for (int p = 0; p < 2; ++p)
                        fb[i][p] = __builtin_bit_cast(f16x8, *(const lds_u4*)(lds + b_frag + i * 32 * PITCH + p * 64 + ks * 32));
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        f32x16 c = acc[i][j];
                        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[i][1], fb[j][0], c, 0, 0, 0);   // small terms first
                        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[i][0], fb[j][1], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[i][0], fb[j][0], c, 0, 0, 0);
                        acc[i][j] = c;
                    }
            }
            __syncthreads();                 // every wave is done reading this slab
            if (more) KN_STAGE2();
            __syncthreads();
        }
    }

#undef KN_STAGE2
#undef KN_STAGE2_A
#undef KN_STAGE2_B
    __device__ __forceinline__ static int acc_row(int wave, int lane, int i, int r) {
        return (wave / WN) * TM * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    }
    __device__ __forceinline__ static int acc_col(int wave, int lane, int j) {
        return (wave % WN) * TN * 32 + j * 32 + (lane & 31);
    }
};

// -------------------------------------------------------------------------------------------------
// Gemm2QuadS: the same 256x256 block / 128x128 wave tile on v_mfma_f32_16x16x32_f16 (the hardware guide measures 1.12-1.15x
// the FLOP/s of 32x32x16 at equal cycles on random data: the gain is clock, i.e. energy per product — exactly what bounds
// Gemm2QuadR).  A k slab is 32 wide (one MFMA deep): LDS row = 128 B (hi 64 | lo 64) in 16-byte chunks, chunk' = chunk ^ (row & 7)
// (conflict-free for the 16-row x 4-chunk fragment reads and for the 8-row x 8-chunk staging writes); two 64 KB stages.
// The 8 x 8 tiles of a wave are walked in four 4 x 4 quadrants per slab; fragments live in two A sets and two B sets (32 VGPRs
// each).  Slab order alternates (I0,J0)(I0,J1)(I1,J1)(I1,J0) / (I0,J1)(I0,J0)(I1,J0)(I1,J1): the A set and the B set a slab
// starts with are exactly the ones the previous slab retired first, so they are prefetched during its last quadrants.
// Per quadrant (16 tile steps of 3 MFMAs) one memory instruction rides behind each tile step:
//   q0  steps 0-7: read the other B set        steps 8-15: write A pieces of slab s+1 (loaded during q2 of slab s-1)
//   q1  steps 0-7: load B pieces of slab s+1   steps 8-15: read the second A set
//   q2  steps 8-15: write B pieces of s+1, load A pieces of s+2;  then lgkmcnt(0) + barrier (slab s+1 is complete in LDS)
//   q3  steps 0-15: read the first A and B sets of slab s+1
// -------------------------------------------------------------------------------------------------
struct Gemm2QuadS {
    typedef f32x4 acc_t;
    static constexpr int NR = 4;
    static constexpr int BM = 256, BN = 256, WM = 2, WN = 2, TM = 8, TN = 8;
    static constexpr int EPI_TN = 4;                                            // wave tile width in 32-column units (epilogue patch)
    static constexpr int ROW = 128, NW = 4, THREADS = 256;
    static constexpr int A_P = BM * ROW / 1024 / NW, B_P = BN * ROW / 1024 / NW;   // 8 + 8 pieces per wave and slab
    static_assert(A_P == 8 && B_P == 8, "schedule below is written for 8 + 8 pieces");
    static constexpr int BOFF = BM * ROW;
    static constexpr int STAGE = (BM + BN) * ROW;
    static constexpr int EPI_BYTES = NW * 32 * (EPI_TN * 32 + 4) * 4;
    static constexpr int LDS_BYTES = 2 * STAGE > EPI_BYTES ? 2 * STAGE : EPI_BYTES;
    static constexpr int OOB_OFF = 0x40000000;

    typedef __attribute__((address_space(3))) char lds_c;
    typedef __attribute__((address_space(3))) g2_u32x4 lds_u4;

    template <class RowOff, class Step, class RA, class RB>
    __device__ __forceinline__ static void mainloop(float* lds_generic, int nk, RowOff a_row_off, Step a_step, RA ra_desc,
                                                    RB rb_desc, int N, int K, int m0, int n0, f32x4 (&acc)[TM][TN]) {
        lds_c* lds = (lds_c*)lds_generic;
        const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int wm = wave / WN, wn = wave % WN;
        const int a_row = (wm * 128 + (lane & 15)) * ROW;
        const int b_row = BOFF + (wn * 128 + (lane & 15)) * ROW;
        int x_off[2];
#pragma unroll
        for (int p = 0; p < 2; ++p) x_off[p] = ((p * 4 + (lane >> 4)) ^ (lane & 7)) * 16;
        int a_src[A_P], b_src[B_P];
        const int row_bytes = (K / 32) * 128;
#pragma unroll
        for (int t = 0; t < A_P; ++t) {
            const int row = (wave * A_P + t) * 8 + (lane >> 3);
            const int ro = a_row_off(m0 + row);
            a_src[t] = ro == OOB_OFF ? OOB_OFF : ro + (lane & 7) * 16;
        }
#pragma unroll
        for (int t = 0; t < B_P; ++t) {
            const int row = (wave * B_P + t) * 8 + (lane >> 3);
            b_src[t] = (n0 + row < N) ? (n0 + row) * row_bytes + (lane & 7) * 16 : OOB_OFF;
        }
        const int st_dst = wave * 8 * 1024 + (lane >> 3) * ROW + (((lane & 7) ^ (lane >> 3)) * 16);
        int ua = 0;
        int a_issued = 0, b_issued = 0;                             // slabs whose A / B pieces have been requested
        g2_u32x4 st[8];
#define KN_S_LOAD_A(T) { st[T] = __builtin_amdgcn_raw_buffer_load_b128(ra_desc, a_src[T] + ua, 0, 0); }
#define KN_S_LOAD_B(T) { st[T] = __builtin_amdgcn_raw_buffer_load_b128(rb_desc, b_src[T], b_issued * 128, 0); }
#define KN_S_WRITE(T, OFF) { *(lds_u4*)(lds + (OFF) + st_dst + (T) * 1024) = st[T]; }
        f16x8 fa[2][4][2], fb[2][4][2];
#define KN_S_READ_A(SET, STG, II, I) { _Pragma("unroll") for (int p = 0; p < 2; ++p) fa[SET][I][p] = __builtin_bit_cast(f16x8, *(const lds_u4*)(lds + (STG) + a_row + ((II) * 4 + (I)) * 16 * ROW + x_off[p])); }
#define KN_S_READ_B(SET, STG, JJ, J) { _Pragma("unroll") for (int p = 0; p < 2; ++p) fb[SET][J][p] = __builtin_bit_cast(f16x8, *(const lds_u4*)(lds + (STG) + b_row + ((JJ) * 4 + (J)) * 16 * ROW + x_off[p])); }
// In-place accumulation spelled out: left to itself the register allocator shuffles the 64 four-register accumulator tiles
// through temporaries (8 v_accvgpr_mov per tile step and spills); "+a" ties each tile to its AGPRs.
#define KN_S_MFMA1(C, A, B) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(C) : "v"(A), "v"(B));
#define KN_S_MFMA(AS, BS, I, J)                                                                                           \
    {                                                                                                                     \
        KN_S_MFMA1(acc[(AS) * 4 + (I)][(BS) * 4 + (J)], fa[AS][I][1], fb[BS][J][0])             /* small terms first */    \
        KN_S_MFMA1(acc[(AS) * 4 + (I)][(BS) * 4 + (J)], fa[AS][I][0], fb[BS][J][1])                                       \
        KN_S_MFMA1(acc[(AS) * 4 + (I)][(BS) * 4 + (J)], fa[AS][I][0], fb[BS][J][0])                                       \
    }
        // prologue: slab 0 -> stage 0 (both operands), A pieces of slab 1 in the staging registers, first sets of slab 0
        ua = a_step(0);
#pragma unroll
        for (int t = 0; t < 8; ++t) KN_S_LOAD_A(t)
        a_issued = 1;
#pragma unroll
        for (int t = 0; t < 8; ++t) KN_S_WRITE(t, 0)
#pragma unroll
        for (int t = 0; t < 8; ++t) KN_S_LOAD_B(t)
        b_issued = 1;
#pragma unroll
        for (int t = 0; t < 8; ++t) KN_S_WRITE(t, BOFF)
        if (nk > 1) {
            ua = a_step(1);
#pragma unroll
            for (int t = 0; t < 8; ++t) KN_S_LOAD_A(t)
            a_issued = 2;
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_s_barrier();
#ifdef KN_QUAD_PROF
        if (threadIdx.x == 0 && blockIdx.x < 8192) kn_quad_prof_buf[blockIdx.x * 4 + 1] = (long long)__builtin_amdgcn_s_memrealtime();
#endif
#pragma unroll
        for (int i = 0; i < 4; ++i) KN_S_READ_A(0, 0, 0, i)
#pragma unroll
        for (int j = 0; j < 4; ++j) KN_S_READ_B(0, 0, 0, j)

        // FB: the B set (= column half) this slab starts with.  HN: slab S + 1 exists.  ML: slab S + 2 exists.  All literals
        // (see Gemm2QuadR: a run-time test inside the slab costs a vmcnt(0) before every ds_write).
#define KN_S_SLAB(FB, HN, ML)                                                                                             \
    {                                                                                                                     \
        constexpr int cstg = (FB) * STAGE, nstg = (1 - (FB)) * STAGE;       /* even slabs start with column half 0 */       \
        /* q0: (I0, J[FB]) */                                                                                             \
        _Pragma("unroll") for (int t = 0; t < 16; ++t) {                                                                  \
            KN_S_MFMA(0, FB, t >> 2, t & 3)                                                                               \
            __builtin_amdgcn_sched_barrier(0);                                                                            \
            if (t < 8) { if (t < 4) KN_S_READ_B(1 - (FB), cstg, 1 - (FB), t) }                                            \
            else if (HN) KN_S_WRITE(t - 8, nstg)                                                                          \
            __builtin_amdgcn_sched_barrier(0);                                                                            \
        }                                                                                                                 \
        /* q1: (I0, J[1-FB]) */                                                                                           \
        _Pragma("unroll") for (int t = 0; t < 16; ++t) {                                                                  \
            KN_S_MFMA(0, 1 - (FB), t >> 2, t & 3)                                                                         \
            __builtin_amdgcn_sched_barrier(0);                                                                            \
            if (t < 8) { if (HN) KN_S_LOAD_B(t) }                                                                         \
            else if (t < 12) KN_S_READ_A(1, cstg, 1, t - 8)                                                               \
            __builtin_amdgcn_sched_barrier(0);                                                                            \
        }                                                                                                                 \
        if (HN) ++b_issued;                                                                                               \
        if (ML) ua = a_step(a_issued);                                                                                    \
        /* q2: (I1, J[1-FB]) */                                                                                           \
        _Pragma("unroll") for (int t = 0; t < 16; ++t) {                                                                  \
            KN_S_MFMA(1, 1 - (FB), t >> 2, t & 3)                                                                         \
            __builtin_amdgcn_sched_barrier(0);                                                                            \
            if (t >= 8) {                                                                                                 \
                if (HN) KN_S_WRITE(t - 8, nstg + BOFF)                                                                    \
                if (ML) KN_S_LOAD_A(t - 8)                                                                                \
            }                                                                                                             \
            __builtin_amdgcn_sched_barrier(0);                                                                            \
        }                                                                                                                 \
        if (ML) ++a_issued;                                                                                               \
        if (HN) { __builtin_amdgcn_s_waitcnt(0xC07F); __builtin_amdgcn_s_barrier(); }                                     \
        /* q3: (I1, J[FB]); prefetch the sets slab S + 1 starts with: A_I0 -> set 0, B_J[1-FB] -> set 1-FB */             \
        _Pragma("unroll") for (int t = 0; t < 16; ++t) {                                                                  \
            KN_S_MFMA(1, FB, t >> 2, t & 3)                                                                               \
            __builtin_amdgcn_sched_barrier(0);                                                                            \
            if (HN) {                                                                                                     \
                if (t < 4) KN_S_READ_A(0, nstg, 0, t)                                                                     \
                else if (t >= 8 && t < 12) KN_S_READ_B(1 - (FB), nstg, 1 - (FB), t - 8)                                   \
            }                                                                                                             \
            __builtin_amdgcn_sched_barrier(0);                                                                            \
        }                                                                                                                 \
    }
        // Every slab runs the full schedule: past the last slab the loads fall outside their buffer resources (zeros), land in
        // the idle stage and are never multiplied — one straight-line loop body, no tail variants for the register allocator
        // to reconcile (with peeled tails the 64 accumulator tiles were copied and spilled at every block boundary).
        int s = 0;
        for (; s + 1 < nk; s += 2) {
            KN_S_SLAB(0, true, true)
            KN_S_SLAB(1, true, true)
        }
        if (s < nk) KN_S_SLAB(0, true, true)
#undef KN_S_SLAB
#undef KN_S_MFMA
#undef KN_S_MFMA1
#undef KN_S_READ_A
#undef KN_S_READ_B
#undef KN_S_WRITE
#undef KN_S_LOAD_A
#undef KN_S_LOAD_B
    }

    // C of a 16x16 tile: col = lane & 15, row = 4 (lane >> 4) + reg
    __device__ __forceinline__ static int acc_row(int wave, int lane, int i, int r) { return (wave / WN) * 128 + i * 16 + 4 * (lane >> 4) + r; }
    __device__ __forceinline__ static int acc_col(int wave, int lane, int j) { return (wave % WN) * 128 + j * 16 + (lane & 15); }

};

// -------------------------------------------------------------------------------------------------
// Windowed variant for stride-1 convolutions with several taps (HiFi-GAN ResBlock convs k = 3/7/11 with dilation,
// WavLM's k = 128 positional conv).  The implicit-GEMM kernels above walk K tap-major and re-stage, for every tap, the
// same input rows shifted by `dil` (a what-if build without A staging ran the generator in 7.6 instead of 10.0 ms).
// Here K is walked CHANNEL-SLAB-major: for each slab of 32 input channels the block stages ONE window of
// BM + HALO input rows (fp32 -> f16x2 split, 144-byte pitch as Gemm2Tile) and every tap reads its A fragments from
// that window at a row offset of tap * dil — A is staged once instead of `taps` times and only the weights stream.
// The next slab's window travels in registers during the tap loop (W_F4 float4 per thread).  Sums over K are taken in
// a different order than in Gemm2Tile (slab-major instead of tap-major): equal up to fp32 rounding of the accumulation.
// -------------------------------------------------------------------------------------------------
template <int BM_, int BN_, int WM_, int WN_, int TM_, int TN_, int HALO_, int DEEP_ = 0>
struct Gemm2Win {
    static constexpr int DEEP = DEEP_;          // 1: weight slabs requested two steps ahead (mainloop_deep)
    typedef f32x16 acc_t;
    static constexpr int NR = 16;
    static constexpr int BM = BM_, BN = BN_, WM = WM_, WN = WN_, TM = TM_, TN = TN_, HALO = HALO_;
    static constexpr int BK = 32, PITCH = 144, THREADS = 256;
    static_assert(WM * WN == 4 && WM * TM * 32 == BM && WN * TN * 32 == BN && HALO % 32 == 0, "tile shape");
    static constexpr int WR = BM + HALO;                       // window rows
    static constexpr int W_F4 = WR / 32;                       // fp32 float4 per thread per window
    static constexpr int B_PIECES = BN * 8;
    static constexpr int B_P = (B_PIECES + 255) / 256;
    static constexpr int BOFF = WR * PITCH;
    static constexpr int LDS_BYTES = (WR + BN) * PITCH;

    typedef __attribute__((address_space(3))) char lds_c;
    typedef __attribute__((address_space(3))) g2_u32x4 lds_u4;
    typedef __attribute__((address_space(3))) g2_u32x2 lds_u2;

    // w_off0: this thread's byte offset of (window row tid>>3, channel (tid&7)*4) in the A buffer resource (may be
    // "negative": conv padding rows fall outside the resource and read as zeros); row_bytes = ldx * 4.
    // bload: Split2BLoader; slab index of (tap, cs) in the tap-major weight image = tap * ncs + cs.
    template <class BLoad, class RA, class RB>
    __device__ __forceinline__ static void mainloop(float* lds_generic, int ncs, int taps, int dil, int w_off0, int row_bytes,
                                                    BLoad& bload, f32x16 (&acc)[TM][TN], RA ra_desc, RB rb_desc,
                                                    float a_scale, float a_slope) {
        lds_c* lds = (lds_c*)lds_generic;
        const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
        const int wm = wave / WN, wn = wave % WN;
        const int w_st = (tid >> 3) * PITCH + (tid & 7) * 8;
        const int li = lane & 31, lh = lane >> 5;
        const int a_frag = (wm * TM * 32 + li) * PITCH + lh * 16;
        const int b_frag = BOFF + (wn * TN * 32 + li) * PITCH + lh * 16;
        const int tap_step = dil * PITCH;

        f32x4 rw[W_F4];
        g2_u32x4 rb[B_P];
#define KN_LOAD_W(CS)                                                                                          \
    _Pragma("unroll") for (int j = 0; j < W_F4; ++j)                                                          \
        rw[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ra_desc, w_off0 + j * 32 * row_bytes + (CS) * 128, 0, 0));
#define KN_STAGE_W()                                                                                           \
    _Pragma("unroll") for (int j = 0; j < W_F4; ++j) {                                                        \
        f32x4 v = rw[j];                                                                                      \
        if (a_slope != 1.0f) { _Pragma("unroll") for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * a_slope; } \
        g2_u32x2 hi, lo;                                                                                      \
        f16x2_split4(v, a_scale, hi, lo);                                                                     \
        *(lds_u2*)(lds + w_st + 32 * j * PITCH) = hi;                                                         \
        *(lds_u2*)(lds + w_st + 32 * j * PITCH + 64) = lo;                                                    \
    }
#define KN_LOAD_B(SLAB)                                                                                        \
    { bload.begin(SLAB); _Pragma("unroll") for (int j = 0; j < B_P; ++j) rb[j] = bload(SLAB, j, rb_desc); }
#define KN_STAGE_B()                                                                                           \
    _Pragma("unroll") for (int j = 0; j < B_P; ++j) {                                                         \
        const int q = tid + 256 * j;                                                                          \
        if (B_PIECES % 256 == 0 || q < B_PIECES) *(lds_u4*)(lds + BOFF + (q >> 3) * PITCH + (q & 7) * 16) = rb[j]; \
    }
#ifdef KN_WIN_ROT
        const int rot = (int)(blockIdx.x % (unsigned)ncs);
#define KN_CS(I) (((I) + rot) % ncs)
#else
#define KN_CS(I) (I)
#endif
        KN_LOAD_W(KN_CS(0))
        KN_LOAD_B(KN_CS(0))
        KN_STAGE_W()
        KN_STAGE_B()
        __syncthreads();
        for (int ci = 0; ci < ncs; ++ci) {
            const bool more_cs = ci + 1 < ncs;
            const int cs = KN_CS(ci);
            if (more_cs) { KN_LOAD_W(KN_CS(ci + 1)) }
            int tap_off = 0;
            for (int tap = 0; tap < taps; ++tap, tap_off += tap_step) {
                const bool last_tap = tap + 1 == taps;
                const bool more = !last_tap || more_cs;
#ifndef KN_WIN_NOBLOAD          // timing aid: the first weight slab serves every step
                if (more) {
                    const int slab = last_tap ? KN_CS(ci + 1) : ((tap + 1) * ncs + cs);
                    KN_LOAD_B(slab)
                }
#endif
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    f16x8 fa[TM][2], fb[TN][2];
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int p = 0; p < 2; ++p)
                            fa[i][p] = __builtin_bit_cast(f16x8, *(const lds_u4*)(lds + a_frag + tap_off + i * 32 * PITCH + p * 64 + ks * 32));
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int p = 0; p < 2; ++p)
                            fb[j][p] = __builtin_bit_cast(f16x8, *(const lds_u4*)(lds + b_frag + j * 32 * PITCH + p * 64 + ks * 32));
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j) {
                            f32x16 c = acc[i][j];
#ifdef KN_WHATIF_NOMFMA
                            c[0] += (float)fa[i][1][0] * (float)fb[j][0][0] + (float)fa[i][0][1] * (float)fb[j][1][1];
#else
                            c = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[i][1], fb[j][0], c, 0, 0, 0);   // small terms first
                            c = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[i][0], fb[j][1], c, 0, 0, 0);
                            c = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[i][0], fb[j][0], c, 0, 0, 0);
#endif
                            acc[i][j] = c;
                        }
                }
#ifndef KN_WIN_NOBAR            // timing aid: no barriers inside the loop
                __syncthreads();                 // every wave is done with this weight slab (and, on the last tap, the window)
#endif
#ifndef KN_WIN_NOBLOAD
                if (more) { KN_STAGE_B() }
#endif
                if (last_tap && more_cs) { KN_STAGE_W() }
#ifndef KN_WIN_NOBAR
                __syncthreads();
#endif
            }
        }
#undef KN_CS
#undef KN_LOAD_W
#undef KN_STAGE_W
#undef KN_LOAD_B
#undef KN_STAGE_B
    }

    // The same loop with the weight slabs requested TWO steps ahead (DEEP = 1: the 64-row tiles of under-filled launches).  In
    // `mainloop` the slab of step s + 1 is requested at the top of step s and staged at its end: one step (~400 cycles of MFMAs) to
    // cover an L2 round trip, and on a grid that leaves one workgroup per CU nothing else runs meanwhile — a what-if build without
    // the weight loads ran the generator's C = 256, k = 11 convolution in 71 instead of 91 us, and dataset mode on 5-10 s utterances
    // spends more GPU time in these launches than in anything else.  Here two register sets alternate: the slab staged at the end
    // of step s was requested at the end of step s - 2.  Steps are walked flat (channel slab outer, tap inner, as `mainloop`: the
    // same products in the same order); the window of the next channel slab travels as before.
    template <class BLoad, class RA, class RB>
    __device__ __forceinline__ static void mainloop_deep(float* lds_generic, int ncs, int taps, int dil, int w_off0, int row_bytes,
                                                         BLoad& bload, f32x16 (&acc)[TM][TN], RA ra_desc, RB rb_desc,
                                                         float a_scale, float a_slope) {
        lds_c* lds = (lds_c*)lds_generic;
        const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
        const int wm = wave / WN, wn = wave % WN;
        const int w_st = (tid >> 3) * PITCH + (tid & 7) * 8;
        const int li = lane & 31, lh = lane >> 5;
        const int a_frag = (wm * TM * 32 + li) * PITCH + lh * 16;
        const int b_frag = BOFF + (wn * TN * 32 + li) * PITCH + lh * 16;
        const int tap_step = dil * PITCH;
        const int nsteps = ncs * taps;
        f32x4 rw[W_F4];
        g2_u32x4 rb[2][B_P];
        // slab of flat step s in the tap-major weight image: tap * ncs + cs, with s = cs * taps + tap
        auto slab_of = [&](int s) { const int c = s / taps; return (s - c * taps) * ncs + c; };
#define KN_D_LOAD_W(CS)                                                                                        \
    _Pragma("unroll") for (int j = 0; j < W_F4; ++j)                                                          \
        rw[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ra_desc, w_off0 + j * 32 * row_bytes + (CS) * 128, 0, 0));
#define KN_D_STAGE_W()                                                                                         \
    _Pragma("unroll") for (int j = 0; j < W_F4; ++j) {                                                        \
        f32x4 v = rw[j];                                                                                      \
        if (a_slope != 1.0f) { _Pragma("unroll") for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * a_slope; } \
        g2_u32x2 hi, lo;                                                                                      \
        f16x2_split4(v, a_scale, hi, lo);                                                                     \
        *(lds_u2*)(lds + w_st + 32 * j * PITCH) = hi;                                                         \
        *(lds_u2*)(lds + w_st + 32 * j * PITCH + 64) = lo;                                                    \
    }
#define KN_D_LOAD_B(SET, S)                                                                                    \
    { const int sl_ = slab_of(S); bload.begin(sl_); _Pragma("unroll") for (int j = 0; j < B_P; ++j) rb[SET][j] = bload(sl_, j, rb_desc); }
#define KN_D_STAGE_B(SET)                                                                                      \
    _Pragma("unroll") for (int j = 0; j < B_P; ++j) {                                                         \
        const int q = tid + 256 * j;                                                                          \
        if (B_PIECES % 256 == 0 || q < B_PIECES) *(lds_u4*)(lds + BOFF + (q >> 3) * PITCH + (q & 7) * 16) = rb[SET][j]; \
    }
        KN_D_LOAD_W(0)
        KN_D_LOAD_B(0, 0)
        KN_D_STAGE_W()
        KN_D_STAGE_B(0)
        if (nsteps > 1) KN_D_LOAD_B(1, 1)              // step s lives in set s & 1
        if (nsteps > 2) KN_D_LOAD_B(0, 2)
        __syncthreads();
        int cs = 0, tap = 0, tap_off = 0;
        // one step; SET = parity of step s + 1, whose slab is staged at the end of step s and whose set is reloaded with step s + 3
#define KN_D_STEP(SET)                                                                                         \
    {                                                                                                          \
        const bool last_tap = tap + 1 == taps;                                                                 \
        const bool more_cs = cs + 1 < ncs;                                                                     \
        if (tap == 0 && more_cs) { KN_D_LOAD_W(cs + 1) }                                                       \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                                    \
            f16x8 fa[TM][2], fb[TN][2];                                                                        \
            _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                    \
                _Pragma("unroll") for (int p = 0; p < 2; ++p)                                                 \
                    fa[i][p] = __builtin_bit_cast(f16x8, *(const lds_u4*)(lds + a_frag + tap_off + i * 32 * PITCH + p * 64 + ks * 32)); \
            _Pragma("unroll") for (int j = 0; j < TN; ++j)                                                    \
                _Pragma("unroll") for (int p = 0; p < 2; ++p)                                                 \
                    fb[j][p] = __builtin_bit_cast(f16x8, *(const lds_u4*)(lds + b_frag + j * 32 * PITCH + p * 64 + ks * 32)); \
            _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                    \
                _Pragma("unroll") for (int j = 0; j < TN; ++j) {                                              \
                    f32x16 c = acc[i][j];                                                                      \
                    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[i][1], fb[j][0], c, 0, 0, 0);               \
                    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[i][0], fb[j][1], c, 0, 0, 0);               \
                    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[i][0], fb[j][0], c, 0, 0, 0);               \
                    acc[i][j] = c;                                                                             \
                }                                                                                              \
        }                                                                                                      \
        __syncthreads();                                                                                       \
        if (s + 1 < nsteps) { KN_D_STAGE_B(SET) }                                                              \
        if (s + 3 < nsteps) { KN_D_LOAD_B(SET, s + 3) }                                                        \
        if (last_tap && more_cs) { KN_D_STAGE_W() }                                                            \
        __syncthreads();                                                                                       \
        if (last_tap) { tap = 0; tap_off = 0; ++cs; } else { ++tap; tap_off += tap_step; }                     \
    }
        int s = 0;
        for (; s + 1 < nsteps; s += 2) {
            KN_D_STEP(1)                                // even step s: stages step s + 1 (odd -> set 1)
            ++s;
            KN_D_STEP(0)                                // odd step: stages step s + 1 (even -> set 0)
            --s;
        }
        if (s < nsteps) { KN_D_STEP(1) }
#undef KN_D_STEP
#undef KN_D_LOAD_W
#undef KN_D_STAGE_W
#undef KN_D_LOAD_B
#undef KN_D_STAGE_B
    }

    // Phase 2 of the fused ResBlock pair (conv_pair_kernel): the same tap loop with the A operand ALREADY resident in LDS for
    // every channel slab — the split image of the pair's inner activation, [slab][IR rows][PITCH] from byte 0, written by the
    // first convolution's epilogue — so only the weights stream (staged at byte `boff`).  Stride-1, dilation-1 taps: the
    // fragments of tap t sit t rows further.  Same products in the same order as `mainloop` (slab-major, taps inside).
    template <class BLoad, class RB>
    __device__ __forceinline__ static void mainloop_resident(float* lds_generic, int ncs, int taps, int img_slab_bytes, int boff,
                                                             BLoad& bload, f32x16 (&acc)[TM][TN], RB rb_desc) {
        lds_c* lds = (lds_c*)lds_generic;
        const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
        const int wm = wave / WN, wn = wave % WN;
        const int li = lane & 31, lh = lane >> 5;
        const int a_frag = (wm * TM * 32 + li) * PITCH + lh * 16;
        const int b_frag = boff + (wn * TN * 32 + li) * PITCH + lh * 16;
        g2_u32x4 rb[B_P];
#define KN_LOAD_B(SLAB)                                                                                        \
    { bload.begin(SLAB); _Pragma("unroll") for (int j = 0; j < B_P; ++j) rb[j] = bload(SLAB, j, rb_desc); }
#define KN_STAGE_B()                                                                                           \
    _Pragma("unroll") for (int j = 0; j < B_P; ++j) {                                                         \
        const int q = tid + 256 * j;                                                                          \
        if (B_PIECES % 256 == 0 || q < B_PIECES) *(lds_u4*)(lds + boff + (q >> 3) * PITCH + (q & 7) * 16) = rb[j]; \
    }
        KN_LOAD_B(0)
        KN_STAGE_B()
        __syncthreads();
        for (int cs = 0; cs < ncs; ++cs) {
            const bool more_cs = cs + 1 < ncs;
            const int a_cs = a_frag + cs * img_slab_bytes;
            for (int tap = 0; tap < taps; ++tap) {
                const bool last_tap = tap + 1 == taps;
                const bool more = !last_tap || more_cs;
                if (more) {
                    const int slab = last_tap ? (cs + 1) : ((tap + 1) * ncs + cs);
                    KN_LOAD_B(slab)
                }
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    f16x8 fa[TM][2], fb[TN][2];
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int p = 0; p < 2; ++p)
                            fa[i][p] = __builtin_bit_cast(f16x8, *(const lds_u4*)(lds + a_cs + tap * PITCH + i * 32 * PITCH + p * 64 + ks * 32));
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int p = 0; p < 2; ++p)
                            fb[j][p] = __builtin_bit_cast(f16x8, *(const lds_u4*)(lds + b_frag + j * 32 * PITCH + p * 64 + ks * 32));
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j) {
                            f32x16 c = acc[i][j];
                            c = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[i][1], fb[j][0], c, 0, 0, 0);   // small terms first
                            c = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[i][0], fb[j][1], c, 0, 0, 0);
                            c = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[i][0], fb[j][0], c, 0, 0, 0);
                            acc[i][j] = c;
                        }
                }
                __syncthreads();                 // every wave is done with this weight slab
                if (more) { KN_STAGE_B() }
                __syncthreads();
            }
        }
#undef KN_LOAD_B
#undef KN_STAGE_B
    }

    __device__ __forceinline__ static int acc_row(int wave, int lane, int i, int r) {
        return (wave / WN) * TM * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    }
    __device__ __forceinline__ static int acc_col(int wave, int lane, int j) {
        return (wave % WN) * TN * 32 + j * 32 + (lane & 31);
    }
};
