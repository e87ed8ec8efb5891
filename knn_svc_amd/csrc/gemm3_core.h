// fp32 GEMM emulated on the bf16 matrix cores ("bf16x3"): every fp32 operand x is split by
// truncation into three bf16 pieces x = hi + mid + lo (+ < 2^-24 |x|), and a product a*b is
// accumulated in fp32 from six bf16 MFMAs  a0b0 + a0b1 + a1b0 + a0b2 + a2b0 + a1b1  (the three
// dropped terms are below 2^-23 |a||b|).  Products of bf16 values are exact in fp32, so the result
// is as accurate as an fp32 FMA chain (measured: 1.6e-7 vs 3.4e-7 rms relative error against fp64
// at K = 1024), while v_mfma_f32_32x32x16_bf16 delivers 16x the MACs per cycle of
// v_mfma_f32_32x32x2_f32: 6/16 of the fp32-MFMA time per product, i.e. the fp32 matrix "peak"
// (157 TFLOP/s) is not the ceiling of this kernel, 2.67x that is.
//
// Block = 256 threads, 4 waves (WM x WN), wave tile (32 TM) x (32 TN), K slab = 32.
// A arrives as fp32 from HBM and is split while it is staged (VALU, hidden behind the other
// waves' MFMAs); B (weights) is split once at load time into [n][K/32][3 planes][32] bf16.
// LDS image per operand row: 3 planes x 64 B + 16 B pad = 208 B (52 dwords: the 16 rows of every
// ds_read_b128 lane group land on 16 distinct 4-bank slots).  One LDS buffer (53 KB at 128x128)
// + register prefetch, two barriers per slab, 2-3 blocks per CU.
#pragma once
#include "common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));

template <int BM_, int BN_, int WM_, int WN_, int TM_, int TN_>
struct Gemm3Tile {
    typedef f32x16 acc_t;
    static constexpr int NR = 16;
    static constexpr int BM = BM_, BN = BN_, WM = WM_, WN = WN_, TM = TM_, TN = TN_;
    static constexpr int BK = 32, PITCH = 208, THREADS = 256;
    static_assert(WM * WN == 4 && WM * TM * 32 == BM && WN * TN * 32 == BN, "tile shape");
    static constexpr int A_F4 = BM / 32;                       // fp32 float4 per thread per slab
    static constexpr int B_PIECES = BN * 12;                   // 16-byte pieces of split weights per slab
    static constexpr int B_P = (B_PIECES + 255) / 256;         // per thread
    static constexpr int LDS_BYTES = (BM + BN) * PITCH;

    typedef __attribute__((address_space(3))) char lds_c;

    // split 4 fp32 into three planes of 4 bf16 (truncation) and store them at row/plane/k
    __device__ __forceinline__ static void split_store(lds_c* dst, f32x4 v) {
        unsigned h[4], m[4], l[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const unsigned x = __float_as_uint(v[e]);
            const float r1 = v[e] - __uint_as_float(x & 0xFFFF0000u);
            const unsigned y = __float_as_uint(r1);
            const float r2 = r1 - __uint_as_float(y & 0xFFFF0000u);
            h[e] = x; m[e] = y; l[e] = __float_as_uint(r2);
        }
        // pack the upper halves of two dwords: bytes {b.3, b.2, a.3, a.2}
        u32x2_t p0 = {__builtin_amdgcn_perm(h[1], h[0], 0x07060302u), __builtin_amdgcn_perm(h[3], h[2], 0x07060302u)};
        u32x2_t p1 = {__builtin_amdgcn_perm(m[1], m[0], 0x07060302u), __builtin_amdgcn_perm(m[3], m[2], 0x07060302u)};
        u32x2_t p2 = {__builtin_amdgcn_perm(l[1], l[0], 0x07060302u), __builtin_amdgcn_perm(l[3], l[2], 0x07060302u)};
        typedef __attribute__((address_space(3))) u32x2_t lds_u2;
        *(lds_u2*)(dst) = p0;
        *(lds_u2*)(dst + 64) = p1;
        *(lds_u2*)(dst + 128) = p2;
    }

    // aload: fp32 A loader (begin(kt), operator()(kt, j) -> f32x4 for row (tid>>3)+32j, k = (tid&7)*4, finish())
    // bload: split-weight loader (begin(kt), operator()(kt, j) -> 16-byte piece j of this thread)
    template <class ALoad, class BLoad, class RA, class RB>
    __device__ __forceinline__ static void mainloop(float* lds_generic, int nk, ALoad& aload, BLoad& bload,
                                                    f32x16 (&acc)[TM][TN], RA ra_desc, RB rb_desc) {
        lds_c* lds = (lds_c*)lds_generic;
        typedef __attribute__((address_space(3))) u32x4_t lds_u4;
        const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
        const int wm = wave / WN, wn = wave % WN;
        const int a_st = (tid >> 3) * PITCH + (tid & 7) * 8;               // staging address of float4 j = 0
        const int li = lane & 31, lh = lane >> 5;
        const int a_frag = (wm * TM * 32 + li) * PITCH + lh * 16;
        const int b_frag = BM * PITCH + (wn * TN * 32 + li) * PITCH + lh * 16;

        f32x4 ra[A_F4];
        u32x4_t rb[B_P];
#define KN_STAGE3()                                                                                           \
    {                                                                                                        \
        _Pragma("unroll") for (int j = 0; j < A_F4; ++j)                                                     \
            split_store(lds + a_st + 32 * j * PITCH, aload.finish(ra[j]));                                   \
        _Pragma("unroll") for (int j = 0; j < B_P; ++j) {                                                    \
            const int q = tid + 256 * j;                                                                     \
            if (B_PIECES % 256 == 0 || q < B_PIECES)                                                         \
                *(lds_u4*)(lds + BM * PITCH + (q / 12) * PITCH + (q % 12) * 16) = rb[j];                      \
        }                                                                                                    \
    }
        aload.begin(0); bload.begin(0);
#pragma unroll
        for (int j = 0; j < A_F4; ++j) ra[j] = aload(0, j, ra_desc);
#pragma unroll
        for (int j = 0; j < B_P; ++j) rb[j] = bload(0, j, rb_desc);
        KN_STAGE3();
        __syncthreads();

        for (int kt = 0; kt < nk; ++kt) {
            const bool more = (kt + 1 < nk);
            if (more) {
                aload.begin(kt + 1); bload.begin(kt + 1);
#pragma unroll
                for (int j = 0; j < A_F4; ++j) ra[j] = aload(kt + 1, j, ra_desc);
#pragma unroll
                for (int j = 0; j < B_P; ++j) rb[j] = bload(kt + 1, j, rb_desc);
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8 fa[TM][3], fb[TN][3];
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int p = 0; p < 3; ++p)
                        fa[i][p] = __builtin_bit_cast(bf16x8, *(const lds_u4*)(lds + a_frag + i * 32 * PITCH + p * 64 + ks * 32));
#pragma unroll
                for (int i = 0; i < TN; ++i)
#pragma unroll
                    for (int p = 0; p < 3; ++p)
                        fb[i][p] = __builtin_bit_cast(bf16x8, *(const lds_u4*)(lds + b_frag + i * 32 * PITCH + p * 64 + ks * 32));
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        f32x16 c = acc[i][j];
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][1], fb[j][1], c, 0, 0, 0);   // smallest terms first
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][2], fb[j][0], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[j][2], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][1], fb[j][0], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[j][1], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[j][0], c, 0, 0, 0);
                        acc[i][j] = c;
                    }
            }
            __syncthreads();                 // every wave is done reading this slab
            if (more) KN_STAGE3();
            __syncthreads();
        }
    }

#undef KN_STAGE3
    __device__ __forceinline__ static int acc_row(int wave, int lane, int i, int r) {
        return (wave / WN) * TM * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    }
    __device__ __forceinline__ static int acc_col(int wave, int lane, int j) {
        return (wave % WN) * TN * 32 + j * 32 + (lane & 31);
    }
};
