// fp32 MFMA tile GEMM main loop shared by the implicit-GEMM convolution and the kNN
// distance kernel.  gfx950: v_mfma_f32_32x32x2_f32 (exact f32, 64 FLOP/clk/SIMD).
//
// Block = 256 threads = 4 waves arranged WM x WN; each wave owns TM x TN MFMA tiles of
// 32x32.  K is consumed in slabs of BK = 32 floats.  Both operands are "K-contiguous":
// the caller supplies loader functors that return the float4 at (row, k..k+3) of the
// current slab, so im2col addressing, masking and prologue activations live in the loader.
//
// Staging is global -> registers -> LDS (ds_write_b128) with two LDS buffers; the next
// slab's global loads are in flight while the current slab is multiplied.
//
// LDS image: [rows][36] floats (32 + 4 pad).  A lane (i = lane & 31, h = lane >> 5) reads
// the 16 bytes at [row i][8*g + 4*h] for k-group g: with the 144-byte row pitch the 16
// rows of every ds_read_b128 lane group fall on 16 distinct 4-bank slots (conflict free).
// The MFMA k order inside a slab is therefore permuted (lane half h supplies k = 8g+4h+j
// at step j), identically for A and B, which leaves the product unchanged.
#pragma once
#include "common.h"

template <int BM_, int BN_, int WM_, int WN_, int TM_, int TN_, int NT_ = 256>
struct GemmTile {
    typedef f32x16 acc_t;
    static constexpr int NR = 16;
    static constexpr int BM = BM_, BN = BN_, WM = WM_, WN = WN_, TM = TM_, TN = TN_;
    static constexpr int BK = 32, LDK = 36, THREADS = NT_;
    static constexpr int RP = THREADS / 8;        // rows staged per pass (8 float4 per 32-float row)
    static_assert(WM * WN * 64 == THREADS, "one wave per (wm, wn)");
    static_assert(WM * TM * 32 == BM && WN * TN * 32 == BN, "tile shape");
    static constexpr int A_F4 = BM / RP;      // float4 per thread per slab
    static constexpr int B_F4 = BN / RP;
    static constexpr int LDS_FLOATS = 2 * (BM + BN) * LDK;
    static constexpr int LDS_BYTES = LDS_FLOATS * 4;

    // aload.begin(kt) once per slab (uniform bookkeeping), then
    // aload(kt, j) -> f32x4 for row (tid>>3) + 32*j, k = kt*32 + (tid&7)*4 ; same for bload.
    // LDS pointers are kept in address space 3 explicitly: through generic pointers hipcc emits
    // flat_load/flat_store for the staging buffers and then waits vmcnt(0) in front of every MFMA
    // group, which also drains the NEXT slab's global prefetch (measured: 82 -> see DESIGN.md).
    typedef __attribute__((address_space(3))) float lds_f;
    typedef __attribute__((address_space(3))) f32x4 lds_f4;

    // aload.begin(kt) once per slab (uniform bookkeeping), then
    // aload(kt, j) -> f32x4 for row (tid>>3) + 32*j, k = kt*32 + (tid&7)*4 ; same for bload.
    // aload.finish(v) is applied when the registers are written to LDS (prologue activation), so
    // that nothing has to wait on the global load right after issuing it.
    // ra_desc / rb_desc: opaque per-operand values handed to every aload()/bload() call BY VALUE (buffer
    // resource descriptors must not live inside the loader structs: SROA cannot split a struct holding an
    // address-space-8 pointer, the struct lands in scratch and every buffer_load becomes a waterfall loop).
    template <class ALoad, class BLoad, class RA, class RB>
    __device__ __forceinline__ static void mainloop(float* lds_generic, int nk, ALoad& aload, BLoad& bload,
                                                    f32x16 (&acc)[TM][TN], RA ra_desc, RB rb_desc) {
        lds_f* lds = (lds_f*)lds_generic;
        const int tid = threadIdx.x;
        const int lane = tid & 63, wave = tid >> 6;
        const int wm = wave / WN, wn = wave % WN;
        const int srow = tid >> 3, skq = (tid & 7) * 4;
        constexpr int BUF = (BM + BN) * LDK;          // floats per staging buffer: [A rows | B rows]

        f32x4 ra[A_F4], rb[B_F4];
        aload.begin(0); bload.begin(0);
#pragma unroll
        for (int j = 0; j < A_F4; ++j) ra[j] = aload(0, j, ra_desc);
#pragma unroll
        for (int j = 0; j < B_F4; ++j) rb[j] = bload(0, j, rb_desc);
#pragma unroll
        for (int j = 0; j < A_F4; ++j) *(lds_f4*)&lds[(srow + RP * j) * LDK + skq] = aload.finish(ra[j]);
#pragma unroll
        for (int j = 0; j < B_F4; ++j) *(lds_f4*)&lds[BM * LDK + (srow + RP * j) * LDK + skq] = rb[j];
        __syncthreads();

        const int li = lane & 31, lh = lane >> 5;
        const int a_frag = (wm * TM * 32 + li) * LDK + lh * 4;
        const int b_frag = BM * LDK + (wn * TN * 32 + li) * LDK + lh * 4;
        for (int kt = 0; kt < nk; ++kt) {
            const int cur = (kt & 1) * BUF;
            const bool more = (kt + 1 < nk);
            if (more) {
                aload.begin(kt + 1); bload.begin(kt + 1);
#pragma unroll
                for (int j = 0; j < A_F4; ++j) ra[j] = aload(kt + 1, j, ra_desc);
#pragma unroll
                for (int j = 0; j < B_F4; ++j) rb[j] = bload(kt + 1, j, rb_desc);
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 fa[TM], fb[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) fa[i] = *(const lds_f4*)&lds[cur + a_frag + i * 32 * LDK + g * 8];
#pragma unroll
                for (int i = 0; i < TN; ++i) fb[i] = *(const lds_f4*)&lds[cur + b_frag + i * 32 * LDK + g * 8];
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][s], fb[j][s], acc[i][j], 0, 0, 0);
            }
            if (more) {
                const int nxt = BUF - cur;
#pragma unroll
                for (int j = 0; j < A_F4; ++j) *(lds_f4*)&lds[nxt + (srow + RP * j) * LDK + skq] = aload.finish(ra[j]);
#pragma unroll
                for (int j = 0; j < B_F4; ++j) *(lds_f4*)&lds[nxt + BM * LDK + (srow + RP * j) * LDK + skq] = rb[j];
            }
            __syncthreads();
        }
    }

    // accumulator element (tile i,j; register r) of this lane -> (row, col) inside the block tile
    __device__ __forceinline__ static int acc_row(int wave, int lane, int i, int r) {
        return (wave / WN) * TM * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    }
    __device__ __forceinline__ static int acc_col(int wave, int lane, int j) {
        return (wave % WN) * TN * 32 + j * 32 + (lane & 31);
    }
};
