/*
 * libknnsvc_hip.so — C ABI of the MI355X (gfx950) kNN-SVC inference kernels.
 *
 * The reference (SmoothKen/knn-svc) has no FFI: its seams are Python callables
 * executed by PyTorch ATen.  Each entry point below replaces one of those
 * callables; the comment above it cites the reference file:line.  A maintainer
 * binds them with ctypes (see INTEGRATION.md for the stubs).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (HBM) unless named host_*;
 *   - activations are channel-last: a [T, C] matrix with row stride `ld*` (floats);
 *   - all work is enqueued on `stream` (a hipStream_t passed as void*; NULL = default
 *     stream); nothing synchronises, nothing allocates — callers own every buffer and
 *     workspace, so each call is hipGraph-capturable;
 *   - return value: 0 on success, a KNNSVC_E* code otherwise; knnsvc_last_error()
 *     gives the thread-local message.
 */
#ifndef KNNSVC_HIP_H
#define KNNSVC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KNNSVC_OK        0
#define KNNSVC_EINVAL    1   /* bad shape / alignment / argument          */
#define KNNSVC_EWORKSPACE 2  /* workspace too small                        */
#define KNNSVC_EHIP      3   /* a HIP runtime call failed                  */
#define KNNSVC_ENAN      4   /* NaN distance (the reference sys.exit()s)   */

#define KNNSVC_ABI_VERSION 17

int knnsvc_abi_version(void);
const char* knnsvc_last_error(void);
/* Short tag of the kernel the calling thread's last knnsvc_conv_gemm dispatched to ("F128a2", "W64", "G128v8", ...):
 * measurement hook (bench.py attributes HIP-event times to kernels with it); "" before the first call. */
const char* knnsvc_conv_gemm_last_kernel(void);
/* ... and which epilogue that launch took: "patch" (LDS patches, 16-byte stores: conv_epilogue_wide32), "lane" (column per lane) or
 * "" (kernels with an epilogue of their own: the 256x256-tile kernel, fp32, bf16x3).  Test hook. */
const char* knnsvc_conv_gemm_last_epilogue(void);
/* The dispatcher's A/B switches (KNNSVC_QUAD, KNNSVC_QUAD_EPI, KNNSVC_EPILOGUE, KNNSVC_WIN, KNNSVC_WIN_SMALL, KNNSVC_WIN_DEEP,
 * KNNSVC_WIN160, KNNSVC_WIN_WIDE, KNNSVC_GEMM_SMALL) are read from the environment once, at the first launch; this re-reads them (tests and A/B
 * runs that switch a route inside one process). */
int knnsvc_reload_knobs(void);
/* Stream-placement probe: `blocks` one-wave workgroups that spin for `spin` shader-clock ticks each (nothing is read or written).
 * The host side (knn_svc_amd/pipeline.py) times it on two streams at once to MEASURE whether they share a hardware queue or
 * dispatch pipe, instead of trusting the stream -> queue mapping. */
int knnsvc_probe_dispatch(int32_t blocks, int32_t spin, void* stream);

/* ------------------------------------------------------------------------------------------
 * Implicit-GEMM convolution / linear layer on fp32 MFMA (v_mfma_f32_32x32x2_f32).
 * Replaces F.conv1d / F.conv_transpose1d / F.linear as used by
 *   wavlm/WavLM.py:401,514-527 (feature extractor, pos_conv), wavlm/WavLM.py:347-348, 671-672,
 *   wavlm/modules.py:540-563 (q/k/v/out projections),
 *   hifigan/ddsp_models.py:113-168,176-233 (every conv of the Generator), :416 (sin_prenet).
 *
 *   out[b,g][m, n] = epilogue( sum_{tap,c} A(m,tap,c) * W[g][n][tap*cin + c] )
 *   A(m,tap,c)     = lrelu_{a_slope}( X[b,g][m*stride + tap*dil - pad][c] )   (0 outside [0,t_in))
 *   epilogue(v)    = ((act(v + bias[n % bias_period]) + resid[m,n]) + (accumulate ? out[m,n] : 0)) / div
 * With convt_u > 0 the output element (m, n) is scattered to row m*convt_u + n/convt_cout - convt_pad,
 * column n % convt_cout (rows outside [0,t_out) dropped): a stride-u transposed convolution whose
 * kernel is taps*u wide, written as one GEMM over K = taps*cin, N = u*cout.
 * X rows are ldx floats apart; ldx >= cin, except that taps == 1 also accepts overlapping rows (ldx < cin):
 * the framed view of a signal, row t = x[t*ldx .. t*ldx + cin) — how the 400-point STFT (hop 320) is a GEMM.
 * ------------------------------------------------------------------------------------------ */
enum { KNNSVC_ACT_NONE = 0, KNNSVC_ACT_GELU = 1, KNNSVC_ACT_LRELU = 2, KNNSVC_ACT_TANH = 3 };

typedef struct knnsvc_conv_desc {
    const float* x;  int64_t x_bstride; int64_t x_gstride; int32_t ldx; int32_t t_in;
    int32_t cin; int32_t taps; int32_t stride; int32_t dil; int32_t pad;
    float a_slope;                     /* 1.0f = no prologue activation                   */
    const float* w;  int64_t w_gstride; int32_t n;
    const float* bias; int64_t bias_gstride; int32_t bias_period;   /* bias may be NULL  */
    float* out; int64_t o_bstride; int64_t o_gstride; int32_t ldo; int32_t m;
    int32_t act; float act_slope;
    const float* resid; int64_t r_bstride; int64_t r_gstride; int32_t ldr;   /* may be NULL */
    int32_t accumulate; float div;     /* div == 1.0f = off                               */
    int32_t batches; int32_t groups;
    int32_t convt_u; int32_t convt_cout; int32_t convt_pad; int32_t t_out;
    const void* w_bf16x3;              /* optional: w pre-split by knnsvc_split_weight_bf16x3 (NULL = fp32 MFMA) */
    const void* w_f16x2;               /* optional: w pre-split by knnsvc_split_weight_f16x2; wins over w_bf16x3 */
    float w_f16x2_scale;               /* the power-of-two scale w_f16x2 was split with                         */
    float a_f16x2_scale;               /* power-of-two activation pre-scale of the f16x2 path; 0 = default 16    */
    int32_t x_f16x2;                   /* 1: x already is in the f16x2 split layout (see below), split with            */
                                       /*    a_f16x2_scale (0 = 16) or with the scale x_absmax implies               */
    int32_t out_f16x2;                 /* 1: write out in the f16x2 split layout (scale 16) for the next GEMM;      */
                                       /* c >= 32 (multiple of 32): only columns >= c are split (QKV: K,V blocks)   */
    /* Range of the f16x2 path without host round trips (all optional, NULL / 0 = off; ignored by the fp32 / bf16x3 paths):
     * A RANGE SLOT is 2048 DEVICE floats (8 KiB) used as 64 stripes of one 128-byte cache line each (float 32 i is stripe i):
     * a producer block folds max|.| into stripe (block id % 64) with one atomicMax — atomics that meet on a cache line
     * serialise, and a launch's first round of blocks all find the slot empty — and consumers use the maximum of the 64.
     * Its content is a pure function of the data (max is order-independent).
     *   x_absmax / w_absmax: range slots holding an upper bound of |x| over the A operand / of |w| over the split
     *     weights; the kernel then derives the operand's power-of-two scale itself — the largest s with bound * s < 2^15
     *     (knnsvc_split_f16x2_dyn splits with the same rule, so a pre-split operand and its consumer agree by sharing a
     *     slot) — instead of a_f16x2_scale / w_f16x2_scale.  No activation range can overflow fp16 this way, and tiny
     *     tensors are lifted into the range instead of losing bits.
     *   out_absmax: range slot; the epilogue folds max|out| over everything this launch stores into it with atomicMax
     *     (on the bit pattern: a NaN output makes the slot NaN).  The caller zeroes the slot; it is the next launch's
     *     x_absmax.
     *   out_f16x2_scale: scale of the split layout written under out_f16x2 (0 = 16); the consumer passes the same value
     *     as its a_f16x2_scale.
     *   x_bound_mul / x_bound_add (0 / 0 = 1 / 0): the A operand's bound is x_bound_mul * max(x_absmax) + x_bound_add — for an
     *     input that was not measured itself but whose bound follows from ITS producer's measured input and weights
     *     (|conv(x)| <= max_n sum_k |w[n,k]| * max|x| + max|b|; activations that do not grow their argument change nothing):
     *     publishing costs short-lived blocks ~1 us each, so only every other tensor of a conv chain is measured. */
    const float* x_absmax; const float* w_absmax; float* out_absmax; float out_f16x2_scale;
    /* Dynamic length (optional; NULL = off): n_dyn is a DEVICE int32 holding a count n (frames); the kernel then uses
     *   t_in = n * dyn_t_in_mul + dyn_t_in_add,  m = n * dyn_m_mul + dyn_m_add,  t_out = n * dyn_t_out_mul (transposed mode)
     * instead of the fields above, which become the MAXIMA the launch is sized for (grid, 31-bit offsets).  Input rows
     * >= t_in read as zero (the convolution's own zero padding at the sequence end), output rows >= m are not written.
     * One launch — one captured hipGraph — thus serves every sequence length up to its bucket with the results of an
     * exact-length launch: the generator's frame-count buckets (hifigan/ddsp_models.py:176-233 is fully convolutional). */
    const int32_t* n_dyn; int32_t dyn_t_in_mul; int32_t dyn_t_in_add; int32_t dyn_m_mul; int32_t dyn_m_add; int32_t dyn_t_out_mul;
    float x_bound_mul; float x_bound_add;
    /* 2: always the 256x256-tile kernel (Gemm2QuadS; needs x_f16x2, n % 4 == 0, 16-byte rows) — what the kNN's dot-matrix
     * route sets, so that it sums over K exactly as the fused screen kernel does.
     * 1: always the 128x128-tile kernel, whatever the launch size.  The default picks the 256x256-tile kernel for large
     * launches; the two sum over K in different groupings, so a row's result would depend (in the last bit) on how many other
     * rows and columns the launch has.  The distance GEMM of the kNN sets this: a pool shard of any size gives the distances the
     * whole pool gives (device-count invariance of the sharded search, lib_ongaku_test.py:148-175 is one formula). */
    int32_t fixed_tile;
} knnsvc_conv_desc;

int knnsvc_conv_gemm(const knnsvc_conv_desc* d, void* stream);

/* `count` (1..4) convolutions of the same output shape [m, n] in ONE launch (hifigan/ddsp_models.py:206-227: the three
 * ResBlock branches of a generator stage — kernel sizes 3 / 7 / 11 — only share their input; the reference runs them one
 * after the other and averages).  When every descriptor takes the windowed kernel (stride 1, >= 3 taps, fp32 input, split
 * weights, (taps - 1) * dil <= 64, batches = groups = 1) the descriptors become the y dimension of one grid: the branches
 * fill the chip together without streams of their own.  Otherwise: one knnsvc_conv_gemm per descriptor, in order.  Results
 * are bit-identical to separate launches either way.  Put the descriptor with the most taps first. */
int knnsvc_conv_gemm_multi(const knnsvc_conv_desc* descs, int32_t count, void* stream);

/* One ResBlock1 iteration of the generator in one launch (hifigan/ddsp_models.py:13-44):
 *     t1 = lrelu(conv1d(lrelu(x), w1, dilation = dil) + b1);   out = conv1d(t1, w2) + b2 + x
 * x, out: channel-last [t, channels] fp32 (row pitches ldx / ldo), both convolutions `taps` wide with "same" zero padding; the
 * inner activation t1 lives only in LDS.  Weights as for knnsvc_conv_gemm: packed [channels, taps * channels] and pre-split
 * (knnsvc_split_weight_f16x2) with their power-of-two scales.  Activation scales as in "Range": x_absmax is x's range slot
 * (t1 is bounded by t1_bound_mul * max|x| + t1_bound_add, as knnsvc_conv_desc.x_bound_*), or fixed a1_scale / a2_scale.
 * out_absmax (optional): max |out| is folded into that slot.  n_dyn (optional): the valid length is n_dyn[0] * dyn_mul <= t.
 * Results are bit-identical to the two knnsvc_conv_gemm launches.  channels = 32 or 64. */
typedef struct knnsvc_pair_desc {
    const float* x; int32_t ldx; int32_t t; int32_t channels; int32_t taps; int32_t dil;
    const void* w1_f16x2; float w1_scale; const float* b1;
    const void* w2_f16x2; float w2_scale; const float* b2;
    float* out; int32_t ldo;
    float slope;
    const float* x_absmax; float t1_bound_mul; float t1_bound_add; float a1_scale; float a2_scale;
    float* out_absmax;
    const int32_t* n_dyn; int32_t dyn_mul;
} knnsvc_pair_desc;
int knnsvc_resblock_pair(const knnsvc_pair_desc* d, void* stream);
/* The pairs of `count` (1..4) ResBlock branches (same channels and length, any odd taps) in one launch: see knnsvc_conv_gemm_multi. */
int knnsvc_resblock_pair_multi(const knnsvc_pair_desc* descs, int32_t count, void* stream);

/* out = alpha * x (accumulate = 0) or out + alpha * x (accumulate = 1) over n floats (n % 4 == 0): one term of the layer
 * weighting `(feats * w[:, None]).sum(0)` over WavLM's layer outputs (ddsp_prematch_dataset.py:349-350) when w is not one-hot
 * (terms are added in ascending layer order, each product rounded on its own, as torch evaluates it). */
int knnsvc_axpy(const float* x, int64_t n, float alpha, int32_t accumulate, float* out, void* stream);

/* out = (c + (b + a)) / div, elementwise over n floats (n % 4 == 0), max |out| folded into the range slot out_absmax (may be
 * NULL): the mean of the parallel ResBlock branches of a generator stage (hifigan/ddsp_models.py:218-227).  n_dyn (optional,
 * device int32): only the first n_dyn * dyn_mul floats are touched (bucketed sequence lengths, see "Dynamic length"). */
int knnsvc_mean3(const float* a, const float* b, const float* c, int64_t n, float div, float* out, float* out_absmax,
                 const int32_t* n_dyn, int64_t dyn_mul, void* stream);

/* Split fp32 weights [rows, K] (K % 32 == 0) into three truncated bf16 planes, layout [rows][K/32][3][32]
 * (6 bytes per weight).  With w_bf16x3 set and cin % 32 == 0, knnsvc_conv_gemm evaluates every fp32
 * product as six bf16 MFMAs (a0b0+a0b1+a1b0+a0b2+a2b0+a1b1, fp32 accumulate): fp32-level accuracy at
 * 6/16 of the fp32-MFMA cost.  Activations are split on the fly inside the kernel. */
int knnsvc_split_weight_bf16x3(const float* w, int64_t rows, int32_t K, void* out, void* stream);

/* Split scale*w (scale = a power of two that puts max|w| in [2^13, 2^14)) into two round-to-nearest fp16
 * planes, layout [rows][K/32][2][32] (4 bytes per weight).  With w_f16x2 set and cin % 32 == 0,
 * knnsvc_conv_gemm evaluates every fp32 product as three fp16 MFMAs (lo*hi + hi*lo + hi*hi, fp32
 * accumulate; activations are scaled by 16 and split on the fly, the epilogue undoes both scales exactly):
 * fp32-GEMM accuracy at 3/16 of the fp32-MFMA cost.  a_f16x2_scale * |activation| must stay below 65504
 * (default scale 16: |x| < 4094) or the output turns NaN (never silently wrong); activations whose rms is
 * below ~0.2 / a_f16x2_scale lose relative accuracy (absolute floor 3e-8 / a_f16x2_scale per element). */
int knnsvc_split_weight_f16x2(const float* w, int64_t rows, int32_t K, float scale, void* out, void* stream);

/* The same split with the scale taken from a range slot (8 KiB, see knnsvc_conv_desc): scale = the largest power of two with max(slot) * scale < 2^15
 * (what knnsvc_conv_gemm derives from x_absmax / w_absmax).  For operands whose range is only known on the device — the
 * kNN's query and pool features (lib_ongaku_test.py:148-175 takes whatever WavLM produced). */
int knnsvc_split_f16x2_dyn(const float* w, int64_t rows, int32_t K, const float* absmax, void* out, void* stream);

/* Folds max |x[r, c]| over a [rows, cols] matrix with row pitch ld into the range slot (8 KiB; atomicMax on the bit
 * pattern, NaN wins); the caller zeroes the slot.  Feeds x_absmax / w_absmax for tensors that no GEMM epilogue produced. */
int knnsvc_absmax(const float* x, int64_t rows, int32_t cols, int32_t ld, float* slot, void* stream);

/* Activations in the f16x2 split layout ("A2"): a [rows, C] matrix (C % 32 == 0, row pitch ld floats, ld % 32 == 0)
 * occupies the same bytes as fp32, but every group of 32 channels is stored as 32 fp16 `hi` values followed by
 * 32 fp16 `lo` values with 16*x = hi + lo (round to nearest twice): element (r, c) -> byte r*ld*4 + (c/32)*128 +
 * plane*64 + (c%32)*2.  Producers that feed ONLY GEMMs write it (out_f16x2 here, the split flags of
 * knnsvc_layernorm / knnsvc_wavlm_conv0 / knnsvc_wavlm_attention), consumers set x_f16x2: the GEMM then stages A with
 * plain copies instead of splitting it once per column tile. */

/* ------------------------------------------------------------------------------------------
 * Row-wise layer norm over the last dim (eps 1e-5, affine), optional exact-erf GELU after it.
 * Replaces F.layer_norm at wavlm/WavLM.py:342, 415 (+nn.GELU :418), 692, 706.  In place allowed.
 * ------------------------------------------------------------------------------------------ */
/* flags: bit 0 = GELU after the norm, bit 1 = write the f16x2 split layout (dim % 32 == 0, ldo % 32 == 0). */
int knnsvc_layernorm(const float* x, int64_t rows, int32_t dim, int32_t ldx, const float* gamma,
                     const float* beta, int32_t flags, float* out, int32_t ldo, void* stream);

/* Layer 0 of the conv feature extractor fused: Conv1d(1->C, k, stride, no bias) -> LayerNorm(C) -> GELU
 * (wavlm/WavLM.py:401-419 with in_d = 1).  x [batches, L] -> out [batches * T, C], T = (L-k)/stride + 1,
 * w [C, k].  C in {64,128,256,512}, k <= 16, stride <= 8. */
int knnsvc_wavlm_conv0(const float* x, int32_t batches, int64_t L, const float* w, int32_t channels, int32_t k,
                       int32_t stride, const float* gamma, const float* beta, float* out, int32_t out_f16x2,
                       void* stream);

/* Gated relative-position multiplier, wavlm/modules.py:523-533:
 *   gate[row, h] = ga*(gb*grep_a[h] - 1) + 2, (ga, gb) = sigmoid(W2 @ xn[row, h*hd:(h+1)*hd] + b2)
 * where W2 [2, hd] / b2 [2] are grep_linear's weight rows / bias summed in groups of four. */
int knnsvc_wavlm_gate(const float* xn, int64_t rows, int32_t heads, int32_t head_dim, int32_t ldx,
                      const float* w2, const float* b2, const float* grep_a, float* gate, int32_t x_f16x2,
                      void* stream);

/* Fused bidirectional self-attention with the gated bucketed relative-position bias
 * (wavlm/modules.py:504-506, 533-563 -> F.multi_head_attention_forward, need_weights=False):
 *   O[b, i, h, :] = softmax_j( q_i.k_j * head_dim^-0.5 + gate[b,i,h] * table[h][j - i + T - 1] ) @ V
 * qkv is [batches*T, 3*E] (q | k | v column blocks, E = heads*64), table is [heads][2T-1]
 * (bucket LUT already applied), gate [batches*T, heads], out [batches*T, E].  head_dim must be 64.
 * Nothing of size T x T is ever written to HBM. */
/* kv_f16x2: the K and V column blocks of `qkv` (columns E..3E) already hold the f16x2 split layout, written by the QKV
 * projection with knnsvc_conv_desc.out_f16x2 = E (split from column E on); Q stays fp32.  f16x2 kernel only.
 * out_f16x2 is a flag word: bit 0 = write the output in the split layout; bit 2 (value 4) = wide range: Q, K or V may
 * exceed what the f16x2 kernel's fixed operand scales hold (|k|, |v| < 4094, |q| < ~20000 — the caller decides this
 * once, from bounds implied by the weights), so the bf16x3 kernel (fp32 exponent range) runs instead; fp32 in and out. */
/* kv_len (may be NULL): DEVICE int32 [batches] — batch row b attends to its first kv_len[b] keys only (clamped to 1..T):
 * WavLM's key_padding_mask for a chunk that sits zero-padded inside a longer, bucketed sequence (wavlm/WavLM.py:311-321;
 * F.multi_head_attention_forward masks padded keys with -inf).  The table stays indexed with the bucket's T.  Read on the
 * device at launch time, so one captured hipGraph serves every length of its bucket.  Query rows >= kv_len[b] still get
 * (finite, unused) outputs. */
int knnsvc_wavlm_attention(const float* qkv, const float* gate, const float* table, int32_t batches,
                           int32_t T, int32_t heads, float* out, int32_t out_f16x2, int32_t kv_f16x2, const int32_t* kv_len,
                           void* stream);

/* x[b, t, :] = 0 for t >= lens[b] on a [batches, T, dim] activation (row pitch ld): WavLM's `x[padding_mask] = 0`
 * (wavlm/WavLM.py:353, 574-575) in front of the positional convolution, for chunks padded up to a bucket length. */
int knnsvc_mask_rows(float* x, int32_t batches, int32_t T, int32_t dim, int32_t ld, const int32_t* lens, void* stream);

/* ------------------------------------------------------------------------------------------
 * Whole-model entry points (round 5): ONE call enqueues the layer sequence of a network on the caller's stream.
 * A handle holds a copy of the descriptor — packed weights, their f16x2 splits and scales, LayerNorm vectors, the
 * range plan decided at load — i.e. POINTERS to device memory the caller keeps alive until *_free; the library
 * allocates nothing on the device and nothing at all inside the hot call (hipGraph-capturable), activations live in
 * a caller-provided workspace.  The sequence is the one knn_svc_amd/wavlm.py issued launch by launch until round 4:
 * same kernels, same arguments, same bits.
 * ------------------------------------------------------------------------------------------ */
typedef struct knnsvc_weight {            /* a packed GEMM weight matrix [n, K] and its split (knnsvc_split_weight_f16x2) */
    const float* w; const void* w_f16x2; float w_f16x2_scale; int32_t pad_;
} knnsvc_weight;

typedef struct knnsvc_wavlm_conv {        /* one layer of the conv feature extractor (wavlm/WavLM.py:401-419, 485-504) */
    knnsvc_weight w;                      /* [dim, k * cin] (layer 0 with cin = 1: [dim, k], fp32 only)                  */
    const float* ln_g; const float* ln_b; /* LayerNorm(dim) behind it (extractor_mode "layer_norm")                        */
    int32_t dim, k, stride, cin;
    int32_t out_split;                    /* the activation behind this layer may travel in the f16x2 split layout (load-time range plan) */
    int32_t pad_;
} knnsvc_wavlm_conv;

typedef struct knnsvc_wavlm_layer {       /* TransformerSentenceEncoderLayer, layer_norm_first (wavlm/WavLM.py:691-714) */
    const float *ln1_g, *ln1_b, *ln2_g, *ln2_b;
    knnsvc_weight wqkv; const float* bqkv;          /* q | k | v fused: [3E, E]                                            */
    knnsvc_weight wo;   const float* bo;
    knnsvc_weight w1;   const float* b1;            /* fc1 [ffn, E] */
    knnsvc_weight w2;   const float* b2;            /* fc2 [E, ffn] */
    const float *gate_w, *gate_b, *grep_a;          /* grep_linear summed to [2, 64] / [2]; grep_a [H] (wavlm/modules.py:523-533) */
    int32_t xn_split, xn2_split, h_split, attn_f16; /* range plan: which activations fit the fixed-scale split layout      */
} knnsvc_wavlm_layer;

typedef struct knnsvc_wavlm_desc {
    int32_t n_conv; int32_t n_layers;
    const knnsvc_wavlm_conv* conv; const knnsvc_wavlm_layer* layers;      /* HOST arrays, copied by knnsvc_wavlm_create    */
    const float *ln_g, *ln_b; int32_t feats_split; int32_t pad_;          /* layer_norm over the extractor's output (WavLM.py:342) */
    knnsvc_weight proj; const float* proj_b;                               /* post_extract_proj (:347-348)                  */
    knnsvc_weight pos; const float* pos_b; int32_t pos_groups, pos_k;      /* pos_conv, weight norm folded, packed per group (:514-527) */
    float pos_a_scale;                    /* fixed activation scale of the positional conv from the load-time bound (0: a range slot) */
    int32_t E, H, ffn;
    const float* layer_mix;               /* HOST [n_layers + 1] weights of a general layer weighting, NULL = the output of layer n_layers */
} knnsvc_wavlm_desc;

int knnsvc_wavlm_create(const knnsvc_wavlm_desc* d, void** handle);
int knnsvc_wavlm_free(void* handle);
/* frames of one chunk of L samples; bytes of workspace knnsvc_wavlm_encode needs for [batches, L] */
int64_t knnsvc_wavlm_frames(const void* handle, int64_t L);
size_t knnsvc_wavlm_workspace_bytes(const void* handle, int32_t batches, int64_t L);
/* WavLM.extract_features up to layer n_layers (wavlm/WavLM.py:323-375, 572-612) for [batches, L] equal-length chunks:
 * wav -> out [batches * T, E].  lens (may be NULL): DEVICE int32 [batches], valid frames per chunk (WavLM's padding mask:
 * rows >= lens[b] zeroed in front of the positional conv, excluded as attention keys).  table: [H][2T - 1] relative
 * position bias with the bucket LUT applied (host-built: wavlm/modules.py:417-455).  Everything is enqueued on `stream`. */
int knnsvc_wavlm_encode(const void* handle, const float* wav, int32_t batches, int64_t L, const int32_t* lens,
                        const float* table, float* out, void* workspace, size_t workspace_bytes, void* stream);

/* The conditioned HiFi-GAN generator behind one call: SynthesizerTrn.forward + Generator.forward
 * (hifigan/ddsp_models.py:108-233, 405-493 'mix'; hifigan/ddsp_models_f0.py:106-216, 320-381 'f0').  Weights are
 * packed as knn_svc_amd/vocoder.py packs them (weight norm folded; conv [Cout, k*Cin]; transposed conv [u*Cout, (k/u)*Cin]),
 * each with its f16x2 split.  Three ResBlocks per stage (resblock_kernel_sizes has three entries), three dilations each. */
typedef struct knnsvc_gen_pair {          /* one iteration of ResBlock1.forward (hifigan/ddsp_models.py:13-44) */
    knnsvc_weight w1; const float* b1; knnsvc_weight w2; const float* b2;
    int32_t dil; float t1_bound_mul, t1_bound_add;   /* |convs1(lrelu(x)) + b1| <= mul * max|x| + add (from the weights, at load) */
    int32_t pad_;
} knnsvc_gen_pair;
typedef struct knnsvc_gen_stage {
    knnsvc_weight up; const float* up_b; int32_t u, k, cin, cout;      /* ups[i]: ConvTranspose1d(cin -> cout, k, stride u)            */
    knnsvc_weight ccv;                                                 /* concat_conv[i] (k = 3, no bias)                               */
    int32_t res_k[3]; int32_t pad_; knnsvc_gen_pair res[3][3];         /* resblocks[3 i + j]: kernel size res_k[j], pairs res[j][0..2]  */
    knnsvc_weight down; const float* down_b; int32_t down_k, down_u;   /* downs[i]: strided conv of the side path                       */
    knnsvc_weight rbd; const float* rbd_b;                             /* resblocks_downs[i].convs[0] (k = 3)                           */
} knnsvc_gen_stage;
typedef struct knnsvc_generator_desc {
    int32_t kind;                         /* 0 = 'mix' (additive-synth excitation, harmonic amplitudes), 1 = 'f0' (sine excitation) */
    int32_t n_up, hop, sample_rate, n_harm_in, uic, hubert_dim, hifi_dim;      /* lin_pre: hubert_dim -> hifi_dim; conv_pre: hifi_dim -> uic */
    int32_t side[8];                      /* channels of the side path's levels: side[0] = condition, side[i + 1] = output of down stage i */
    knnsvc_weight lin; const float* lin_b; knnsvc_weight pre; const float* pre_b;       /* lin_pre, conv_pre (k = 7)              */
    knnsvc_weight cpre; const float* cpre_b; knnsvc_weight post;                        /* concat_pre (k = 3), conv_post (k = 7)  */
    const float *prenet_w, *prenet_b;                                                   /* sin_prenet: Conv1d(1 -> side[0], 3)     */
    const knnsvc_gen_stage* stages;       /* HOST array [n_up], copied by knnsvc_generator_create */
} knnsvc_generator_desc;

int knnsvc_generator_create(const knnsvc_generator_desc* d, void** handle);
int knnsvc_generator_free(void* handle);
size_t knnsvc_generator_workspace_bytes(const void* handle, int64_t frames);
/* c [frames, hubert_dim], f0 [frames], harm [frames, n_harm_in] (NULL for kind 1) -> out [frames * hop].  n_dyn (may be NULL):
 * DEVICE int32 holding the valid frame count <= frames — the launch sequence is laid out for `frames` (a length bucket) and
 * computes exactly what an exact-length run computes (knnsvc_conv_desc, "Dynamic length").  Enqueues only; capturable. */
int knnsvc_generator_forward(const void* handle, const float* c, const float* f0, const float* harm, int64_t frames,
                             const int32_t* n_dyn, float* out, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * Cosine-distance kNN (lib_ongaku_test.py:148-175 fast_cosine_dist + Tensor.topk(k, largest=False),
 * driver loop ddsp_prematch_dataset.py:1195-1210).
 * ------------------------------------------------------------------------------------------ */
/* norm[r] = sqrt(sum x^2) (torch.norm, lib_ongaku_test.py:150-151); sq[r] = sum x^2 (cdist's own term).
 * max_slot (may be NULL): range slot (8 KiB), folded with the largest row norm (atomicMax on the bit pattern; the caller
 * zeroes it) — an upper bound of max|x| that costs no extra pass: the range slot the kNN's f16x2 GEMM scales by. */
int knnsvc_row_norms(const float* x, int64_t rows, int32_t dim, int32_t ldx, float* norm, float* sq, float* max_slot,
                     void* stream);

size_t knnsvc_knn_workspace_bytes(int64_t nq, int64_t np, int32_t k);

/* Wide lists and the exact re-score (round 5).  "Bit-exact top-k indices" is defined against the reference's fp32 formula
 * (lib_ongaku_test.py:162-165) evaluated on ITS matrix product; whatever product a GPU kernel forms (here: fp16-split MFMAs with 96
 * fp32 roundings along K, or fp32 MFMAs) differs from it in the last bits, and neighbours closer together than those bits swap.
 * The selection kernels below therefore only PICK candidates: each keeps, per query row, a WIDE list of up to 64 keys
 * (uint64: order-preserving bits of the screening distance << 32 | pool row) — the k best and every pair within 4e-6 of the
 * k-th — and knnsvc_knn_rescore recomputes q.p for exactly those pairs in fp64 from the fp32 operands, rounds once to fp32,
 * replays the reference's formula and sorts by (distance bits, lower index first).  What is left between this order and the
 * reference's is the reference's own BLAS rounding.  exact = 0 passes the screening order through (A/B aid).
 * wide: [nq][64] keys ascending, unused places 0xFFFF...F; out_idx / out_dist: [nq][k], indices + idx_offset; a row with fewer
 * than k finite distances (a NaN row) is filled with VALID row indices and NaN distances.  [mask_lo, mask_hi) as below. */
int knnsvc_knn_rescore(const void* wide, int64_t nq, int32_t k, const float* q, const float* q_norm, const float* q_sq,
                       const float* pool, const float* p_norm, const float* p_sq, int64_t np, int32_t dim,
                       int64_t idx_offset, int64_t mask_lo, int64_t mask_hi, int32_t exact,
                       int64_t* out_idx, float* out_dist, void* stream);

/* Ascending top-k of d(q_i, p_j) per query row, d evaluated with the reference's operation
 * sequence on top of an fp32-MFMA dot product (candidates), then re-scored exactly (rescore != 0: knnsvc_knn_rescore).
 * Ties: lower pool index first.  Indices are written
 * as idx_offset + j (so that a pool shard reports global rows).  k <= 32.
 * [mask_lo, mask_hi): local pool rows whose distance is replaced by exactly 1 before selection — the
 * self-matching rule of per_spk_extract (ddsp_prematch_dataset.py:1606-1607, `dists[:, start:end] = 1`);
 * mask_lo >= mask_hi disables it. */
int knnsvc_knn_topk(const float* q, const float* q_norm, const float* q_sq, int64_t nq,
                    const float* pool, const float* p_norm, const float* p_sq, int64_t np,
                    int32_t dim, int32_t k, int64_t idx_offset, int64_t mask_lo, int64_t mask_hi,
                    int64_t* out_idx, float* out_dist, void* workspace, size_t workspace_bytes,
                    int32_t* nan_flag, int32_t rescore, void* stream);

/* Second half of the two-kernel kNN route: `dots`[nq, np] (row pitch ld) holds q.p^T computed by knnsvc_conv_gemm
 * on the emulated-fp32 matrix-core path (q as the A operand, the pool rows as pre-split "weights"); this replays the
 * reference's distance formula (lib_ongaku_test.py:148-175) on every entry and keeps each row's WIDE list (wide_out [nq][64],
 * see knnsvc_knn_rescore, which turns it into the top-k) with the same (distance, lower index) order and NaN flag
 * semantics as knnsvc_knn_topk. */
int knnsvc_knn_select(const float* dots, int64_t ld, const float* q_norm, const float* q_sq, int64_t nq,
                      const float* p_norm, const float* p_sq, int64_t np, int32_t k,
                      int64_t mask_lo, int64_t mask_hi,
                      void* wide_out, int32_t* nan_flag, void* stream);

/* Fused route (no [nq, np] dot matrix in HBM; lib_ongaku_test.py:148-175 + ddsp_prematch_dataset.py:1195-1210: the reference's
 * 20-row cdist + topk loop over the whole pool).  The caller walks the pool's rows in EPOCHS [p_base, p_base + np) of growing
 * size (knn_svc_amd/ops.py: knn_epochs); each epoch is one knnsvc_knn_screen + one knnsvc_knn_refine.
 * knnsvc_knn_screen: q.p^T on the f16x2 matrix-core path (both operands pre-split by knnsvc_split_f16x2_dyn with the range
 * slots q_absmax / p_absmax; p_f16x2 / p_norm / p_sq point at the epoch's first row), every product screened in registers: first
 * conservatively against the row's threshold, then — the few survivors — exactly: the reference's distance of the pair as a
 * (distance, pool index) key.
 *   thr == thr_idx == NULL (the first epoch): every 256 x 256 tile bounds its rows itself — at least 32 of its columns lie at
 *     or below the bound it derives, so no row's top-k (k <= 32) can lie beyond it;
 *   otherwise (thr[row], thr_idx[row]) = the row's threshold key over the rows searched so far (knnsvc_knn_refine's thr_out /
 *     thr_idx_out: the k-th best distance + the guard band of the exact re-score; the index in the chunk's index space, i.e.
 *     without idx_offset): exactly the pairs with key <= it pass.
 * Survivors are appended to cand[row][0 .. cap) as (pool index = p_base + row in epoch, order-preserving bits of the distance;
 * 0xFFFFFFFF = NaN), 8 bytes each; cand_count[row] counts them (the caller zeroes cand_count and the flag once; refine
 * resets the counts).  Bit 1 of *overflow_flag set afterwards: some row had more than cap survivors — the caller must fall
 * back to the dot-matrix route (the same word and bit as knnsvc_knn_refine's `flags`: one flag per search).  mask_lo / mask_hi: chunk rows [lo, hi) compete at distance exactly 1.
 * cold_ws (first epoch only, else NULL; nq * (1 + 2 * ceil(np / 256)) + 32 * ceil(nq / 256) words the caller zeroes): [nq] the best
 * bound found for each row — knnsvc_knn_refine's row_bound: it starts from it and drops the survivors of weaker tiles unseen —,
 * then the per-half-tile bounds and arrival counters through which the tiles of a row tile, running side by side, tighten each
 * other's bounds (a bounded wait; only tightness depends on it, never the result).
 * knnsvc_knn_refine: the row's wide list so far (`wide` [nq][64], read when has_prev != 0: the previous refine's output) + the
 * candidates -> the wide list, written back to `wide` — after the last epoch identical to knnsvc_knn_select's over every
 * pair; knnsvc_knn_rescore turns it into the top-k.  flags: bit 0 = a NaN distance was met, bit 1 (final_pass only) = a row
 * ended with fewer than k entries although no NaN was seen (the caller repeats the search on the dot-matrix route).
 * thr_out / thr_idx_out (NULL after the last epoch): the next epoch's thresholds (the k-th distance + the guard band).
 * nq * dim and np * dim below 2^28 per call (chunk larger searches).
 * max_blocks: the screen kernel is persistent (a block walks tiles); 0 = one block per CU, otherwise at most this many blocks
 * (rounded down to a multiple of 8), so that a search inside a stream pipeline leaves CUs to the other streams' kernels. */
int knnsvc_knn_screen(const void* q_f16x2, const float* q_absmax, const float* q_norm, const float* q_sq, int64_t nq,
                      const void* p_f16x2, const float* p_absmax, const float* p_norm, const float* p_sq, int64_t np,
                      int32_t dim, const float* thr, const int64_t* thr_idx, int64_t mask_lo, int64_t mask_hi, int64_t p_base,
                      int32_t* cand_count, void* cand, int32_t cap, uint32_t* cold_ws, int32_t* overflow_flag, int32_t max_blocks,
                      void* stream);
int knnsvc_knn_refine(int32_t* cand_count, const void* cand, int32_t cap, int64_t nq, int32_t k,
                      const uint32_t* row_bound, int32_t has_prev, void* wide, float* thr_out,
                      int64_t* thr_idx_out, int32_t final_pass, int32_t* flags, void* stream);

/* Merge `parts` per-shard top-k lists ([parts][nq][k], e.g. after an RCCL all-gather) into one. */
int knnsvc_knn_merge(const float* part_dist, const int64_t* part_idx, int32_t parts, int64_t nq,
                     int32_t k, int64_t* out_idx, float* out_dist, void* stream);

/* ------------------------------------------------------------------------------------------
 * Neighbour post-processing.
 * ------------------------------------------------------------------------------------------ */
/* lower median of log(f0) over voiced frames (torch.median, ddsp_prematch_dataset.py:1224-1225);
 * result[0] = median, result[1] = voiced count (as float).  workspace: n floats. */
int knnsvc_log_f0_median(const float* f0, int64_t n, float* result, float* workspace, void* stream);

/* shifted[i] = f0[i] ? exp(log f0[i] + (pool_median - query_median)) : 0   (:1232-1233) */
int knnsvc_shift_f0(const float* f0, int64_t n, const float* query_median, const float* pool_median,
                    float* shifted, void* stream);

/* sort_by_f0_compatibility (ddsp_prematch_dataset.py:954-1016): stable ascending re-order of each
 * row's k neighbours by |log2(pool_f0[idx]+1e-5) - log2(shifted[i]+1e-5)|. k <= 64. */
int knnsvc_f0_rerank(const int64_t* nn_idx, int64_t nq, int32_t k, const float* shifted_f0,
                     const float* pool_f0, int64_t* out_idx, void* stream);

/* knn_with_concat_cost (lib_ongaku_test.py:270-369), frame-sequential, one workgroup per sequence.
 * idx_in/out [nq, 4]; use_f0 selects the pitched variant. */
int knnsvc_concat_reselect(const int64_t* idx_in, const float* q, const float* q_norm, int64_t nq,
                           const float* pool, const float* p_norm, int64_t np, int32_t dim,
                           const float* shifted_f0, const float* pool_f0, int32_t use_f0,
                           float concat_weight, int64_t* idx_out, void* stream);

/* compute_wavlm_weight / compute_extended_weight (ddsp_prematch_dataset.py:574-680, 807-924):
 * Adam(amsgrad) on softmax weights with the reference's stopping rules, run entirely on the
 * device (no per-iteration host sync).  scale = 0.1 (WavLM) or 1000 (harmonics).
 * out_w [nq,4]; out_iters[0] = iterations executed.  workspace from knnsvc_smooth_workspace_bytes.
 * max_iter > 0: the reference's cap (100 000 there).  max_iter < 0 (measurement aid): exactly -max_iter iterations
 * with the stopping rules switched off (bench.py's sensitivity figure for data on which the loops run long). */
/* row_scale (may be NULL) [nq,4]: compute_weight_with_amp (ddsp_prematch_dataset.py:684-804) — candidate k of
 * frame t and its two neighbours are multiplied by row_scale[t,k] (the amp_ratio of per_spk_extract) before the
 * loss is formed; with scale = 1000 this is the prematch weight optimisation. */
size_t knnsvc_smooth_workspace_bytes(int64_t nq);
int knnsvc_smooth_weights(const int64_t* idx, int64_t nq, const float* pool, int64_t np, int32_t dim,
                          int32_t ld, float scale, const float* row_scale, int32_t max_iter, float* out_w,
                          int32_t* out_iters, void* workspace, size_t workspace_bytes, void* stream);

/* out[i,:] = sum_k w[i,k] * pool[idx[i,k], :]   (ddsp_prematch_dataset.py:1358, 1444; w NULL = mean of 4
 * for harmonics :1446 / softmax(ones) :1361-1364) */
int knnsvc_weighted_gather(const int64_t* idx, const float* w, int64_t nq, int32_t k, const float* pool,
                           int32_t dim, int32_t ld, int32_t mean_mode, float* out, void* stream);

/* ------------------------------------------------------------------------------------------
 * Prematch (training-pool generation, ddsp_prematch_dataset.py:1464-1772 per_spk_extract).
 * ------------------------------------------------------------------------------------------ */
/* out = float(half(x)) element-wise, round-to-nearest-even (`.half().float()`, :1509, 1561, 1592) */
int knnsvc_round_f16(const float* x, int64_t n, float* out, void* stream);
/* amp_ratio[t,k] = ||spec_q[t,:]||_1 / (||spec_pool[idx[t,k],:]||_1 + 1e-5)   (:1657-1660) */
int knnsvc_amp_ratio(const float* spec_q, int32_t ld_q, const float* spec_pool, int32_t ld_pool, int64_t np,
                     const int64_t* idx, int64_t nq, int32_t k, int32_t bins, float* out, void* stream);

/* ------------------------------------------------------------------------------------------
 * f0 front end.  The reference runs pyworld.harvest(f0_floor 65, f0_ceil 1047, frame period 20 ms) and
 * zeroes values below 80 Hz when no `<stem>_f0.npy` exists (ddsp_prematch_dataset.py:121-128, 376-379).
 * ------------------------------------------------------------------------------------------ */
/* Harvest (M. Morise, Interspeech 2017) — what pyworld.harvest computes where the reference calls it
 * (ddsp_prematch_dataset.py:121-128: fs = 16000, f0_floor = 65, f0_ceil = 1047, frame_period = 20 ms, then `f0[f0 < 80] = 0`;
 * :376-379 when `<stem>_f0.npy` is missing).  x: [L] fp32 at 16 kHz on the device; f0: [n_frames] fp32,
 * n_frames = (int)(1000 L / fs / frame_period) + 1, 0 = unvoiced.  All stages run in fp64 on `stream` inside `workspace`
 * (256-byte aligned device memory of at least the size knnsvc_f0_harvest_workspace reports: ~35 MB per second of audio);
 * no host synchronisation.  status (device int32, may be NULL): 0, or a bit mask if a fixed-capacity list overflowed
 * (1: more than 16 candidates in a frame, 2 / 4: section storage) — the track is then incomplete.
 * Restates the published algorithm, pinned by oracle/f0_ref.py on the reference's two shipped harvest tracks. */
int knnsvc_f0_harvest_workspace(int64_t L, int32_t sample_rate, float f0_floor, float f0_ceil, float frame_period,
                                int64_t* n_frames, int64_t* bytes);
int knnsvc_f0_harvest(const float* x, int64_t L, int32_t sample_rate, float f0_floor, float f0_ceil, float frame_period,
                      float zero_below, float* f0, int64_t n_frames, void* workspace, int64_t workspace_bytes,
                      int32_t* status, void* stream);

/* ------------------------------------------------------------------------------------------
 * FLAC container (HOST functions: plain pointers into host memory, no stream, no GPU).  The reference reads .flac through
 * torchaudio.load (ddsp_prematch_dataset.py:332; its prematch builder globs *.wav and *.flac, :1469-1473) and writes .flac
 * through pydub / ffmpeg (lib_ongaku_test.py:122-143).  RFC 9639; every frame's CRC-8 / CRC-16 is verified while decoding.
 * ------------------------------------------------------------------------------------------ */
/* STREAMINFO of a FLAC file image: sample rate, channels, bits per sample, samples per channel, MD5 of the PCM (16 bytes, may be NULL) */
int knnsvc_flac_info(const uint8_t* data, int64_t size, int32_t* sample_rate, int32_t* channels, int32_t* bits,
                     int64_t* total_samples, uint8_t* md5);
/* decode into out[channels][capacity] (planar, sign-extended to int32); *decoded = samples per channel */
int knnsvc_flac_decode(const uint8_t* data, int64_t size, int32_t* out, int64_t capacity, int64_t* decoded);
/* encode pcm[channels][n] (`bits`-bit signed samples in int32; bits in {8, 12, 16, 20, 24}) with fixed predictors + Rice coding,
 * block size 4096; md5: 16 bytes for STREAMINFO or NULL (= unknown); *size = bytes written to out (capacity: 5 n channels + 8192 is safe) */
int knnsvc_flac_encode(const int32_t* pcm, int32_t channels, int64_t n, int32_t bits, int32_t sample_rate, const uint8_t* md5,
                       uint8_t* out, int64_t capacity, int64_t* size);

/* ------------------------------------------------------------------------------------------
 * Pool side features and the additive synthesiser.
 * ------------------------------------------------------------------------------------------ */
/* reflect-pad by `pad` samples each side (torch.stft center=True) : out[n + 2*pad] */
int knnsvc_reflect_pad(const float* x, int64_t n, int32_t pad, float* out, void* stream);
/* the same for a batch of signals packed back to back: item b = x[offs[b] .. offs[b+1]) (offs: device int64 [batches+1]) ->
 * row b of out [batches, stride], zero from n_b + 2*pad on.  One launch per pool instead of one per file
 * (ddsp_prematch_dataset.py:361 runs torchaudio's Spectrogram file by file). */
int knnsvc_reflect_pad_batch(const float* x, const int64_t* offs, int32_t batches, int32_t pad, float* out, int64_t stride,
                             void* stream);
/* knnsvc_complex_mag + knnsvc_harmonic_amps in one pass over the DFT product (ddsp_prematch_dataset.py:361, 391-404):
 * reim [rows, ld >= 2*bins], f0 [rows] -> spec [rows, bins], harm [rows, n_harm]; results identical to the two calls. */
int knnsvc_spec_harm(const float* reim, int64_t rows, int32_t bins, int32_t ld, const float* f0, int32_t n_harm,
                     float* spec, float* harm, void* stream);
/* |re + i im| of a [rows, 2*bins] (cos block | sin block) DFT product -> [rows, bins] */
int knnsvc_complex_mag(const float* reim, int64_t rows, int32_t bins, int32_t ld, float* out, void* stream);
/* harmonic amplitudes (ddsp_prematch_dataset.py:391-404): spec [T,200], f0 [T] -> harm [T,49] */
int knnsvc_harmonic_amps(const float* spec, const float* f0, int64_t T, int32_t bins, int32_t n_harm,
                         float* harm, void* stream);
/* get_bulk_dsp_choral (ddsp_prematch_dataset.py:165-208) fused with sin_prenet
 * (hifigan/ddsp_models.py:416,476): f0 [N], amp [N,H] -> cond [N*hop, n_ch] channel-last, and
 * optionally the raw excitation exc [N*hop].  mode 0 = additive (mix), 1 = plain sine of f0
 * (hifigan/ddsp_models_f0.py:348-356; amp ignored).  frame_phase: N doubles of workspace. */
/* n_dyn (may be NULL): DEVICE int32 — the valid frame count (<= N): amplitude taps clamp at frame n - 1 and samples past n * hop
 * are left alone, so a launch sized for a bucket of N frames reproduces the exact-length result (see knnsvc_conv_desc.n_dyn). */
int knnsvc_additive_synth(const float* f0, const float* amp, int64_t N, int32_t H, int32_t hop,
                          int32_t sample_rate, int32_t mode, const float* prenet_w, const float* prenet_b,
                          int32_t n_ch, float* cond, int32_t ld_cond, float* exc, double* frame_phase,
                          const int32_t* n_dyn, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* KNNSVC_HIP_H */
