// Whole-model entry points (include/knnsvc_hip.h, "Whole-model entry points"): the layer SEQUENCE of a network behind one C call.
// Host code only — every launch goes through the library's own extern "C" entry points, with the arguments the Python host
// (knn_svc_amd/wavlm.py, round 1-4) passed launch by launch, so the results are the same bits.
//
// knnsvc_wavlm_encode = WavLM.extract_features up to the exit layer (wavlm/WavLM.py:323-375): conv feature extractor
// (ConvFeatureExtractionModel.forward, :485-504), LayerNorm + post_extract_proj (:342-348), TransformerEncoder.extract_features
// (:572-612: positional conv, layer loop) with the pre-LN layer of :691-714 and the gated relative-position attention of
// wavlm/modules.py:457-564.
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "common.h"

namespace {

constexpr size_t SLOT_FLOATS = 64 * 32;            // a range slot: 64 stripes of one cache line (include/knnsvc_hip.h, "Range")

// Zero-fill of the range slots as an ORDINARY KERNEL on the caller's stream.  Round 5 first used hipMemsetAsync here: with three
// generators in flight on three (high-priority) streams the dataset-mode pipeline then wrote different samples from run to run
// (tests/test_gpu_models.py::test_bulk_match_pipelined_vocoder_equals_sequential: 3-5 failures in 5 runs; 0 in 16 with this kernel,
// 0 in 15 with the host-sequenced forward, which zeroes its slots with a torch fill kernel).  The async memset was not ordered
// with the kernels that follow it on the same stream the way a kernel is.
__global__ __launch_bounds__(256) void zero_fill_kernel(f32x4* __restrict__ p, long n4) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n4) p[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
}
int zero_slots(float* p, size_t floats, void* st) {
    const long n4 = (long)(floats / 4);
    hipLaunchKernelGGL(zero_fill_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)st, (f32x4*)p, n4);
    return knnsvc_check_launch("zero_slots");
}

struct WavLM {
    knnsvc_wavlm_desc d;
    std::vector<knnsvc_wavlm_conv> conv;
    std::vector<knnsvc_wavlm_layer> layers;
    std::vector<float> mix;
};

// bump allocator over the caller's workspace; with base == nullptr it only measures
struct Arena {
    char* base; size_t off = 0, cap;
    Arena(void* b, size_t c) : base((char*)b), cap(c) {}
    float* f(size_t floats) {
        const size_t o = off;
        off += (floats * 4 + 255) / 256 * 256;
        return base ? (float*)(base + o) : nullptr;
    }
};

int64_t frames_of(const WavLM& m, int64_t L) {
    int64_t n = L;
    for (const auto& c : m.conv) {
        if (n < c.k) return 0;                       // shorter than the receptive field
        n = (n - c.k) / c.stride + 1;
    }
    return n;
}

knnsvc_conv_desc base_desc() {
    knnsvc_conv_desc c;
    memset(&c, 0, sizeof(c));
    c.taps = 1; c.stride = 1; c.dil = 1; c.a_slope = 1.0f; c.div = 1.0f; c.batches = 1; c.groups = 1;
    return c;
}
void set_w(knnsvc_conv_desc& c, const knnsvc_weight& w) { c.w = w.w; c.w_f16x2 = w.w_f16x2; c.w_f16x2_scale = w.w_f16x2 ? w.w_f16x2_scale : 0.f; }

// out[M, N] = act(x[M, K] @ w[N, K]^T + bias) (+ resid): ops.linear
int linear(const float* x, const knnsvc_weight& w, const float* bias, float* out, int64_t M, int N, int K, int act, const float* resid,
           bool x_split, int out_split, const float* x_absmax, float* out_absmax, void* st) {
    knnsvc_conv_desc c = base_desc();
    c.x = x; c.ldx = K; c.t_in = (int32_t)M; c.cin = K; set_w(c, w); c.n = N; c.bias = bias; c.out = out; c.ldo = N; c.m = (int32_t)M;
    c.act = act; c.resid = resid; c.ldr = resid ? N : 0; c.x_f16x2 = x_split ? 1 : 0; c.out_f16x2 = out_split;
    c.x_absmax = x_absmax; c.out_absmax = out_absmax;
    return knnsvc_conv_gemm(&c, st);
}

// the sequence; `ar` hands out the activations (a measuring arena: nothing is launched)
int encode(const WavLM& m, const float* wav, int B, int64_t L, const int32_t* lens, const float* table, float* out, Arena& ar, void* st) {
    const bool run = ar.base != nullptr;
    const knnsvc_wavlm_desc& d = m.d;
    const int E = d.E, H = d.H;
    // range slots: one zeroed block, handed out in order (the Python host made a fresh zeroed tensor per slot)
    const int n_slots = 2 + (int)m.conv.size() + 4 * d.n_layers;
    float* slots = ar.f((size_t)n_slots * SLOT_FLOATS);
    int next_slot = 0, rc0 = 0;
    if (run && (rc0 = zero_slots(slots, (size_t)n_slots * SLOT_FLOATS, st)) != 0) return rc0;
    auto new_slot = [&]() -> float* { float* s = run ? slots + (size_t)next_slot * SLOT_FLOATS : nullptr; ++next_slot; return s; };
    auto slot_of = [&](const float* x, int64_t rows, int cols, int ld, float*& s) -> int {      // ops.absmax into a fresh slot
        s = new_slot();
        return run ? knnsvc_absmax(x, rows, cols, ld, s, st) : 0;
    };
    int rc = 0;
#define KN_RUN(CALL) do { if (run) { rc = (CALL); if (rc) return rc; } } while (0)

    // ---- conv feature extractor: activations ping-pong between two buffers (each LayerNorm runs in place)
    const float* x = wav;
    int64_t t_in = L; int cin = 1;
    bool x_sp = false;
    size_t need[2] = {0, 0};
    {
        int64_t t = L;
        for (size_t li = 0; li < m.conv.size(); ++li) {
            KN_REQUIRE(t >= m.conv[li].k, "wavlm_encode: chunk shorter than the extractor's receptive field");
            t = (t - m.conv[li].k) / m.conv[li].stride + 1;
            const size_t fl = (size_t)B * t * m.conv[li].dim;
            need[li & 1] = fl > need[li & 1] ? fl : need[li & 1];
        }
    }
    float* pp[2] = {ar.f(need[0]), ar.f(need[1])};
    for (size_t li = 0; li < m.conv.size(); ++li) {
        const knnsvc_wavlm_conv& c = m.conv[li];
        const int64_t t_out = (t_in - c.k) / c.stride + 1;
        KN_REQUIRE(t_out > 0, "wavlm_encode: chunk shorter than the extractor's receptive field");
        float* y = pp[li & 1];
        const bool fused0 = li == 0 && cin == 1 && (c.dim == 64 || c.dim == 128 || c.dim == 256 || c.dim == 512) && c.k <= 16 && c.stride <= 8;
        if (fused0) {
            x_sp = c.out_split && c.dim % 32 == 0 && c.dim >= 256 && m.conv.size() > 1;
            KN_RUN(knnsvc_wavlm_conv0(x, B, t_in, c.w.w, c.dim, c.k, c.stride, c.ln_g, c.ln_b, y, x_sp ? 1 : 0, st));
        } else {
            float* xs = nullptr;
            if (!x_sp && cin % 32 == 0) { rc = slot_of(x, (int64_t)B * t_in, cin, cin, xs); if (rc) return rc; }
            knnsvc_conv_desc g = base_desc();
            g.x = x; g.x_bstride = t_in * cin; g.ldx = cin; g.t_in = (int32_t)t_in; g.cin = cin; g.taps = c.k; g.stride = c.stride;
            set_w(g, c.w); g.n = c.dim; g.out = y; g.o_bstride = t_out * c.dim; g.ldo = c.dim; g.m = (int32_t)t_out; g.batches = B;
            g.x_f16x2 = x_sp ? 1 : 0; g.x_absmax = xs;
            KN_RUN(knnsvc_conv_gemm(&g, st));
            x_sp = c.out_split && c.dim % 32 == 0 && li + 1 < m.conv.size();      // the last layer's output feeds a LayerNorm, not a GEMM
            KN_RUN(knnsvc_layernorm(y, (int64_t)B * t_out, c.dim, c.dim, c.ln_g, c.ln_b, 1 | (x_sp ? 2 : 0), y, c.dim, st));
        }
        x = y; t_in = t_out; cin = c.dim;
    }
    const int64_t T = t_in, R = (int64_t)B * T;
    KN_REQUIRE(!x_sp && R * (int64_t)(d.ffn > 3 * E ? d.ffn : 3 * E) < (1ll << 31), "wavlm_encode: batch too large");
    // ---- LayerNorm + projection + positional conv
    const bool f_sp = d.feats_split && cin % 32 == 0;
    float* feats = ar.f((size_t)R * cin);
    KN_RUN(knnsvc_layernorm(x, R, cin, cin, d.ln_g, d.ln_b, f_sp ? 2 : 0, feats, cin, st));
    float* fs = nullptr;
    if (!f_sp) { rc = slot_of(feats, R, cin, cin, fs); if (rc) return rc; }
    float* x_slot = d.pos_a_scale > 0.f ? nullptr : new_slot();
    float* xa = ar.f((size_t)R * E);
    float* xb = ar.f((size_t)R * E);
    KN_RUN(linear(feats, d.proj, d.proj_b, xa, R, E, cin, KNNSVC_ACT_NONE, nullptr, f_sp, 0, fs, x_slot, st));
    if (lens) KN_RUN(knnsvc_mask_rows(xa, B, (int32_t)T, E, E, lens, st));      // x[padding_mask] = 0 before the positional conv (WavLM.py:574-575)
    {
        const int G = d.pos_groups, K = d.pos_k, cg = E / G;
        knnsvc_conv_desc g = base_desc();
        g.x = xa; g.x_bstride = T * E; g.x_gstride = cg; g.ldx = E; g.t_in = (int32_t)T; g.cin = cg; g.taps = K; g.pad = K / 2;
        set_w(g, d.pos); g.w_gstride = (int64_t)cg * cg * K; g.n = cg; g.bias = d.pos_b; g.bias_gstride = cg;
        g.out = xb; g.o_bstride = T * E; g.o_gstride = cg; g.ldo = E; g.m = (int32_t)T; g.act = KNNSVC_ACT_GELU;
        g.resid = xa; g.r_bstride = T * E; g.r_gstride = cg; g.ldr = E; g.batches = B; g.groups = G;
        g.x_absmax = x_slot; g.a_f16x2_scale = d.pos_a_scale;
        KN_RUN(knnsvc_conv_gemm(&g, st));
    }
    float* cur = xb;                       // the residual stream alternates between xa / xb / a third buffer (two new tensors per layer)
    float* spare[2] = {xa, ar.f((size_t)R * E)};
    float* acc = nullptr;
    const bool mixed = !m.mix.empty();
    if (mixed) {
        acc = ar.f((size_t)R * E);
        KN_RUN(knnsvc_axpy(cur, R * E, m.mix[0], 0, acc, st));                     // layer_results[0]: the encoder's input (WavLM.py:583-585)
    }
    float* xn = ar.f((size_t)R * E);
    float* gate = ar.f((size_t)R * H);
    float* qkv = ar.f((size_t)R * 3 * E);
    float* att = ar.f((size_t)R * E);
    float* hmid = ar.f((size_t)R * d.ffn);
    for (int l = 0; l < d.n_layers; ++l) {
        const knnsvc_wavlm_layer& ly = m.layers[l];
        const bool last = l == d.n_layers - 1;
        const bool e_sp = ly.xn_split && E % 32 == 0;
        KN_RUN(knnsvc_layernorm(cur, R, E, E, ly.ln1_g, ly.ln1_b, e_sp ? 2 : 0, xn, E, st));
        KN_RUN(knnsvc_wavlm_gate(xn, R, H, 64, E, ly.gate_w, ly.gate_b, ly.grep_a, gate, e_sp ? 1 : 0, st));
        // K and V leave the projection pre-split (every query block of a head re-split the same keys otherwise); Q stays fp32
        const bool narrow = ly.attn_f16 != 0;
        const bool kv_sp = E % 32 == 0 && narrow;
        float* s1 = nullptr;
        if (!e_sp) { rc = slot_of(xn, R, E, E, s1); if (rc) return rc; }
        KN_RUN(linear(xn, ly.wqkv, ly.bqkv, qkv, R, 3 * E, E, KNNSVC_ACT_NONE, nullptr, e_sp, kv_sp ? E : 0, s1, nullptr, st));
        const bool a_sp = E % 32 == 0 && narrow;                                   // attention output <= max|V|: same bound
        KN_RUN(knnsvc_wavlm_attention(qkv, gate, table, B, (int32_t)T, H, att, (a_sp ? 1 : 0) | (narrow ? 0 : 4), kv_sp ? 1 : 0, lens, st));
        float* s2 = nullptr;
        if (!a_sp) { rc = slot_of(att, R, E, E, s2); if (rc) return rc; }
        float* x1 = spare[0];
        KN_RUN(linear(att, ly.wo, ly.bo, x1, R, E, E, KNNSVC_ACT_NONE, cur, a_sp, 0, s2, nullptr, st));
        const bool e2_sp = ly.xn2_split && E % 32 == 0;
        KN_RUN(knnsvc_layernorm(x1, R, E, E, ly.ln2_g, ly.ln2_b, e2_sp ? 2 : 0, xn, E, st));
        const bool h_sp = ly.h_split && d.ffn % 32 == 0;
        float* h_slot = h_sp ? nullptr : new_slot();
        float* s3 = nullptr;
        if (!e2_sp) { rc = slot_of(xn, R, E, E, s3); if (rc) return rc; }
        KN_RUN(linear(xn, ly.w1, ly.b1, hmid, R, d.ffn, E, KNNSVC_ACT_GELU, nullptr, e2_sp, h_sp ? 1 : 0, s3, h_slot, st));
        float* x2 = (last && !mixed) ? out : spare[1];                             // the exit layer's FFN2 writes the caller's buffer
        KN_RUN(linear(hmid, ly.w2, ly.b2, x2, R, E, d.ffn, KNNSVC_ACT_NONE, x1, h_sp, 0, h_slot, nullptr, st));
        spare[0] = cur; spare[1] = x1;                                             // both are dead now
        cur = x2;
        if (mixed && m.mix[l + 1] != 0.0f) KN_RUN(knnsvc_axpy(cur, R * E, m.mix[l + 1], 1, acc, st));
    }
    // (copies as kernels on the caller's stream — x * 1.0f is exact —, not hipMemcpyAsync: see zero_slots)
    if (mixed) KN_RUN(knnsvc_axpy(acc, R * E, 1.0f, 0, out, st));
    else if (d.n_layers == 0) KN_RUN(knnsvc_axpy(cur, R * E, 1.0f, 0, out, st));
    if (next_slot > n_slots) return knnsvc_fail(KNNSVC_EINVAL, "wavlm_encode: slot plan exceeded (%d > %d)", next_slot, n_slots);
#undef KN_RUN
    return KNNSVC_OK;
}

}  // namespace

extern "C" int knnsvc_wavlm_create(const knnsvc_wavlm_desc* d, void** handle) {
    KN_REQUIRE(d && handle, "wavlm_create: null pointer");
    KN_REQUIRE(d->n_conv >= 1 && d->n_layers >= 0 && d->conv && (d->layers || d->n_layers == 0), "wavlm_create: layer arrays");
    KN_REQUIRE(d->E > 0 && d->H > 0 && d->E == d->H * 64, "wavlm_create: the attention kernel is built for head_dim 64");
    KN_REQUIRE(d->ln_g && d->ln_b && d->proj.w && d->pos.w && d->pos_groups > 0 && d->E % d->pos_groups == 0 && d->pos_k > 0, "wavlm_create: missing weights");
    WavLM* m = new WavLM();
    m->d = *d;
    m->conv.assign(d->conv, d->conv + d->n_conv);
    if (d->n_layers) m->layers.assign(d->layers, d->layers + d->n_layers);
    if (d->layer_mix) m->mix.assign(d->layer_mix, d->layer_mix + d->n_layers + 1);
    m->d.conv = nullptr; m->d.layers = nullptr; m->d.layer_mix = nullptr;
    *handle = m;
    return KNNSVC_OK;
}

extern "C" int knnsvc_wavlm_free(void* handle) {
    delete (WavLM*)handle;
    return KNNSVC_OK;
}

extern "C" int64_t knnsvc_wavlm_frames(const void* handle, int64_t L) {
    return handle ? frames_of(*(const WavLM*)handle, L) : -1;
}

extern "C" size_t knnsvc_wavlm_workspace_bytes(const void* handle, int32_t batches, int64_t L) {
    if (!handle || batches <= 0 || L <= 0 || frames_of(*(const WavLM*)handle, L) <= 0) return 0;
    Arena ar(nullptr, 0);
    if (encode(*(const WavLM*)handle, nullptr, batches, L, nullptr, nullptr, nullptr, ar, nullptr)) return 0;
    return ar.off;
}

extern "C" int knnsvc_wavlm_encode(const void* handle, const float* wav, int32_t batches, int64_t L, const int32_t* lens,
                                   const float* table, float* out, void* workspace, size_t workspace_bytes, void* stream) {
    KN_REQUIRE(handle && wav && table && out && workspace, "wavlm_encode: null pointer");
    KN_REQUIRE(batches > 0 && L > 0, "wavlm_encode: empty batch");
    KN_REQUIRE(((uintptr_t)workspace & 255) == 0, "wavlm_encode: workspace must be 256-byte aligned");
    const size_t need = knnsvc_wavlm_workspace_bytes(handle, batches, L);
    if (need == 0) return knnsvc_fail(KNNSVC_EINVAL, "wavlm_encode: chunk shorter than the extractor's receptive field");
    if (workspace_bytes < need) return knnsvc_fail(KNNSVC_EWORKSPACE, "wavlm_encode: workspace %zu < %zu bytes", workspace_bytes, need);
    Arena ar(workspace, workspace_bytes);
    return encode(*(const WavLM*)handle, wav, batches, L, lens, table, out, ar, stream);
}

// -------------------------------------------------------------------------------------------------
// knnsvc_generator_forward = SynthesizerTrn.forward + Generator.forward (hifigan/ddsp_models.py:108-233, 405-493; the 'f0' variant
// hifigan/ddsp_models_f0.py:106-216, 320-381): excitation (additive synth or sine) + sin_prenet, the side (down) path, lin_pre +
// conv_pre + concat_pre, four [transposed conv -> concat_conv -> three ResBlocks -> mean] stages, conv_post + tanh.  The launch
// sequence — concat buffers written in place, one range slot per logical tensor, the three ResBlock branches of a step as ONE
// grid — is the one knn_svc_amd/vocoder.py issued launch by launch (Vocoder._forward_host); same arguments, same bits.
// -------------------------------------------------------------------------------------------------
namespace {

constexpr float LRELU = 0.1f;

struct Generator {
    knnsvc_generator_desc d;
    std::vector<knnsvc_gen_stage> st;
};

struct Dyn { const int32_t* n; int64_t nb; };       // device frame count + the bucket's frame count (NULL: exact length)

void set_dyn(knnsvc_conv_desc& c, const Dyn& dy) {
    if (!dy.n) return;
    c.n_dyn = dy.n;
    c.dyn_t_in_mul = (int32_t)(c.t_in / dy.nb); c.dyn_t_in_add = (int32_t)(c.t_in % dy.nb);
    c.dyn_m_mul = (int32_t)(c.m / dy.nb); c.dyn_m_add = (int32_t)(c.m % dy.nb);
    c.dyn_t_out_mul = c.convt_u ? (int32_t)(c.t_out / dy.nb) : 0;
}

// Vocoder._conv: a channel-last conv1d through knnsvc_conv_gemm
knnsvc_conv_desc conv_desc(const float* x, const knnsvc_weight& w, float* out, int64_t t_in, int cin, int cout, int k, int64_t m, const Dyn& dy) {
    knnsvc_conv_desc c = base_desc();
    c.x = x; c.ldx = cin; c.t_in = (int32_t)t_in; c.cin = cin; c.taps = k; set_w(c, w); c.n = cout; c.out = out; c.ldo = cout; c.m = (int32_t)m;
    set_dyn(c, dy);
    return c;
}

bool pair_ok(int channels, int taps, int dil) { return (channels == 32 || channels == 64) && (taps & 1) && taps <= 11 && dil * (taps - 1) <= 64; }

int generate(const Generator& g, const float* cf, const float* f0, const float* harm, int64_t N, const int32_t* n_dyn, float* out, Arena& ar, void* st) {
    const bool run = ar.base != nullptr;
    const knnsvc_generator_desc& d = g.d;
    const int n_up = d.n_up, uic = d.uic, hop = d.hop;
    const int64_t L = N * hop;
    const Dyn dy{n_dyn, N};
    int rc = 0;
#define KN_RUN(CALL) do { if (run) { rc = (CALL); if (rc) return rc; } } while (0)
    constexpr int N_SLOTS = 256;
    float* slots = ar.f((size_t)N_SLOTS * SLOT_FLOATS);
    if (run && (rc = zero_slots(slots, (size_t)N_SLOTS * SLOT_FLOATS, st)) != 0) return rc;
    int n_slot = 0;
    auto slot = [&]() -> float* { float* s = run ? slots + (size_t)n_slot * SLOT_FLOATS : nullptr; ++n_slot; return s; };
    // lengths of the time axis at each level of the side path: lens[0] = L ... lens[n_up] = N
    int64_t lens[9];
    lens[0] = L;
    for (int i = 0; i < n_up; ++i) lens[i + 1] = lens[i] / g.st[i].down_u;
    KN_REQUIRE(lens[n_up] == N, "generator_forward: the down path's strides do not multiply to the hop");
    // concat buffers: up stage i consumes cat[i] = [ups_i output | res[n_up - 1 - i]]
    float* cat[8]; float* cat_slot[8]; int cat_ld[8];
    for (int i = 0; i < n_up; ++i) {
        const int ch = uic >> (i + 1);
        cat_ld[i] = ch + d.side[n_up - 1 - i];
        cat[i] = ar.f((size_t)lens[n_up - 1 - i] * cat_ld[i]);
        cat_slot[i] = slot();
    }
    const int pre_ld = uic + d.side[n_up];
    float* cat_pre = ar.f((size_t)N * pre_ld);
    float* cat_pre_slot = slot();
    struct View { float* p; int ld, ch; float* s; };
    auto res_view = [&](int level) -> View {
        if (level == n_up) return {cat_pre ? cat_pre + uic : nullptr, pre_ld, d.side[level], cat_pre_slot};
        const int b = n_up - 1 - level, ch = uic >> (n_up - level);
        return {cat[b] ? cat[b] + ch : nullptr, cat_ld[b], d.side[level], cat_slot[b]};
    };
    // ---- head of the main path: input projection + conv_pre -> cat_pre[:, :uic]
    float* s_x0 = slot(); float* s_c = slot();
    float* x0 = ar.f((size_t)N * d.hifi_dim);
    KN_RUN(knnsvc_absmax(cf, N, d.hubert_dim, d.hubert_dim, s_c, st));
    {
        knnsvc_conv_desc c = conv_desc(cf, d.lin, x0, N, d.hubert_dim, d.hifi_dim, 1, N, dy);
        c.bias = d.lin_b; c.x_absmax = s_c; c.out_absmax = s_x0;
        KN_RUN(knnsvc_conv_gemm(&c, st));
        c = conv_desc(x0, d.pre, cat_pre, N, d.hifi_dim, uic, 7, N, dy);
        c.pad = 3; c.bias = d.pre_b; c.ldo = pre_ld; c.x_absmax = s_x0; c.out_absmax = cat_pre_slot;
        KN_RUN(knnsvc_conv_gemm(&c, st));
    }
    // ---- excitation + sin_prenet -> res[0]
    {
        const View v0 = res_view(0);
        double* ph = (double*)ar.f((size_t)N * 2);
        KN_RUN(knnsvc_additive_synth(f0, d.kind == 0 ? harm : nullptr, N, d.kind == 0 ? d.n_harm_in : 0, hop, d.sample_rate, d.kind == 0 ? 0 : 1,
                                     d.prenet_w, d.prenet_b, d.side[0], v0.p, v0.ld, nullptr, ph, n_dyn, st));
        KN_RUN(knnsvc_absmax(v0.p, L, v0.ch, L > 1 ? v0.ld : v0.ch, v0.s, st));
    }
    // ---- side (down) path
    for (int i = 0; i < n_up; ++i) {
        const View src = res_view(i), dst = res_view(i + 1);
        const knnsvc_gen_stage& sg = g.st[i];
        const int64_t t_in = lens[i], t_mid = t_in / sg.down_u + 1;        // the conv yields one more row than the crop keeps; the k = 3
        float* mid = ar.f((size_t)t_mid * dst.ch);                          // resblock conv still reads it (ddsp_models.py:189-194)
        float* s_mid = slot();
        knnsvc_conv_desc c = conv_desc(src.p, sg.down, mid, t_in, src.ch, dst.ch, sg.down_k, t_mid, dy);
        c.stride = sg.down_u; c.pad = sg.down_k / 2; c.ldx = src.ld; c.bias = sg.down_b; c.x_absmax = src.s; c.out_absmax = s_mid;
        KN_RUN(knnsvc_conv_gemm(&c, st));
        c = conv_desc(mid, sg.rbd, dst.p, t_mid, dst.ch, dst.ch, 3, lens[i + 1], dy);
        c.pad = 1; c.bias = sg.rbd_b; c.a_slope = LRELU; c.resid = mid; c.ldr = dst.ch; c.ldo = dst.ld; c.x_absmax = s_mid; c.out_absmax = dst.s;
        KN_RUN(knnsvc_conv_gemm(&c, st));
    }
    // ---- main path
    float* x = ar.f((size_t)N * uic);
    float* s_x = slot();
    {
        knnsvc_conv_desc c = conv_desc(cat_pre, d.cpre, x, N, pre_ld, uic, 3, N, dy);
        c.pad = 1; c.bias = d.cpre_b; c.x_absmax = cat_pre_slot; c.out_absmax = s_x;
        KN_RUN(knnsvc_conv_gemm(&c, st));
    }
    int64_t t_cur = N;
    int x_ch = uic;
    // the stage outputs alternate between two buffers sized for the largest stage; everything else of a stage is released at its end
    size_t xs_floats = 0;
    { int64_t t = N; for (int i = 0; i < n_up; ++i) { t *= g.st[i].u; const size_t fl = (size_t)t * g.st[i].cout; xs_floats = fl > xs_floats ? fl : xs_floats; } }
    float* xs_buf[2] = {ar.f(xs_floats), ar.f(xs_floats)};
    const size_t stage_mark = ar.off;
    size_t high = ar.off;
    for (int i = 0; i < n_up; ++i) {
        ar.off = stage_mark;
        const knnsvc_gen_stage& sg = g.st[i];
        const int u = sg.u, k = sg.k, cout = sg.cout, cin = sg.cin, R = k / u;
        const int64_t t_out = t_cur * u;
        KN_REQUIRE(cin == x_ch && t_out == lens[n_up - 1 - i], "generator_forward: stage shapes do not chain");
        {
            knnsvc_conv_desc c = base_desc();
            c.x = x; c.ldx = cin; c.t_in = (int32_t)t_cur; c.cin = cin; c.taps = R; c.stride = 1; c.dil = -1; c.pad = 0; c.a_slope = LRELU;
            set_w(c, sg.up); c.n = u * cout; c.bias = sg.up_b; c.bias_period = cout; c.out = cat[i]; c.ldo = cat_ld[i]; c.m = (int32_t)(t_cur + R - 1);
            c.convt_u = u; c.convt_cout = cout; c.convt_pad = (k - u) / 2; c.t_out = (int32_t)t_out; c.x_absmax = s_x; c.out_absmax = cat_slot[i];
            set_dyn(c, dy);
            KN_RUN(knnsvc_conv_gemm(&c, st));
        }
        float* xc = ar.f((size_t)t_out * cout);
        float* s_xc = slot();
        {
            knnsvc_conv_desc c = conv_desc(cat[i], sg.ccv, xc, t_out, cat_ld[i], cout, 3, t_out, dy);
            c.pad = 1; c.x_absmax = cat_slot[i]; c.out_absmax = s_xc;
            KN_RUN(knnsvc_conv_gemm(&c, st));
        }
        float* xs = xs_buf[i & 1];
        float* s_xs = slot();
        float* outs[3]; float* tmp[3][3];
        for (int j = 0; j < 3; ++j) outs[j] = ar.f((size_t)t_out * cout);
        for (int j = 0; j < 3; ++j) for (int b = 0; b < 3; ++b) tmp[j][b] = ar.f((size_t)t_out * cout);        // t1, ra, rb of each branch
        int order[3] = {0, 1, 2};                                           // most taps first: workgroups are dispatched y-major
        for (int a = 0; a < 3; ++a) for (int b = a + 1; b < 3; ++b) if (sg.res_k[order[b]] > sg.res_k[order[a]]) { const int t = order[a]; order[a] = order[b]; order[b] = t; }
        const float* cur[3] = {xc, xc, xc}; float* s_cur[3] = {s_xc, s_xc, s_xc};
        for (int m = 0; m < 3; ++m) {
            knnsvc_pair_desc pairs[3]; knnsvc_conv_desc c1[3], c2[3];
            int np = 0, nc = 0;
            for (int oi = 0; oi < 3; ++oi) {
                const int j = order[oi];
                const knnsvc_gen_pair& cv = sg.res[j][m];
                const int kr = sg.res_k[j], dl = cv.dil;
                const bool last = m == 2;
                float* dst = last ? outs[j] : (cur[j] != tmp[j][1] ? tmp[j][1] : tmp[j][2]);
                float* s_dst = last ? nullptr : slot();
                if (pair_ok(cout, kr, dl)) {
                    knnsvc_pair_desc& p = pairs[np++];
                    memset(&p, 0, sizeof(p));
                    p.x = cur[j]; p.ldx = cout; p.t = (int32_t)t_out; p.channels = cout; p.taps = kr; p.dil = dl;
                    p.w1_f16x2 = cv.w1.w_f16x2; p.w1_scale = cv.w1.w_f16x2_scale; p.b1 = cv.b1;
                    p.w2_f16x2 = cv.w2.w_f16x2; p.w2_scale = cv.w2.w_f16x2_scale; p.b2 = cv.b2;
                    p.out = dst; p.ldo = cout; p.slope = LRELU; p.x_absmax = s_cur[j]; p.t1_bound_mul = cv.t1_bound_mul; p.t1_bound_add = cv.t1_bound_add;
                    p.out_absmax = s_dst;
                    if (n_dyn) { p.n_dyn = n_dyn; p.dyn_mul = (int32_t)(t_out / N); }
                } else {
                    // t1 = lrelu(convs1(lrelu(cur)) + b1) is not measured: its consumer bounds it by t1_bound applied to cur's slot
                    knnsvc_conv_desc& a = c1[nc];
                    a = conv_desc(cur[j], cv.w1, tmp[j][0], t_out, cout, cout, kr, t_out, dy);
                    a.dil = dl; a.pad = (kr * dl - dl) / 2; a.bias = cv.b1; a.a_slope = LRELU; a.act = KNNSVC_ACT_LRELU; a.act_slope = LRELU; a.x_absmax = s_cur[j];
                    knnsvc_conv_desc& b = c2[nc++];
                    b = conv_desc(tmp[j][0], cv.w2, dst, t_out, cout, cout, kr, t_out, dy);
                    b.pad = (kr - 1) / 2; b.bias = cv.b2; b.resid = cur[j]; b.ldr = cout; b.x_absmax = s_cur[j];
                    b.x_bound_mul = cv.t1_bound_mul; b.x_bound_add = cv.t1_bound_add; b.out_absmax = s_dst;
                }
                cur[j] = dst; s_cur[j] = s_dst;
            }
            if (np) KN_RUN(knnsvc_resblock_pair_multi(pairs, np, st));
            if (nc) { KN_RUN(knnsvc_conv_gemm_multi(c1, nc, st)); KN_RUN(knnsvc_conv_gemm_multi(c2, nc, st)); }
        }
        // (rb2 + (rb1 + rb0)) / 3: the association of the reference's running sum (ddsp_models.py:218-227); publishes the stage's slot
        KN_RUN(knnsvc_mean3(outs[0], outs[1], outs[2], t_out * cout, 3.0f, xs, s_xs, n_dyn, n_dyn ? t_out * cout / N : 0, st));
        x = xs; s_x = s_xs; t_cur = t_out; x_ch = cout;
        high = ar.off > high ? ar.off : high;
    }
    ar.off = high;
    {
        knnsvc_conv_desc c = conv_desc(x, d.post, out, t_cur, x_ch, 1, 7, t_cur, dy);
        c.pad = 3; c.a_slope = 0.01f; c.act = KNNSVC_ACT_TANH; c.x_absmax = s_x;
        KN_RUN(knnsvc_conv_gemm(&c, st));
    }
    if (n_slot > N_SLOTS) return knnsvc_fail(KNNSVC_EINVAL, "generator_forward: slot plan exceeded");
#undef KN_RUN
    return KNNSVC_OK;
}

}  // namespace

extern "C" int knnsvc_generator_create(const knnsvc_generator_desc* d, void** handle) {
    KN_REQUIRE(d && handle && d->stages, "generator_create: null pointer");
    KN_REQUIRE(d->n_up >= 1 && d->n_up <= 7 && d->hop > 0 && d->uic > 0 && (d->kind == 0 || d->kind == 1), "generator_create: bad configuration");
    Generator* g = new Generator();
    g->d = *d;
    g->st.assign(d->stages, d->stages + d->n_up);
    g->d.stages = nullptr;
    for (const auto& s : g->st)             // the fused pairs (C = 32 / 64) read the split images only
        for (int j = 0; j < 3; ++j)
            for (int m = 0; m < 3; ++m)
                if (pair_ok(s.cout, s.res_k[j], s.res[j][m].dil) && (!s.res[j][m].w1.w_f16x2 || !s.res[j][m].w2.w_f16x2)) {
                    delete g;
                    return knnsvc_fail(KNNSVC_EINVAL, "generator_create: the ResBlock weights of a 32- / 64-channel stage need their f16x2 split");
                }
    *handle = g;
    return KNNSVC_OK;
}

extern "C" int knnsvc_generator_free(void* handle) {
    delete (Generator*)handle;
    return KNNSVC_OK;
}

extern "C" size_t knnsvc_generator_workspace_bytes(const void* handle, int64_t frames) {
    if (!handle || frames <= 0) return 0;
    Arena ar(nullptr, 0);
    if (generate(*(const Generator*)handle, nullptr, nullptr, nullptr, frames, nullptr, nullptr, ar, nullptr)) return 0;
    return ar.off;
}

extern "C" int knnsvc_generator_forward(const void* handle, const float* c, const float* f0, const float* harm, int64_t frames,
                                        const int32_t* n_dyn, float* out, void* workspace, size_t workspace_bytes, void* stream) {
    KN_REQUIRE(handle && c && f0 && out && workspace, "generator_forward: null pointer");
    const Generator& g = *(const Generator*)handle;
    KN_REQUIRE(frames > 0 && (g.d.kind == 1 || harm), "generator_forward: empty input, or the 'mix' generator without harmonic amplitudes");
    KN_REQUIRE(((uintptr_t)workspace & 255) == 0, "generator_forward: workspace must be 256-byte aligned");
    const size_t need = knnsvc_generator_workspace_bytes(handle, frames);
    if (need == 0) return knnsvc_fail(KNNSVC_EINVAL, "generator_forward: bad configuration for %lld frames", (long long)frames);
    if (workspace_bytes < need) return knnsvc_fail(KNNSVC_EWORKSPACE, "generator_forward: workspace %zu < %zu bytes", workspace_bytes, need);
    Arena ar(workspace, workspace_bytes);
    return generate(g, c, f0, harm, frames, n_dyn, out, ar, stream);
}
