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
    int next_slot = 0;
    if (run && hipMemsetAsync(slots, 0, (size_t)n_slots * SLOT_FLOATS * 4, (hipStream_t)st) != hipSuccess)
        return knnsvc_fail(KNNSVC_EHIP, "wavlm_encode: hipMemsetAsync failed");
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
    if (mixed) {
        if (run && hipMemcpyAsync(out, acc, (size_t)R * E * 4, hipMemcpyDeviceToDevice, (hipStream_t)st) != hipSuccess)
            return knnsvc_fail(KNNSVC_EHIP, "wavlm_encode: hipMemcpyAsync failed");
    } else if (d.n_layers == 0) {
        if (run && hipMemcpyAsync(out, cur, (size_t)R * E * 4, hipMemcpyDeviceToDevice, (hipStream_t)st) != hipSuccess)
            return knnsvc_fail(KNNSVC_EHIP, "wavlm_encode: hipMemcpyAsync failed");
    }
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
