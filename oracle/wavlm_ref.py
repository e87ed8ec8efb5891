"""Oracle: WavLM forward restricted to what the kNN-SVC path consumes.

Follows the reference's ``wavlm/WavLM.py`` / ``wavlm/modules.py`` (file:line in
each docstring).  Operates on a flat state dict with the reference's parameter
names.  Test infrastructure only (see ``oracle/__init__.py``).
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


def _fp32(x: torch.Tensor) -> torch.Tensor:
    """The reference's explicit ``.float()`` casts (Fp32LayerNorm, the fp32 GELU).  A float64 tensor passes through: the
    tests evaluate this same oracle in fp64 (state dict and input cast to double) as the exact-arithmetic yardstick."""
    return x if x.dtype == torch.float64 else x.float()


def conv_layers_of(cfg):
    return list(eval(cfg["conv_feature_layers"]))


def rel_bucket(rel: torch.Tensor, num_buckets: int, max_distance: int) -> torch.Tensor:
    """Bidirectional relative-position bucket (wavlm/modules.py:417-442).

    rel = key_pos - query_pos (int64).  Half the buckets per sign; inside a
    half, |rel| < max_exact is exact and the rest is log-spaced and truncated.
    """
    half = num_buckets // 2
    out = (rel > 0).to(torch.long) * half
    a = rel.abs()
    max_exact = half // 2
    large = max_exact + (
        torch.log(a.float() / max_exact) / math.log(max_distance / max_exact) * (half - max_exact)
    ).to(torch.long)
    large = torch.clamp(large, max=half - 1)
    return out + torch.where(a < max_exact, a, large)


def rel_bucket_table(T: int, num_buckets: int, max_distance: int) -> torch.Tensor:
    """LUT over rel = -(T-1)..(T-1) (index rel + T - 1) -> bucket id."""
    rel = torch.arange(-(T - 1), T, dtype=torch.long)
    return rel_bucket(rel, num_buckets, max_distance)


def position_bias(sd, cfg, T: int) -> torch.Tensor:
    """[H, T, T] table lookup (wavlm/modules.py:444-455)."""
    ctx = torch.arange(T)[:, None]
    mem = torch.arange(T)[None, :]
    b = rel_bucket(mem - ctx, cfg["num_buckets"], cfg["max_distance"])
    emb = sd["encoder.layers.0.self_attn.relative_attention_bias.weight"]
    return emb[b].permute(2, 0, 1).contiguous()


def feature_extractor(sd, cfg, wav: torch.Tensor) -> torch.Tensor:
    """[B, L] -> [B, C, T]: 7 x (Conv1d no bias -> LayerNorm over channels -> GELU)
    (wavlm/WavLM.py:409-419, 485-504; Fp32LayerNorm wavlm/modules.py:30-42)."""
    x = wav[:, None, :]
    for i, (dim, k, s) in enumerate(conv_layers_of(cfg)):
        p = f"feature_extractor.conv_layers.{i}."
        x = F.conv1d(x, sd[p + "0.weight"], None, stride=s)
        x = F.layer_norm(_fp32(x.transpose(1, 2)), (dim,), sd[p + "2.1.weight"], sd[p + "2.1.bias"], 1e-5)
        x = F.gelu(x.transpose(1, 2))
    return x


def pos_conv_weight(sd) -> torch.Tensor:
    """weight_norm(dim=2) fold (wavlm/WavLM.py:526): w = v * g / ||v||_(0,1)."""
    return torch._weight_norm(sd["encoder.pos_conv.0.weight_v"], sd["encoder.pos_conv.0.weight_g"], 2)


def encoder_front(sd, cfg, feats_bct: torch.Tensor) -> torch.Tensor:
    """conv features [B,C,T] -> transformer input [B,T,E]:
    LayerNorm(C) -> post_extract_proj (WavLM.py:341-348), then
    x += GELU(SamePad(pos_conv(x))) (WavLM.py:577-579, modules.py:72-83)."""
    x = feats_bct.transpose(1, 2)
    C = x.shape[-1]
    x = F.layer_norm(x, (C,), sd["layer_norm.weight"], sd["layer_norm.bias"], 1e-5)
    x = F.linear(x, sd["post_extract_proj.weight"], sd["post_extract_proj.bias"])
    K = cfg["conv_pos"]
    pc = F.conv1d(x.transpose(1, 2), pos_conv_weight(sd), sd["encoder.pos_conv.0.bias"],
                  padding=K // 2, groups=cfg["conv_pos_groups"])
    if K % 2 == 0:
        pc = pc[:, :, :-1]
    return x + F.gelu(pc).transpose(1, 2)


def gate(sd, cfg, l: int, xn_tbe: torch.Tensor) -> torch.Tensor:
    """Per (batch, head, query) multiplier of the position bias
    (wavlm/modules.py:523-533): computed from the layer-normed layer input."""
    T, B, E = xn_tbe.shape
    H = cfg["encoder_attention_heads"]
    p = f"encoder.layers.{l}.self_attn."
    ql = xn_tbe.transpose(0, 1).reshape(B, T, H, E // H).permute(0, 2, 1, 3)
    g = torch.sigmoid(F.linear(ql, sd[p + "grep_linear.weight"], sd[p + "grep_linear.bias"])
                      .view(B, H, T, 2, 4).sum(-1))
    ga, gb = g.chunk(2, dim=-1)
    return ga * (gb * sd[p + "grep_a"] - 1.0) + 2.0        # [B,H,T,1]


def encoder_layer(sd, cfg, l: int, x_tbe: torch.Tensor, pbias_htt: torch.Tensor) -> torch.Tensor:
    """Pre-LN block (wavlm/WavLM.py:691-714): LN -> MHA(+gated bias) -> +res -> LN -> fc1 -> GELU -> fc2 -> +res.
    Attention follows F.multi_head_attention_forward's need_weights=False route
    (scaled_dot_product_attention with a float additive mask), wavlm/modules.py:540-563."""
    T, B, E = x_tbe.shape
    H = cfg["encoder_attention_heads"]
    d = E // H
    p = f"encoder.layers.{l}."
    xn = F.layer_norm(x_tbe, (E,), sd[p + "self_attn_layer_norm.weight"], sd[p + "self_attn_layer_norm.bias"], 1e-5)
    bias = gate(sd, cfg, l, xn) * pbias_htt[None]                      # [B,H,T,T]
    q = F.linear(xn, sd[p + "self_attn.q_proj.weight"], sd[p + "self_attn.q_proj.bias"])
    k = F.linear(xn, sd[p + "self_attn.k_proj.weight"], sd[p + "self_attn.k_proj.bias"])
    v = F.linear(xn, sd[p + "self_attn.v_proj.weight"], sd[p + "self_attn.v_proj.bias"])
    sh = lambda t: t.reshape(T, B * H, d).transpose(0, 1).reshape(B, H, T, d)
    o = F.scaled_dot_product_attention(sh(q), sh(k), sh(v), attn_mask=bias)
    o = o.permute(2, 0, 1, 3).reshape(T, B, E)
    o = F.linear(o, sd[p + "self_attn.out_proj.weight"], sd[p + "self_attn.out_proj.bias"])
    x = x_tbe + o
    xn = F.layer_norm(x, (E,), sd[p + "final_layer_norm.weight"], sd[p + "final_layer_norm.bias"], 1e-5)
    hmid = F.gelu(_fp32(F.linear(xn, sd[p + "fc1.weight"], sd[p + "fc1.bias"])))
    return x + F.linear(hmid, sd[p + "fc2.weight"], sd[p + "fc2.bias"])


@torch.inference_mode()
def extract_layer(sd, cfg, wav: torch.Tensor, n_layers: int, all_layers: bool = False):
    """[B, L] raw waveform (no normalisation, WavLM.py:323-375) -> residual stream after
    ``n_layers`` layers, [B, T, E], no final LN (WavLM.py:567-568, 602-604).
    all_layers=True returns the list layer_results[0..n_layers]."""
    x = encoder_front(sd, cfg, feature_extractor(sd, cfg, wav))
    T = x.shape[1]
    pb = position_bias(sd, cfg, T)
    x = x.transpose(0, 1)
    outs = [x.transpose(0, 1)]
    for l in range(n_layers):
        x = encoder_layer(sd, cfg, l, x, pb)
        outs.append(x.transpose(0, 1))
    return outs if all_layers else outs[-1]


def n_frames(n_samples: int, cfg) -> int:
    """Frame-count law of the conv stack (no padding): out = (in - k)//s + 1 per layer."""
    n = n_samples
    for (_d, k, s) in conv_layers_of(cfg):
        n = (n - k) // s + 1
    return n


def chunk_plan(n_samples: int, sr: int = 16000, hop: int = 320):
    """30 s windows; tails of <= 0.02*sr samples dropped; right pad hop - len % hop
    (a full hop when aligned) — ddsp_prematch_dataset.py:275-293.  Returns [(start, length, n_pad)]."""
    plan = []
    start = 0
    while start < n_samples:
        ln = min(30 * sr, n_samples - start)
        if ln <= 0.02 * sr:
            break
        plan.append((start, ln, hop - (ln % hop)))
        start += 30 * sr
    return plan


@torch.inference_mode()
def full_features(sd, cfg, wav_1d: torch.Tensor, n_layers: int) -> torch.Tensor:
    """get_full_wavlm_features + one-hot layer mix == layer ``n_layers`` output, [T_total, E]
    (ddsp_prematch_dataset.py:270-296, 349-350; SURVEY.md §3.2 early-exit probe)."""
    feats = []
    for (st, ln, npad) in chunk_plan(wav_1d.shape[-1]):
        ch = F.pad(wav_1d[st:st + ln], (0, npad))[None]
        feats.append(extract_layer(sd, cfg, ch, n_layers)[0])
    return torch.cat(feats, 0)
