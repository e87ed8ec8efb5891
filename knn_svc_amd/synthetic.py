"""Seeded weights and synthetic audio for tests and benchmarks.

There is no network for the released checkpoints (``WavLM-Large.pt``, the
kNN-SVC generator ``.pt`` files), so parity tests and ``bench.py`` use
random-init weights of the real architecture.  The state-dict *names and
shapes* are those of the reference modules (``wavlm/WavLM.py``,
``hifigan/ddsp_models.py``, ``hifigan/ddsp_models_f0.py``) so that a real
checkpoint loads into the same packers; ``tests/gen_golden.py`` asserts that
against the imported reference.

Init is variance preserving (std = fan_in**-0.5) rather than the reference's
training init so that activations stay O(1) through every layer and the
numerics tests are sensitive.
"""
from __future__ import annotations

import math

import numpy as np
import torch

from . import config as C


# ---------------------------------------------------------------------------
# parameter specs
# ---------------------------------------------------------------------------
def wavlm_param_spec(cfg: dict, n_layers: int | None = None) -> list:
    """[(name, shape, kind)] for the WavLM tensors the 6-layer path reads."""
    E = cfg["encoder_embed_dim"]
    Fd = cfg["encoder_ffn_embed_dim"]
    H = cfg["encoder_attention_heads"]
    L = cfg["encoder_layers"] if n_layers is None else n_layers
    spec = []
    cin = 1
    for i, (dim, k, _s) in enumerate(C.conv_layers(cfg)):
        spec.append((f"feature_extractor.conv_layers.{i}.0.weight", (dim, cin, k), "w"))
        spec.append((f"feature_extractor.conv_layers.{i}.2.1.weight", (dim,), "ln_w"))
        spec.append((f"feature_extractor.conv_layers.{i}.2.1.bias", (dim,), "ln_b"))
        cin = dim
    spec.append(("layer_norm.weight", (cin,), "ln_w"))
    spec.append(("layer_norm.bias", (cin,), "ln_b"))
    spec.append(("post_extract_proj.weight", (E, cin), "w"))
    spec.append(("post_extract_proj.bias", (E,), "b"))
    G = cfg["conv_pos_groups"]
    K = cfg["conv_pos"]
    spec.append(("encoder.pos_conv.0.bias", (E,), "b"))
    spec.append(("encoder.pos_conv.0.weight_g", (1, 1, K), "wn_g_dim2"))
    spec.append(("encoder.pos_conv.0.weight_v", (E, E // G, K), "w"))
    for l in range(L):
        p = f"encoder.layers.{l}."
        for nm in ("k_proj", "v_proj", "q_proj", "out_proj"):
            spec.append((p + f"self_attn.{nm}.weight", (E, E), "w"))
            spec.append((p + f"self_attn.{nm}.bias", (E,), "b"))
        spec.append((p + "self_attn.grep_linear.weight", (8, E // H), "w"))
        spec.append((p + "self_attn.grep_linear.bias", (8,), "b"))
        spec.append((p + "self_attn.grep_a", (1, H, 1, 1), "ones"))
        if l == 0:
            spec.append((p + "self_attn.relative_attention_bias.weight", (cfg["num_buckets"], H), "emb"))
        spec.append((p + "self_attn_layer_norm.weight", (E,), "ln_w"))
        spec.append((p + "self_attn_layer_norm.bias", (E,), "ln_b"))
        spec.append((p + "fc1.weight", (Fd, E), "w"))
        spec.append((p + "fc1.bias", (Fd,), "b"))
        spec.append((p + "fc2.weight", (E, Fd), "w"))
        spec.append((p + "fc2.bias", (E,), "b"))
        spec.append((p + "final_layer_norm.weight", (E,), "ln_w"))
        spec.append((p + "final_layer_norm.bias", (E,), "ln_b"))
    return spec


def _wn(spec, name, shape):
    """weight-normed conv (dim=0): weight_g [shape0,1,1], weight_v shape, bias."""
    spec.append((name + ".weight_g", (shape[0], 1, 1), "wn_g_dim0"))
    spec.append((name + ".weight_v", shape, "w"))


def generator_param_spec(h: dict, kind: str) -> list:
    """[(name, shape, kind)] of ``SynthesizerTrn`` — kind 'mix' (ddsp_models.py) or 'f0' (ddsp_models_f0.py)."""
    assert kind in ("mix", "f0")
    nh = h["n_harmonic"]
    uic = h["upsample_initial_channel"]
    rates, ksz = h["upsample_rates"], h["upsample_kernel_sizes"]
    n_up = len(rates)
    spec = []
    spec.append(("dec.lin_pre.weight", (h["hifi_dim"], h["hubert_dim"]), "w"))
    spec.append(("dec.lin_pre.bias", (h["hifi_dim"],), "b"))
    spec.append(("dec.conv_pre.weight", (uic, h["hifi_dim"], 7), "w"))
    spec.append(("dec.conv_pre.bias", (uic,), "b"))
    for i in range(n_up):
        j = n_up - 1 - i
        if kind == "mix":
            cin, cout = nh * 2 ** i, nh * 2 ** (i + 1)
        else:
            cin = cout = nh + 2
        _wn(spec, f"dec.downs.{i}", (cout, cin, ksz[j]))
        spec.append((f"dec.downs.{i}.bias", (cout,), "b"))
    for i in range(n_up):
        ch = nh * 2 ** (i + 1) if kind == "mix" else nh + 2
        _wn(spec, f"dec.resblocks_downs.{i}.convs.0", (ch, ch, 3))
        spec.append((f"dec.resblocks_downs.{i}.convs.0.bias", (ch,), "b"))
    side = uic if kind == "mix" else nh + 2
    spec.append(("dec.concat_pre.weight", (uic, uic + side, 3), "w"))
    spec.append(("dec.concat_pre.bias", (uic,), "b"))
    for i in range(n_up):
        ch = uic // 2 ** (i + 1)
        side = ch if kind == "mix" else nh + 2
        spec.append((f"dec.concat_conv.{i}.weight", (ch, ch + side, 3), "w"))
    for i in range(n_up):
        cin, cout = uic // 2 ** i, uic // 2 ** (i + 1)
        _wn(spec, f"dec.ups.{i}", (cin, cout, ksz[i]))     # ConvTranspose1d weight is [Cin, Cout, k]
        spec.append((f"dec.ups.{i}.bias", (cout,), "b"))
    nk = len(h["resblock_kernel_sizes"])
    for i in range(n_up):
        ch = uic // 2 ** (i + 1)
        for j, k in enumerate(h["resblock_kernel_sizes"]):
            for grp in ("convs1", "convs2"):
                for m in range(3):
                    nm = f"dec.resblocks.{i * nk + j}.{grp}.{m}"
                    _wn(spec, nm, (ch, ch, k))
                    spec.append((nm + ".bias", (ch,), "b"))
    spec.append(("dec.conv_post.weight", (1, uic // 2 ** n_up, 7), "w"))
    pc = nh if kind == "mix" else nh + 2
    spec.append(("sin_prenet.weight", (pc, 1, 3), "w"))
    spec.append(("sin_prenet.bias", (pc,), "b"))
    return spec


def seeded_state(spec: list, seed: int) -> dict:
    """Deterministic CPU fp32 state dict for a spec (torch CPU generator: same bits on every box)."""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    sd = {}
    for name, shape, kind in spec:
        if kind == "w":
            fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else shape[0]
            if ".ups." in name and name.endswith("weight_v"):
                # transposed conv: each output sample sees Cin * k / stride taps; keep it O(1)
                fan_in = shape[0] * 2
            t = torch.randn(shape, generator=g) * (fan_in ** -0.5)
        elif kind == "b":
            t = torch.randn(shape, generator=g) * 0.02
        elif kind == "ln_w":
            t = 1.0 + 0.1 * torch.randn(shape, generator=g)
        elif kind == "ln_b":
            t = 0.1 * torch.randn(shape, generator=g)
        elif kind == "emb":
            t = torch.randn(shape, generator=g) * 0.5
        elif kind == "ones":
            t = 1.0 + 0.1 * torch.randn(shape, generator=g)
        elif kind in ("wn_g_dim0", "wn_g_dim2"):
            t = None   # filled below from the matching weight_v
        else:
            raise ValueError(kind)
        sd[name] = t
    for name, shape, kind in spec:
        if kind == "wn_g_dim0":
            v = sd[name[:-1] + "v"]
            sd[name] = (v.flatten(1).norm(dim=1).reshape(shape)
                        * (0.8 + 0.4 * torch.rand(shape, generator=g)))
        elif kind == "wn_g_dim2":
            v = sd[name[:-1] + "v"]
            sd[name] = (v.permute(2, 0, 1).flatten(1).norm(dim=1).reshape(shape)
                        * (0.8 + 0.4 * torch.rand(shape, generator=g)))
    return {k: v.float().contiguous() for k, v in sd.items()}


def state_checksum(sd: dict) -> float:
    """Order-independent fingerprint used by fixtures to detect RNG drift."""
    tot = 0.0
    for k in sorted(sd):
        tot += float(sd[k].double().abs().sum())
    return tot


# ---------------------------------------------------------------------------
# synthetic audio: gliding harmonic tone + unvoiced gaps + noise
# ---------------------------------------------------------------------------
def synth_clip(n_samples: int, seed: int, sr: int = C.SAMPLE_RATE, hop: int = C.HOP):
    """Returns (wav float32 [n_samples], f0 float32 [n_samples//hop + 1]) with 0 = unvoiced."""
    rng = np.random.default_rng(seed)
    n_frames = n_samples // hop + 1
    # slow random-walk pitch in semitones around a per-clip centre, 110..440 Hz
    centre = rng.uniform(math.log2(150), math.log2(330))
    walk = np.cumsum(rng.normal(0, 0.01, n_frames))
    walk -= np.linspace(0, walk[-1], n_frames)
    vib = 0.02 * np.sin(2 * np.pi * 5.5 * np.arange(n_frames) * hop / sr + rng.uniform(0, 6.28))
    f0 = 2.0 ** np.clip(centre + walk + vib, math.log2(110), math.log2(440))
    # ~10 % unvoiced, in gaps of 0.2-0.5 s
    voiced = np.ones(n_frames, bool)
    target_unv = int(0.1 * n_frames)
    tries = 0
    while (~voiced).sum() < target_unv and tries < 1000:
        glen = int(rng.uniform(0.2, 0.5) * sr / hop)
        st = int(rng.integers(0, max(1, n_frames - glen)))
        voiced[st:st + glen] = False
        tries += 1
    f0 = np.where(voiced, f0, 0.0).astype(np.float32)
    f0_s = np.repeat(f0, hop)[:n_samples].astype(np.float64)
    phase = 2 * np.pi * np.cumsum(f0_s / sr)
    env = rng.uniform(0.2, 1.0, 8) / np.arange(1, 9)
    wav = np.zeros(n_samples)
    for k in range(8):
        wav += env[k] * np.sin((k + 1) * phase) * ((k + 1) * f0_s < sr / 2)
    wav *= (f0_s > 0)
    wav = 0.3 * wav / max(1e-6, np.abs(wav).max())
    wav += 10 ** (-40 / 20) * rng.standard_normal(n_samples)
    # unvoiced gaps carry breath-like noise so frames are not degenerate
    wav += (f0_s == 0) * 0.03 * rng.standard_normal(n_samples)
    return wav.astype(np.float32), f0


def clustered_features(n: int, dim: int, seed: int, n_centres: int = 200, sigma: float = 0.3,
                       centre_seed: int = 1234) -> torch.Tensor:
    """kNN micro-benchmark features: shared mean offset + Gaussian clusters."""
    gc = torch.Generator().manual_seed(centre_seed)
    centres = torch.randn(n_centres, dim, generator=gc)
    offset = 0.5 * torch.randn(1, dim, generator=gc)
    g = torch.Generator().manual_seed(seed)
    which = torch.randint(0, n_centres, (n,), generator=g)
    return (centres[which] + offset + sigma * torch.randn(n, dim, generator=g)).float().contiguous()
