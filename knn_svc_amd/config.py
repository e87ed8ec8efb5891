"""Static configuration of the kNN-SVC inference path.

The reference keeps these values in two places that are absent offline:
the WavLM-Large checkpoint's ``cfg`` dict (reference ``ddsp_hubconf.py:113-119``,
defaults in ``wavlm/WavLM.py:162-214``) and ``hifigan/config_v1_wavlm.json``.
They are restated here as plain dicts so that the GPU box needs neither file.
"""
from __future__ import annotations

import copy

SAMPLE_RATE = 16000
HOP = 320                      # WavLM stride / vocoder hop (ddsp_prematch_dataset.py DOWNSAMPLE_FACTOR)
CHUNK_SAMPLES = 30 * SAMPLE_RATE   # get_full_wavlm_features: 30 s windows (ddsp_prematch_dataset.py:277)
MATCH_LAYER = 6                # SPEAKER_INFORMATION_WEIGHTS one-hot index (ddsp_matcher.py:88-89)
KNN_K = 32                     # hard-coded k (ddsp_prematch_dataset.py:1203)
KNN_USE = 4                    # first 4 neighbours used (ddsp_prematch_dataset.py:1246,1398)
N_HARM = 49                    # harmonics gathered per frame (ddsp_prematch_dataset.py:391)

# Public WavLM-Large values; keys follow WavLMConfig attribute names.
WAVLM_LARGE = dict(
    extractor_mode="layer_norm",
    encoder_layers=24,
    encoder_embed_dim=1024,
    encoder_ffn_embed_dim=4096,
    encoder_attention_heads=16,
    activation_fn="gelu",
    layer_norm_first=True,
    conv_feature_layers="[(512,10,5)] + [(512,3,2)] * 4 + [(512,2,2)] * 2",
    conv_bias=False,
    normalize=True,            # present in the checkpoint cfg but never applied (WavLM.py:323-375)
    conv_pos=128,
    conv_pos_groups=16,
    relative_position_embedding=True,
    num_buckets=320,
    max_distance=800,
    gru_rel_pos=True,
    dropout=0.0, attention_dropout=0.0, activation_dropout=0.0,
    encoder_layerdrop=0.0, dropout_input=0.0, dropout_features=0.0,
    feature_grad_mult=1.0, mask_prob=0.0,
)

# A structurally identical small model used by the parity tests (head_dim stays 64).
WAVLM_TINY = dict(WAVLM_LARGE,
    encoder_layers=3,
    encoder_embed_dim=128,
    encoder_ffn_embed_dim=256,
    encoder_attention_heads=2,
    conv_feature_layers="[(64,10,5)] + [(64,3,2)] * 4 + [(64,2,2)] * 2",
)

# hifigan/config_v1_wavlm.json (the keys the inference path reads).
HIFIGAN_V1 = dict(
    resblock="1",
    upsample_rates=[10, 8, 2, 2],
    upsample_kernel_sizes=[20, 16, 4, 4],
    upsample_initial_channel=512,
    resblock_kernel_sizes=[3, 7, 11],
    resblock_dilation_sizes=[[1, 3, 5], [1, 3, 5], [1, 3, 5]],
    n_harmonic=32,
    hubert_dim=1024,
    hifi_dim=512,
    hop_size=320,
    sampling_rate=16000,
)

HIFIGAN_TINY = dict(HIFIGAN_V1,
    upsample_initial_channel=64,
    n_harmonic=4,
    hubert_dim=128,
    hifi_dim=32,
)


def conv_layers(cfg) -> list:
    """[(dim, kernel, stride), ...] of the feature extractor."""
    return list(eval(cfg["conv_feature_layers"]))  # same literal grammar as WavLM.py:229


def clone(cfg: dict) -> dict:
    return copy.deepcopy(cfg)
