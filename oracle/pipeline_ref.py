"""Oracle: the orchestration of the hot path on in-memory audio — pool building,
matching, weighting, vocoding, PCM scaling (reference ddsp_prematch_dataset.py:301-414,
1074-1459; ddsp_matcher.py:375-406, 937-1023; lib_ongaku_test.py:89-120).
File handling (load / resample / rglob) is outside the oracle.  Test infrastructure only."""
from __future__ import annotations

import numpy as np
import torch

from . import knn_ref, select_ref, smooth_ref, synth_ref, vocoder_ref, wavlm_ref


def utterance_features(sd_w, cfg, wav_1d: torch.Tensor, f0: torch.Tensor, n_layers: int = 6, synth_layer=None) -> dict:
    """One file of get_complete_spk_pool (ddsp_prematch_dataset.py:331-404).  ``synth_layer``: a synthesis weighting that is the
    one-hot on another layer than the matching weighting (:349-350: two feature sets per file) -> extra entry "feats_synth"."""
    feats = wavlm_ref.full_features(sd_w, cfg, wav_1d, n_layers)
    T = len(feats)
    assert len(wav_1d) >= 320 * T
    spec = synth_ref.stft_mag(wav_1d)
    assert spec.shape[0] >= T
    spec = spec[:T]
    assert abs(len(f0) - T) <= 1 and len(f0) >= T
    f0 = f0[:T].float()
    out = dict(feats=feats, spec=spec, f0=f0, harm=synth_ref.harmonic_amps(spec, f0))
    if synth_layer is not None and synth_layer != n_layers:
        out["feats_synth"] = wavlm_ref.full_features(sd_w, cfg, wav_1d, synth_layer)
        assert out["feats_synth"].shape == feats.shape
    return out


def build_pool(sd_w, cfg, wavs, f0s, duration_limit=None, n_layers: int = 6, synth_layer=None) -> dict:
    """Concatenated pool over utterances with the overshooting duration limit
    (ddsp_prematch_dataset.py:408-411, 1152-1168)."""
    parts, dur = [], 0.0
    for w, f in zip(wavs, f0s):
        u = utterance_features(sd_w, cfg, w, f, n_layers, synth_layer)
        parts.append(u)
        dur += len(u["spec"]) * 320 / 16000
        if duration_limit is not None and dur >= duration_limit:
            break
    return {k: torch.cat([p[k] for p in parts], 0) for k in parts[0]}


def match(query: dict, pool: dict, ckpt_type: str = "mix", post_opt: str = "no_post_opt",
          return_debug: bool = False):
    """The per-query body of match_at_inference_time (ddsp_prematch_dataset.py:1189-1450)."""
    q, P = query["feats"], pool["feats"]
    Ps = pool.get("feats_synth", P)      # synth_list (:1157, 1260, 1347): what is gathered and smoothed; the search runs on matching_list
    nn32, _ = knn_ref.knn_topk(q, P, k=32)
    f0s = select_ref.shift_query_f0(query["f0"], pool["f0"])
    cw, run_adam = select_ref.parse_post_opt(post_opt)
    idx = nn32[:, :4].clone()
    if cw != -1:
        idx = select_ref.concat_reselect(idx, q, P, concat_weight=cw)
    gathered = Ps[idx.reshape(-1)].reshape(idx.shape[0], 4, Ps.shape[-1])
    if run_adam:
        w = smooth_ref.smooth_weights(idx, Ps, 0.1)
    else:
        w = torch.softmax(torch.ones(idx.shape), dim=1)
    out_feats = torch.sum(gathered * w[..., None], dim=1).float()
    dbg = dict(nn32=nn32, idx_wavlm=idx, w_wavlm=w)
    ranked = select_ref.rerank_by_f0(f0s, pool["f0"], nn32)
    idx2 = ranked[:, :4].clone()
    if cw != -1:
        idx2 = select_ref.concat_reselect(idx2, q, P, f0s, pool["f0"], concat_weight=cw)
    harm_w = None
    if "wavlm_only" not in ckpt_type and "no_harm_no_amp" not in ckpt_type:
        hg = pool["harm"][idx2.reshape(-1)].reshape(idx2.shape[0], 4, pool["harm"].shape[-1])
        if run_adam:
            w2 = smooth_ref.smooth_weights(idx2, pool["harm"], 1000.0)
            harm_w = torch.sum(hg * w2[..., None], dim=1)
            dbg["w_harm"] = w2
        else:
            harm_w = torch.mean(hg, dim=1)
    dbg["idx_harm"] = idx2
    res = (out_feats, harm_w, f0s)
    return res + (dbg,) if return_debug else res


def convert(sd_w, cfg_w, sd_g, h, kind, src_wav, src_f0, pool_wavs, pool_f0s, ckpt_type="mix",
            post_opt="no_post_opt", duration_limit=None, n_layers: int = 6, synth_layer=None) -> torch.Tensor:
    """special_match minus file I/O (ddsp_matcher.py:937-995).  special_match does not forward
    post_opt for wavlm_only / no_harm_no_amp checkpoints (ddsp_matcher.py:970)."""
    query = utterance_features(sd_w, cfg_w, src_wav, src_f0, n_layers)
    pool = build_pool(sd_w, cfg_w, pool_wavs, pool_f0s, duration_limit, n_layers, synth_layer)
    f0only = "wavlm_only" in ckpt_type or "no_harm_no_amp" in ckpt_type
    out_feats, harm_w, f0s = match(query, pool, ckpt_type, "no_post_opt" if f0only else post_opt)
    y = vocoder_ref.synthesizer(sd_g, h, kind, out_feats[None], f0s[None, :, None],
                                None if f0only else harm_w[None])
    return y.squeeze()


def to_pcm32(wave: np.ndarray) -> np.ndarray:
    """save_audio scaling (lib_ongaku_test.py:102-112): /max only if max>1, *(2^31-1), truncate."""
    m = np.max(np.abs(wave))
    if m > 1:
        wave = wave / m
    return (wave * (2 ** 31 - 1)).astype(np.int32)
