"""Seeded inputs shared by tests/gen_golden.py (which needs the reference) and the tests (which do not)."""
import torch

from knn_svc_amd import synthetic as S


def north_star_inputs():
    """Inputs of fixture G4c: 1500 query / 30 000 pool frames of temporally smooth 1024-d features and their f0 tracks — the
    BASELINE north-star point."""
    sm = lambda x: (x + torch.roll(x, 1, 0) + torch.roll(x, 2, 0)) / 3
    q = sm(S.clustered_features(1500, 1024, 1, n_centres=80))
    p = sm(S.clustered_features(30000, 1024, 2, n_centres=80))
    _, f0 = S.synth_clip(30000 * 320, 3); pf0 = torch.from_numpy(f0[:30000].copy())
    _, f0 = S.synth_clip(1500 * 320, 4); qf0 = torch.from_numpy(f0[:1500].copy() * 1.2)
    return q, p, qf0, pf0


def vocoder_full_inputs(n=60):
    """Inputs of fixture G7c: features, f0 track and harmonic amplitudes for the full-size generators."""
    import numpy as np
    from knn_svc_amd import config as C
    g = torch.Generator().manual_seed(71)
    c = torch.randn(1, n, C.HIFIGAN_V1["hubert_dim"], generator=g)
    _, f0 = S.synth_clip(n * 320, 72)
    f0 = torch.from_numpy(f0[:n].copy())[None, :, None]
    harm = torch.rand(1, n, 49, generator=g) * 0.02
    return c, f0, harm
