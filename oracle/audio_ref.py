"""Oracle: torchaudio.functional.resample (sinc_interp_hann defaults), restated from the library's published algorithm
(the reference calls it at ddsp_prematch_dataset.py:338-341 for non-16 kHz input).  torchaudio is absent offline, so this
boundary is PARITY UNPINNED against the library itself; the GPU implementation (knn_svc_amd.features.resample) is checked
against this restatement.  Test infrastructure only."""
from __future__ import annotations

import numpy as np


def resample(x: np.ndarray, orig_freq: int, new_freq: int, lowpass_filter_width: int = 6, rolloff: float = 0.99):
    """torchaudio.functional.resample(x, sr, 16000) with its defaults (sinc_interp_hann), restated from
    the library's published algorithm (ddsp_prematch_dataset.py:338-341 calls it).  torchaudio is absent
    offline, so this boundary is PARITY UNPINNED; it only runs for non-16 kHz input.  [C, L] -> [C, L']."""
    import math

    import torch
    import torch.nn.functional as F
    if orig_freq == new_freq:
        return x
    g = math.gcd(int(orig_freq), int(new_freq))
    orig, new = int(orig_freq) // g, int(new_freq) // g
    base = min(orig, new) * rolloff
    width = math.ceil(lowpass_filter_width * orig / base)
    idx = torch.arange(-width, width + orig, dtype=torch.float64)[None, None] / orig
    t = torch.arange(0, -new, -1, dtype=torch.float64)[:, None, None] / new + idx
    t = (t * base).clamp_(-lowpass_filter_width, lowpass_filter_width)
    window = torch.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    t = t * math.pi
    kern = torch.where(t == 0, torch.ones_like(t), t.sin() / t) * window * (base / orig)
    kern = kern.to(torch.float32)
    w = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32))
    n = w.shape[-1]
    w = F.pad(w, (width, width + orig))
    y = F.conv1d(w[:, None], kern, stride=orig).transpose(1, 2).reshape(w.shape[0], -1)
    return y[:, : math.ceil(new * n / orig)].numpy()
