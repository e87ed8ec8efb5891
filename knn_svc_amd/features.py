"""Pool side-features on the GPU: STFT magnitude and harmonic amplitudes.

Mirrors the per-file body of the reference's ``get_complete_spk_pool``
(ddsp_prematch_dataset.py:326, 361, 391-404).  The 400-point STFT is a
[T,400] x [400,400] DFT product on the MFMA conv kernel (window folded into the
basis), so nothing here touches rocFFT or the host.
"""
from __future__ import annotations

import math

import numpy as np
import torch

from . import ops

_DFT_CACHE = {}


def _dft_basis(n_fft: int, device) -> torch.Tensor:
    """[2*(n_fft/2), n_fft]: rows 0..199 = w[n] cos(2 pi k n / N), rows 200..399 = -w[n] sin(.), fp64 -> fp32."""
    key = (n_fft, str(device))
    if key not in _DFT_CACHE:
        n = np.arange(n_fft)
        win = 0.5 - 0.5 * np.cos(2 * np.pi * n / n_fft)           # periodic Hann (torch.hann_window default)
        k = np.arange(n_fft // 2)[:, None]                          # Nyquist bin is dropped by the reference
        ang = 2 * np.pi * k * n[None, :] / n_fft
        basis = np.concatenate([np.cos(ang) * win, -np.sin(ang) * win], 0)
        _DFT_CACHE[key] = torch.from_numpy(basis.astype(np.float32)).to(device).contiguous()
    return _DFT_CACHE[key]


def stft_mag(wav_1d: torch.Tensor, n_fft: int = 400, hop: int = 320) -> torch.Tensor:
    """[L] -> [1 + L//hop, n_fft//2] magnitude (centre=True reflect padding, power=1, Nyquist dropped)."""
    L = wav_1d.numel()
    xp = ops.reflect_pad(wav_1d.contiguous(), n_fft // 2)
    T = 1 + L // hop
    basis = _dft_basis(n_fft, wav_1d.device)
    reim = torch.empty(T, n_fft, device=wav_1d.device, dtype=torch.float32)
    ops.conv_gemm(xp, basis, reim, m=T, n=n_fft, cin=1, taps=n_fft, stride=hop, pad=0, t_in=xp.numel(), ldx=1)
    return ops.complex_mag(reim, n_fft // 2)
