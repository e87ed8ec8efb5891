"""Pool side-features on the GPU: STFT magnitude and harmonic amplitudes.

Mirrors the per-file body of the reference's ``get_complete_spk_pool``
(ddsp_prematch_dataset.py:326, 361, 391-404).  The 400-point STFT is a
[T,416] x [416,400] DFT product on the MFMA conv kernel (window folded into the
basis, K zero-padded to a multiple of 32), so nothing here touches rocFFT or the host.
"""
from __future__ import annotations

import math

import numpy as np
import torch

from . import ops

_DFT_CACHE = {}


K_PAD = 32          # the framed-signal GEMM runs on the f16x2 matrix-core kernel, whose K must be a multiple of 32
X_SCALE = 4096.0     # waveform samples are <= 1 in magnitude: lift them well inside the fp16 range (exact power of two)


def _dft_basis(n_fft: int, device) -> torch.Tensor:
    """[2*(n_fft/2), Kp]: rows 0..199 = w[n] cos(2 pi k n / N), rows 200..399 = -w[n] sin(.), fp64 -> fp32, columns
    n_fft..Kp-1 zero (Kp = n_fft rounded up to 32); pre-split for the emulated-fp32 GEMM (ops.attach_split)."""
    key = (n_fft, str(device), ops.gemm_mode())
    if key not in _DFT_CACHE:
        n = np.arange(n_fft)
        win = 0.5 - 0.5 * np.cos(2 * np.pi * n / n_fft)           # periodic Hann (torch.hann_window default)
        k = np.arange(n_fft // 2)[:, None]                          # Nyquist bin is dropped by the reference
        ang = 2 * np.pi * k * n[None, :] / n_fft
        basis = np.concatenate([np.cos(ang) * win, -np.sin(ang) * win], 0)
        kp = -(-n_fft // K_PAD) * K_PAD
        padded = np.zeros((basis.shape[0], kp), np.float32)
        padded[:, :n_fft] = basis.astype(np.float32)
        _DFT_CACHE[key] = ops.attach_split(torch.from_numpy(padded).to(device).contiguous())
    return _DFT_CACHE[key]


def stft_mag(wav_1d: torch.Tensor, n_fft: int = 400, hop: int = 320) -> torch.Tensor:
    """[L] -> [1 + L//hop, n_fft//2] magnitude (centre=True reflect padding, power=1, Nyquist dropped).

    One GEMM over the framed view of the padded signal: row t = xp[t*hop .. t*hop + Kp) (rows overlap: ldx = hop <
    cin = Kp, allowed for taps == 1), against the zero-padded windowed DFT basis."""
    L = wav_1d.numel()
    basis = _dft_basis(n_fft, wav_1d.device)
    kp = basis.shape[1]
    xp = ops.reflect_pad(wav_1d.contiguous(), n_fft // 2, extra=kp - n_fft)
    T = 1 + L // hop
    reim = torch.empty(T, n_fft, device=wav_1d.device, dtype=torch.float32)
    ops.conv_gemm(xp, basis, reim, m=T, n=n_fft, cin=kp, taps=1, stride=1, pad=0, t_in=T, ldx=hop, a_scale=X_SCALE)
    return ops.complex_mag(reim, n_fft // 2)
