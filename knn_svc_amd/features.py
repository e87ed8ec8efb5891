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


def stft_harm_batch(wavs, f0s, Ts, n_fft: int = 400, hop: int = 320, n_harm: int = 49):
    """STFT magnitudes + harmonic amplitudes of MANY signals in four launches (pad, DFT GEMM, magnitude + harmonics) instead of
    four per signal: wavs = list of [L_i] device tensors, f0s = list of [>= T_i] device tensors, Ts = frames kept per signal
    (T_i <= 1 + L_i // hop).  -> list of (spec [T_i, n_fft/2], harm [T_i, n_harm]).  Row for row the same arithmetic as
    ``stft_mag`` + ``ops.harmonic_amps`` (shorter signals ride along in a batch laid out for the longest one)."""
    if not wavs:
        return []
    dev = wavs[0].device
    basis = _dft_basis(n_fft, dev)
    kp = basis.shape[1]
    bins = n_fft // 2
    out = [None] * len(wavs)
    order = sorted(range(len(wavs)), key=lambda i: -wavs[i].numel())          # similar lengths share a batch
    i0 = 0
    while i0 < len(order):
        Lmax = wavs[order[i0]].numel()
        Tmax = 1 + Lmax // hop
        stride = ((Tmax - 1) * hop + kp + 3) // 4 * 4
        # batch: as many as fit ~256 MB of padded signal, and no item shorter than half the longest
        nb = 1
        while (i0 + nb < len(order) and (nb + 1) * stride <= (1 << 26) and nb < 4096 and
               wavs[order[i0 + nb]].numel() * 2 >= Lmax):
            nb += 1
        ids = order[i0:i0 + nb]
        i0 += nb
        lens = [wavs[i].numel() for i in ids]
        assert min(lens) > n_fft // 2
        offs = torch.tensor(np.concatenate([[0], np.cumsum(lens)]), dtype=torch.int64).to(dev, non_blocking=True)
        flat = torch.cat([wavs[i].contiguous() for i in ids]) if nb > 1 else wavs[ids[0]].contiguous()
        xp = ops.reflect_pad_batch(flat, offs, n_fft // 2, stride)
        reim = torch.empty(nb * Tmax, n_fft, device=dev, dtype=torch.float32)
        ops.conv_gemm(xp, basis, reim, m=Tmax, n=n_fft, cin=kp, taps=1, stride=1, pad=0, t_in=Tmax, ldx=hop, a_scale=X_SCALE,
                      batches=nb, x_bstride=stride, o_bstride=Tmax * n_fft)
        f0b = torch.zeros(nb * Tmax, device=dev, dtype=torch.float32)
        if all(Ts[i] == Tmax for i in ids):
            f0b = torch.cat([f0s[i][:Tmax] for i in ids])
        else:                                         # ragged: one gather-free scatter of the concatenated tracks
            pos = np.concatenate([r * Tmax + np.arange(Ts[i]) for r, i in enumerate(ids)])
            f0b[torch.from_numpy(pos).to(dev, non_blocking=True)] = torch.cat([f0s[i][:Ts[i]] for i in ids])
        spec, harm = ops.spec_harm(reim, bins, f0b.view(-1), n_harm)
        spec, harm = spec.view(nb, Tmax, bins), harm.view(nb, Tmax, n_harm)
        for r, i in enumerate(ids):
            assert Ts[i] <= 1 + lens[r] // hop
            out[i] = (spec[r, :Ts[i]].contiguous(), harm[r, :Ts[i]].contiguous())
    return out


_RESAMPLE_CACHE = {}


def _resample_kernel(orig: int, new: int, lowpass_filter_width: int, rolloff: float, device):
    """torchaudio's sinc_interp_hann filter bank (``_get_sinc_resample_kernel``): [new, 2*width + orig] in fp64 -> fp32,
    K zero-padded to a multiple of 32 and pre-split for the matrix-core GEMM.  Weight preparation, done once per rate."""
    key = (orig, new, lowpass_filter_width, rolloff, str(device), ops.gemm_mode())
    if key not in _RESAMPLE_CACHE:
        base = min(orig, new) * rolloff
        width = math.ceil(lowpass_filter_width * orig / base)
        idx = np.arange(-width, width + orig, dtype=np.float64)[None, :] / orig
        t = (np.arange(0, -new, -1, dtype=np.float64)[:, None] / new + idx) * base
        t = np.clip(t, -lowpass_filter_width, lowpass_filter_width)
        window = np.cos(t * math.pi / lowpass_filter_width / 2) ** 2
        t = t * math.pi
        with np.errstate(invalid="ignore", divide="ignore"):
            kern = np.where(t == 0, 1.0, np.sin(t) / t) * window * (base / orig)
        k = kern.shape[1]
        kp = -(-k // K_PAD) * K_PAD
        padded = np.zeros((new, kp), np.float32)
        padded[:, :k] = kern.astype(np.float32)
        _RESAMPLE_CACHE[key] = (ops.attach_split(torch.from_numpy(padded).to(device).contiguous()), width, k)
    return _RESAMPLE_CACHE[key]


def resample(wav_1d: torch.Tensor, orig_freq: int, new_freq: int, lowpass_filter_width: int = 6,
             rolloff: float = 0.99) -> torch.Tensor:
    """torchaudio.functional.resample with its defaults (the call at ddsp_prematch_dataset.py:338-341) on the GPU: the
    polyphase FIR is a strided convolution, i.e. one GEMM over the framed view of the zero-padded signal (row m =
    xp[m*orig .. m*orig + K), as in stft_mag) against the [new, K] filter bank; the channel-last output [frames, new] IS the
    interleaved output stream.  torchaudio is absent offline: PARITY UNPINNED against the library itself, checked against
    the restatement in oracle/audio_ref.py."""
    if orig_freq == new_freq:
        return wav_1d
    g = math.gcd(int(orig_freq), int(new_freq))
    orig, new = int(orig_freq) // g, int(new_freq) // g
    kern, width, k = _resample_kernel(orig, new, lowpass_filter_width, rolloff, wav_1d.device)
    kp = kern.shape[1]
    n = wav_1d.numel()
    frames = n // orig + 1                                   # conv1d(pad(x, (width, width + orig)), stride=orig)
    xp = torch.zeros((frames - 1) * orig + kp, device=wav_1d.device, dtype=torch.float32)
    xp[width:width + n] = wav_1d
    out = torch.empty(frames, new, device=wav_1d.device, dtype=torch.float32)
    ops.conv_gemm(xp, kern, out, m=frames, n=new, cin=kp, taps=1, stride=1, pad=0, t_in=frames, ldx=orig, a_scale=X_SCALE)
    return out.reshape(-1)[: math.ceil(new * n / orig)]
