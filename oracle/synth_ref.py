"""Oracle: pool side-features (STFT magnitude, harmonic amplitudes) and the additive
synthesiser (reference ddsp_prematch_dataset.py:131-208, 326, 361, 391-404).
Test infrastructure only."""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


def stft_mag(wav_1d: torch.Tensor, n_fft: int = 400, hop: int = 320) -> torch.Tensor:
    """torchaudio.transforms.Spectrogram(n_fft=400, hop_length=320, center=True, power=1)
    restated from its documented defaults: periodic Hann window of n_fft, reflect padding
    of n_fft//2, onesided, no normalisation; then ``.T[:, :-1]`` drops the Nyquist bin
    (ddsp_prematch_dataset.py:326, 361).  [L] -> [T, n_fft//2].
    torchaudio itself is absent offline: PARITY UNPINNED at this boundary."""
    win = torch.hann_window(n_fft, periodic=True, dtype=wav_1d.dtype)
    s = torch.stft(wav_1d, n_fft, hop_length=hop, win_length=n_fft, window=win, center=True,
                   pad_mode="reflect", normalized=False, onesided=True, return_complex=True)
    return s.abs().T[:, :-1].contiguous()


def harmonic_amps(spec: torch.Tensor, f0: torch.Tensor, n_harm: int = 49) -> torch.Tensor:
    """[T,200] magnitude + [T] f0 -> [T,49] amplitudes (ddsp_prematch_dataset.py:391-404):
    x8 linear bin interpolation, gather at round(clamp(f0*k*2*1600/16000, max=1600)) on the
    spectrum padded with one zero bin; unvoiced frames: h1 = max bin, h2.. = 0; x0.0108."""
    k = torch.arange(1, n_harm + 1, device=spec.device)[None, :]
    harm = f0[:, None].to(spec.device) * k
    interp = F.interpolate(spec[None, :], scale_factor=8, mode="linear").squeeze(0)
    nb = interp.shape[-1]
    gi = torch.round(torch.clamp(harm * 2 * nb / 16000, max=nb)).to(int)
    out = torch.gather(F.pad(interp, (0, 1)), dim=-1, index=gi)
    unv = f0 == 0
    out[:, 1:][unv] = 0
    out[:, 0][unv] = torch.max(spec, dim=1)[0][unv]
    return 0.0108 * out


def additive_synth(f0: torch.Tensor, amp: torch.Tensor, sr: int = 16000, hop: int = 320) -> torch.Tensor:
    """get_bulk_dsp_choral (ddsp_prematch_dataset.py:165-208).  f0 [B,N,1], amp [B,N,H] -> [B,N*hop,1].
    f0 nearest x hop; amp bicubic x hop (F.interpolate on [B,H,1,N] -> (1, N*hop));
    phase = fp64 inclusive cumsum(f0/sr), wrapped 2*pi*(p - round(p)) -> f32; harmonics
    phase*k in f32; Nyquist mask (f0*k < sr/2) + 1e-7; sum_k sin(.)*amp*mask."""
    f0u = F.interpolate(f0.transpose(1, 2), size=f0.shape[1] * hop).transpose(1, 2)
    a = amp.transpose(1, 2)
    au = F.interpolate(a[:, :, None], size=(1, a.shape[-1] * hop), mode="bicubic").squeeze(2).transpose(1, 2)
    H = amp.shape[-1]
    ph = torch.cumsum(f0u.double() / sr, dim=1)
    ph = (2 * math.pi * (ph - torch.round(ph))).float()
    k = torch.arange(1, H + 1, device=ph.device)
    phases = ph * k
    mask = ((f0u * k.to(f0u)[None, None, :]) < sr / 2).float() + 1e-7
    return (torch.sin(phases) * (au * mask)).sum(-1, keepdim=True)


def sine_excitation(f0: torch.Tensor, sr: int = 16000, hop: int = 320) -> torch.Tensor:
    """f0-only variant (hifigan/ddsp_models_f0.py:334-356): sin of the wrapped fp64 phase.
    f0 [B,N,1] -> [B,1,N*hop]."""
    f0u = F.interpolate(f0.permute(0, 2, 1), size=f0.shape[1] * hop).permute(0, 2, 1)
    om = torch.cumsum(f0u.double() / sr, dim=1)
    om = (2 * math.pi * (om - torch.round(om))).float()
    return torch.sin(om).transpose(1, 2)
