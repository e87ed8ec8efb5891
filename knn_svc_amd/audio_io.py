"""Audio file plumbing for the host side: RIFF/WAVE reader and PCM_32 writer.

The reference goes through ``torchaudio.load`` (ddsp_prematch_dataset.py:332) and
``soundfile.write(..., subtype='PCM_32')`` (lib_ongaku_test.py:118-120).  Neither
library is in this image, so WAV is handled here with numpy and FLAC (what the
reference's prematch builder globs next to .wav: LibriSpeech) by the library's own
RFC 9639 decoder / encoder (csrc/flac.hip, host code: frame CRCs and the stream MD5 are
verified on every read).  .mp3 is delegated to soundfile when importable and refused otherwise — at LISTING
time (matching.list_audio), before anything is encoded.  Sample values follow torchaudio's convention: integer PCM is scaled
by 2**-(bits-1) to float32 in [-1, 1).
"""
from __future__ import annotations

import os
import struct

import numpy as np


def read_wav(path: str):
    """-> (float32 [channels, samples], sample_rate)."""
    with open(path, "rb") as f:
        data = f.read()
    if data[:4] != b"RIFF" or data[8:12] != b"WAVE":
        raise ValueError(f"{path}: not a RIFF/WAVE file")
    pos = 12
    fmt = None
    pcm = None
    while pos + 8 <= len(data):
        cid = data[pos:pos + 4]
        (sz,) = struct.unpack("<I", data[pos + 4:pos + 8])
        body = data[pos + 8:pos + 8 + sz]
        if cid == b"fmt ":
            tag, ch, sr, _br, _ba, bits = struct.unpack("<HHIIHH", body[:16])
            if tag == 0xFFFE and len(body) >= 26:          # WAVE_FORMAT_EXTENSIBLE
                tag = struct.unpack("<H", body[24:26])[0]
            fmt = (tag, ch, sr, bits)
        elif cid == b"data":
            pcm = body
        pos += 8 + sz + (sz & 1)
    if fmt is None or pcm is None:
        raise ValueError(f"{path}: missing fmt/data chunk")
    tag, ch, sr, bits = fmt
    if tag == 1:
        if bits == 16:
            x = np.frombuffer(pcm, "<i2").astype(np.float32) / 32768.0
        elif bits == 32:
            x = (np.frombuffer(pcm, "<i4").astype(np.float64) / 2147483648.0).astype(np.float32)
        elif bits == 24:
            b = np.frombuffer(pcm, np.uint8).reshape(-1, 3).astype(np.int32)
            v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
            v = np.where(v & 0x800000, v - 0x1000000, v)
            x = (v.astype(np.float64) / 8388608.0).astype(np.float32)
        elif bits == 8:
            x = (np.frombuffer(pcm, np.uint8).astype(np.float32) - 128.0) / 128.0
        else:
            raise ValueError(f"{path}: unsupported PCM width {bits}")
    elif tag == 3:
        x = np.frombuffer(pcm, "<f4" if bits == 32 else "<f8").astype(np.float32)
    else:
        raise ValueError(f"{path}: unsupported WAVE format tag {tag}")
    n = len(x) // ch
    return np.ascontiguousarray(x[:n * ch].reshape(n, ch).T), sr


def read_flac_pcm(path: str):
    """-> (int32 [channels, samples] holding `bits`-bit samples, sample_rate, bits).  Frame CRCs are checked by the decoder, the
    MD5 of the decoded PCM against STREAMINFO here (skipped when the file carries no MD5)."""
    import ctypes as C
    import hashlib
    from . import _lib
    from .ops import check
    lib = _lib.load()
    with open(path, "rb") as f:
        data = f.read()
    buf = (C.c_uint8 * len(data)).from_buffer_copy(data)
    sr, ch, bits, total = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int64()
    md5 = (C.c_uint8 * 16)()
    check(lib.knnsvc_flac_info(buf, len(data), C.byref(sr), C.byref(ch), C.byref(bits), C.byref(total), md5), f"flac_info({path})")
    if total.value <= 0:
        raise ValueError(f"{path}: FLAC stream without a sample count in STREAMINFO is not supported")
    out = np.empty((ch.value, total.value), np.int32)
    done = C.c_int64()
    check(lib.knnsvc_flac_decode(buf, len(data), out.ctypes.data_as(C.c_void_p), total.value, C.byref(done)), f"flac_decode({path})")
    want = bytes(md5)
    if any(want):
        nb = (bits.value + 7) // 8
        inter = np.ascontiguousarray(out.T)
        raw = inter.astype("<i4").view(np.uint8).reshape(-1, 4)[:, :nb].tobytes()
        if hashlib.md5(raw).digest() != want:
            raise ValueError(f"{path}: decoded PCM does not match the MD5 in STREAMINFO")
    return out, sr.value, bits.value


def read_flac(path: str):
    """-> (float32 [channels, samples], sample_rate), scaled like torchaudio.load."""
    pcm, sr, bits = read_flac_pcm(path)
    return (pcm.astype(np.float64) / float(1 << (bits - 1))).astype(np.float32), sr


def write_flac(path: str, pcm: np.ndarray, sr: int, bits: int = 24) -> None:
    """pcm: int32 [samples] or [channels, samples] of `bits`-bit samples -> FLAC (fixed predictors + Rice, MD5 set)."""
    import ctypes as C
    import hashlib
    from . import _lib
    from .ops import check
    lib = _lib.load()
    pcm = np.ascontiguousarray(np.atleast_2d(np.asarray(pcm, np.int32)))
    ch, n = pcm.shape
    nb = (bits + 7) // 8
    raw = np.ascontiguousarray(pcm.T).astype("<i4").view(np.uint8).reshape(-1, 4)[:, :nb].tobytes()
    md5 = (C.c_uint8 * 16).from_buffer_copy(hashlib.md5(raw).digest())
    cap = 5 * n * ch + 8192
    out = (C.c_uint8 * cap)()
    size = C.c_int64()
    check(lib.knnsvc_flac_encode(pcm.ctypes.data_as(C.c_void_p), ch, n, bits, int(sr), md5, out, cap, C.byref(size)), "flac_encode")
    with open(path, "wb") as f:
        f.write(bytes(out[:size.value]) if size.value < (1 << 20) else memoryview(out)[:size.value])


def can_decode(ext: str) -> bool:
    """Whether ``load_audio`` has a decoder for this extension HERE: .wav / .flac natively, anything else only through an
    importable soundfile (absent in this image — the reference reads .mp3 through torchaudio/ffmpeg)."""
    if ext.lower() in (".wav", ".flac"):
        return True
    try:
        import soundfile  # noqa: F401
        return True
    except ImportError:
        return False


def load_audio(path: str):
    """torchaudio.load stand-in: -> (float32 ndarray [channels, samples], sr)."""
    ext = os.path.splitext(path)[-1].lower()
    if ext == ".wav":
        return read_wav(path)
    if ext == ".flac":
        return read_flac(path)
    try:                                    # optional decoders; absent in this image
        import soundfile as sf
        x, sr = sf.read(path, dtype="float32", always_2d=True)
        return np.ascontiguousarray(x.T), sr
    except ImportError:
        pass
    raise RuntimeError(f"{path}: only .wav and .flac can be decoded without soundfile installed")


def to_pcm32(wave: np.ndarray) -> np.ndarray:
    """The reference's save_audio scaling (lib_ongaku_test.py:102-112): divide by the
    peak only when it exceeds 1, multiply by 2**31-1, truncate toward zero."""
    wave = np.asarray(wave)
    if wave.dtype == np.int32:
        return wave
    peak = np.max(np.abs(wave)) if wave.size else 0.0
    if peak > 1:
        wave = wave / peak
    return (wave * (2 ** 31 - 1)).astype(np.int32)


def write_wav_pcm32(path: str, wave: np.ndarray, sr: int) -> None:
    """PCM_32 little-endian WAV; ``wave`` is [samples] or [channels, samples] float or int32."""
    pcm = to_pcm32(wave)
    if pcm.ndim == 2:
        pcm = pcm.T if pcm.shape[0] in (1, 2) else pcm
        ch = pcm.shape[1]
    else:
        ch = 1
    raw = np.ascontiguousarray(pcm).astype("<i4").tobytes()
    hdr = b"RIFF" + struct.pack("<I", 36 + len(raw)) + b"WAVE"
    hdr += b"fmt " + struct.pack("<IHHIIHH", 16, 1, ch, sr, sr * ch * 4, ch * 4, 32)
    hdr += b"data" + struct.pack("<I", len(raw))
    with open(path, "wb") as f:
        f.write(hdr + raw)


def write_wav_pcm16(path: str, wave: np.ndarray, sr: int) -> None:
    """Helper for synthetic test inputs (mono float -> PCM_16)."""
    pcm = np.clip(np.round(np.asarray(wave, np.float64) * 32768.0), -32768, 32767).astype("<i2")
    raw = pcm.tobytes()
    hdr = b"RIFF" + struct.pack("<I", 36 + len(raw)) + b"WAVE"
    hdr += b"fmt " + struct.pack("<IHHIIHH", 16, 1, 1, sr, sr * 2, 2, 16)
    hdr += b"data" + struct.pack("<I", len(raw))
    with open(path, "wb") as f:
        f.write(hdr + raw)


def save_audio(filename: str, waveform, sample_rate: int) -> str:
    """lib_ongaku_test.py:89-143: PCM_32 .wav and 24-bit .flac natively.  The reference
    encodes .mp3 / .flac through pydub + ffmpeg; without an encoder for the requested container (.mp3; dataset mode names
    its outputs after the source file's extension, ddsp_matcher.py:1133) the audio is written as PCM_32 .wav next to
    the requested name, with a warning, instead of losing a finished conversion.  Returns the path written."""
    wave = np.asarray(waveform)
    if filename.endswith(".wav"):
        write_wav_pcm32(filename, wave, sample_rate)
        return filename
    if filename.endswith(".flac"):
        # the reference hands int32 samples to ffmpeg's FLAC encoder, which keeps their top 24 bits
        pcm = to_pcm32(wave)
        if pcm.ndim == 2 and pcm.shape[0] not in (1, 2):
            pcm = pcm.T
        write_flac(filename, np.ascontiguousarray(pcm) >> 8, sample_rate, bits=24)
        return filename
    alt = os.path.splitext(filename)[0] + ".wav"
    import warnings
    warnings.warn(f"{filename}: no encoder for this container in this environment (the reference uses pydub/ffmpeg); "
                  f"writing PCM_32 WAV to {alt} instead")
    write_wav_pcm32(alt, wave, sample_rate)
    return alt
