"""CPU: the FLAC decoder / encoder of the library (csrc/flac.hip, host code behind the C-ABI; the reference reads .flac
through torchaudio.load, ddsp_prematch_dataset.py:332, and its prematch builder globs *.flac, :1469-1473).  FLAC is
lossless: the decoder must return exactly the encoded samples.  Streams come from (a) the library's own encoder and (b) an
independent bit-level writer (tests/flac_writer.py) that exercises every construct of RFC 9639 the decoder knows."""
import os

import numpy as np
import pytest

from knn_svc_amd import audio_io, synthetic as S
from tests import flac_writer


def _clip(n, seed, bits, ch=1):
    w, _ = S.synth_clip(n, seed)
    rng = np.random.default_rng(seed)
    x = np.clip(np.round(w * (1 << (bits - 1)) * 0.9 + rng.normal(0, 2.0, n)), -(1 << (bits - 1)), (1 << (bits - 1)) - 1).astype(np.int32)
    if ch == 1:
        return x[None]
    return np.stack([x, np.clip(np.roll(x, 5) // 2 + rng.integers(-3, 4, n), -(1 << (bits - 1)), (1 << (bits - 1)) - 1).astype(np.int32)])


@pytest.mark.parametrize("n,bits,ch,sr", [(16000 * 3 + 123, 16, 1, 16000), (4096 * 2, 24, 2, 44100), (700, 16, 2, 22050), (1, 16, 1, 16000),
                                          (4097, 8, 1, 8000), (9000, 20, 1, 48000), (5000, 12, 2, 32000)])
def test_flac_library_encoder_round_trip(tmp_path, n, bits, ch, sr):
    pcm = _clip(n, 10 + bits, bits, ch)
    p = str(tmp_path / "a.flac")
    audio_io.write_flac(p, pcm, sr, bits=bits)
    got, gsr, gbits = audio_io.read_flac_pcm(p)
    assert gsr == sr and gbits == bits and got.shape == pcm.shape and np.array_equal(got, pcm)
    if n > 4096 and bits >= 16:
        assert os.path.getsize(p) < pcm.size * ((bits + 7) // 8)              # it does compress a tonal signal
    x, _ = audio_io.load_audio(p)
    assert x.dtype == np.float32 and np.array_equal(x, (pcm.astype(np.float64) / (1 << (bits - 1))).astype(np.float32))


@pytest.mark.parametrize("bits,ch,sr,n", [(16, 1, 16000, 12000), (24, 2, 44100, 9000), (8, 1, 8000, 5000), (12, 2, 22050, 7000),
                                          (20, 1, 48000, 6000), (16, 2, 37123, 8000), (16, 1, 96000, 3000)])
@pytest.mark.parametrize("seed", [0, 1, 2])
def test_flac_decoder_on_independent_writer_streams(tmp_path, bits, ch, sr, n, seed):
    """CONSTANT / VERBATIM / FIXED 0-4 / LPC 1-12 subframes, Rice and Rice2, partition orders 0-3, escaped partitions, wasted
    bits, left/side, side/right and mid/side stereo, every block-size code incl. the 8- and 16-bit explicit sizes, explicit and
    implicit sample-rate / sample-size codes, sample-number (variable block size) headers, ID3v2 prefix, extra metadata."""
    pcm = _clip(n, 100 * seed + bits, bits, ch)
    if seed == 1:
        pcm[:, 1000:1600] = 0; pcm[:, 2000:2600] = (pcm[:, 2000:2600] >> 3) << 3      # a digital-silence run and wasted bits
    data = flac_writer.encode(pcm, bits, sr, seed=seed, id3=(seed == 2), extra_blocks=(seed != 0))
    p = str(tmp_path / "w.flac")
    open(p, "wb").write(data)
    got, gsr, gbits = audio_io.read_flac_pcm(p)
    assert gsr == sr and gbits == bits and np.array_equal(got, pcm)


def test_flac_writer_streams_cover_the_format(tmp_path):
    """The randomised writer really produced every construct (so the test above is not vacuous)."""
    flac_writer.USED.clear()
    for bits, ch, sr, n in [(16, 1, 16000, 12000), (24, 2, 44100, 9000), (12, 2, 22050, 7000), (16, 2, 37123, 8000), (16, 1, 96000, 3000),
                            (20, 1, 123450, 4000)]:
        for seed in range(4):
            pcm = _clip(n, 100 * seed + bits, bits, ch)
            pcm[:, 1000:1600] = 0; pcm[:, 2000:2600] = (pcm[:, 2000:2600] >> 3) << 3
            p = str(tmp_path / "w.flac")
            open(p, "wb").write(flac_writer.encode(pcm, bits, sr, seed=seed))
            got, gsr, _b = audio_io.read_flac_pcm(p)
            assert gsr == sr and np.array_equal(got, pcm)
    want = {"constant", "verbatim", "fixed", "lpc", "rice0", "rice1", "escape", "wasted", "porder0", "porder1", "porder2", "porder3",
            "stereo0", "stereo1", "stereo2", "stereo3", "bs6", "bs7", "bs1", "bs2", "bs8", "bs10", "bs12", "rate0", "rate12", "rate13", "rate14"}
    assert want <= flac_writer.USED, want - flac_writer.USED


def test_flac_corruption_is_detected(tmp_path):
    from knn_svc_amd._lib import KnnSvcError
    pcm = _clip(20000, 5, 16, 1)
    p = str(tmp_path / "a.flac")
    audio_io.write_flac(p, pcm, 16000, bits=16)
    data = bytearray(open(p, "rb").read())
    bad = bytearray(data); bad[len(bad) // 2] ^= 0x10                               # a payload bit: frame CRC-16
    open(p, "wb").write(bad)
    with pytest.raises((KnnSvcError, ValueError)):
        audio_io.read_flac_pcm(p)
    bad = bytearray(data); bad[4 + 4 + 18 + 3] ^= 0xFF                              # the MD5 in STREAMINFO
    open(p, "wb").write(bad)
    with pytest.raises(ValueError):
        audio_io.read_flac_pcm(p)
    open(p, "wb").write(data[:len(data) // 2])                                      # truncated file
    with pytest.raises((KnnSvcError, ValueError)):
        audio_io.read_flac_pcm(p)
    open(p, "wb").write(b"RIFFxxxxWAVE")
    with pytest.raises(KnnSvcError):
        audio_io.read_flac_pcm(p)


def test_save_audio_flac_keeps_the_top_24_bits(tmp_path):
    """save_audio('.flac'): the reference hands its int32 samples to ffmpeg's FLAC encoder (lib_ongaku_test.py:122-143), which
    stores 24 bits; reading the file back gives to_pcm32(wave) >> 8."""
    w, _ = S.synth_clip(16000, 3)
    p = audio_io.save_audio(str(tmp_path / "o.flac"), w[None], 16000)
    assert p.endswith(".flac")
    got, sr, bits = audio_io.read_flac_pcm(p)
    assert sr == 16000 and bits == 24 and np.array_equal(got[0], audio_io.to_pcm32(w) >> 8)
