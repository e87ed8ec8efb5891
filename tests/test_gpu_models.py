"""GPU: model-level parity of the HIP path (through the C ABI) against the golden fixtures
captured from the reference and against the CPU oracle at full model dimensions."""
import os

import numpy as np
import pytest
import torch

from knn_svc_amd import audio_io, config as C, synthetic as S

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _t(a):
    return torch.from_numpy(np.asarray(a))


def _maxdiff(a, b):
    return float((a.detach().cpu().double() - _t(b).double()).abs().max())


def _rms(a, b):
    d = a.detach().cpu().double() - _t(b).double()
    return float(d.pow(2).mean().sqrt())


def test_wavlm_tiny_golden(golden):
    from knn_svc_amd.wavlm import WavLMEncoder
    g = golden("g1_wavlm_tiny")
    cfg = C.WAVLM_TINY
    sd = S.seeded_state(S.wavlm_param_spec(cfg), int(g["seed"]))
    wav, _ = S.synth_clip(int(g["n_samples"]), int(g["clip_seed"]))
    x = torch.from_numpy(np.pad(wav, (0, 320)))[None].to(DEV)
    for nl in (0, 1, 3):
        enc = WavLMEncoder(sd, cfg, DEV, n_layers=nl)
        out = enc.encode_batch(x)[0]
        d = _maxdiff(out, g[f"layer{nl}"])
        print(f"tiny WavLM layer {nl}: max|d| {d:.2e}")
        assert d < 1e-4, (nl, d)


def test_wavlm_general_layer_weighting_golden(golden):
    """A layer weighting that is not one-hot (ddsp_prematch_dataset.py:349-350: ``(feats * w[:, None]).sum(0)`` over the stacked
    layer results): the encoder's weighted sum against the same combination of the reference's own layer outputs (fixture g1:
    layers 0, 1 and 3 of the tiny model), and the one-hot special case back to the plain layer output."""
    from knn_svc_amd.wavlm import WavLMEncoder
    g = golden("g1_wavlm_tiny")
    cfg = C.WAVLM_TINY
    sd = S.seeded_state(S.wavlm_param_spec(cfg), int(g["seed"]))
    wav, _ = S.synth_clip(int(g["n_samples"]), int(g["clip_seed"]))
    x = torch.from_numpy(np.pad(wav, (0, 320)))[None].to(DEV)
    enc = WavLMEncoder(sd, cfg, DEV, n_layers=3)
    plain = enc.encode_batch(x)[0].clone()
    w = [0.25, 0.5, 0.0, 0.25]
    enc.set_layer_mix(w)
    uid = enc.uid
    for _ in range(3):                                   # eager, capture, replay
        out = enc.encode_batch(x)[0]
        ref = (torch.from_numpy(g["layer0"]) * w[0] + torch.from_numpy(g["layer1"]) * w[1]) + torch.from_numpy(g["layer3"]) * w[3]
        d = _maxdiff(out, ref.numpy())
        assert d < 1e-4, d
    with pytest.raises(ValueError):
        enc.set_layer_mix([0.0, 0.0, 0.0, 0.5, 0.5])     # layer 4 of a 3-layer encoder
    enc.set_layer_mix([0.0, 0.0, 0.0, 1.0])              # one-hot on the exit layer: the plain path again
    assert enc.layer_mix is None and enc.uid != uid
    assert torch.equal(enc.encode_batch(x)[0], plain)


def test_wavlm_chunked_golden(golden):
    from knn_svc_amd.wavlm import WavLMEncoder
    g = golden("g1b_wavlm_chunked")
    cfg = C.WAVLM_TINY
    sd = S.seeded_state(S.wavlm_param_spec(cfg), int(g["seed"]))
    wav, _ = S.synth_clip(int(g["n_samples"]), int(g["clip_seed"]))
    enc = WavLMEncoder(sd, cfg, DEV, n_layers=2)
    f = enc.full_features(torch.from_numpy(wav).to(DEV))
    assert f.shape[0] == int(g["n_frames"])
    assert _maxdiff(f[::10], g["rows"]) < 1e-4
    many = enc.encode_many([torch.from_numpy(wav).to(DEV), torch.from_numpy(wav[:50000]).to(DEV)])
    assert torch.equal(many[0], f) and many[1].shape[0] == enc.n_frames(50000 + 320 - 50000 % 320)


def test_wavlm_large_one_layer_golden(golden):
    from knn_svc_amd.wavlm import WavLMEncoder
    g = golden("g1c_wavlm_large1")
    cfg = dict(C.WAVLM_LARGE, encoder_layers=1)
    sd = S.seeded_state(S.wavlm_param_spec(cfg), int(g["seed"]))
    wav, _ = S.synth_clip(int(g["n_samples"]), int(g["clip_seed"]))
    x = torch.from_numpy(np.pad(wav, (0, 320)))[None].to(DEV)
    for nl in (0, 1):
        out = WavLMEncoder(sd, cfg, DEV, n_layers=nl).encode_batch(x)[0]
        d = _maxdiff(out, g[f"layer{nl}"])
        scale = float(np.abs(g[f"layer{nl}"]).max())
        print(f"large WavLM layer {nl}: max|d| {d:.2e} (|x|max {scale:.2f})")
        assert d < 2e-4 * max(1.0, scale)


def test_wavlm_large_six_layers_vs_oracle():
    """Full WavLM-Large dimensions, 6 layers, two 2 s chunks in one batch vs the CPU oracle."""
    from knn_svc_amd.wavlm import WavLMEncoder
    from oracle import wavlm_ref
    cfg = C.WAVLM_LARGE
    sd = S.seeded_state(S.wavlm_param_spec(cfg, 6), seed=1)
    w0, _ = S.synth_clip(32000, 11); w1, _ = S.synth_clip(32000, 12)
    x = torch.from_numpy(np.stack([np.pad(w0, (0, 320)), np.pad(w1, (0, 320))]))
    ref = wavlm_ref.extract_layer(sd, cfg, x, 6)
    out = WavLMEncoder(sd, cfg, DEV, n_layers=6).encode_batch(x.to(DEV))
    d = float((out.cpu() - ref).abs().max())
    rel = d / float(ref.abs().max())
    print(f"WavLM-Large 6 layers: max|d| {d:.2e}, relative {rel:.2e}")
    assert rel < 1e-4


def test_vocoder_tiny_golden(golden):
    from knn_svc_amd.vocoder import Vocoder
    g = golden("g7_vocoder")
    h = C.HIFIGAN_TINY
    c, f0, harm = _t(g["c"]).to(DEV), _t(g["f0"]).to(DEV), _t(g["harm"]).to(DEV)
    for kind, seed in (("mix", 63), ("f0", 64)):
        sd = S.seeded_state(S.generator_param_spec(h, kind), seed)
        y = Vocoder(sd, h, kind, DEV).forward(c, f0, harm if kind == "mix" else None)
        d, r = _maxdiff(y, g["wave_" + kind]), _rms(y, g["wave_" + kind])
        print(f"tiny generator {kind}: max|d| {d:.2e} rms {r:.2e}")
        assert r < 1e-5 and d < 1e-4


def test_vocoder_full_vs_oracle():
    """Full-size 'mix' generator (22.9 M parameters), 24 frames, vs the CPU oracle."""
    from knn_svc_amd.vocoder import Vocoder
    from oracle import vocoder_ref
    h = C.HIFIGAN_V1
    sd = S.seeded_state(S.generator_param_spec(h, "mix"), 2)
    g = torch.Generator().manual_seed(3)
    N = 24
    c = torch.randn(N, 1024, generator=g)
    _, f0 = S.synth_clip(N * 320, 5); f0 = torch.from_numpy(f0[:N].copy())
    harm = torch.rand(N, 49, generator=g) * 0.02
    ref = vocoder_ref.synthesizer(sd, h, "mix", c[None], f0[None, :, None], harm[None])[0, 0]
    y = Vocoder(sd, h, "mix", DEV).forward(c.to(DEV), f0.to(DEV), harm.to(DEV))
    r = float((y.cpu() - ref).pow(2).mean().sqrt())
    print(f"full generator: rms {r:.2e}, ref rms {float(ref.pow(2).mean().sqrt()):.3f}")
    assert r < 1e-4


def test_vocoder_full_size_golden(golden):
    """Fixture G7c: the reference's own full-size generators (22.9 M parameters; 'mix' with the additive synthesiser and 'f0'),
    60 frames — the GPU waveform against the reference's, not only against the oracle."""
    from knn_svc_amd.vocoder import Vocoder
    from tests.gen_golden_inputs import vocoder_full_inputs
    g = golden("g7c_vocoder_full")
    h = C.HIFIGAN_V1
    c, f0, harm = vocoder_full_inputs(int(g["n"]))
    for kind, seed in (("mix", 2), ("f0", 3)):
        sd = S.seeded_state(S.generator_param_spec(h, kind), seed)
        voc = Vocoder(sd, h, kind, DEV)
        y = voc.forward(c[0].to(DEV), f0[0, :, 0].to(DEV), harm[0].to(DEV)) if kind == "mix" else voc.forward(c[0].to(DEV), f0[0, :, 0].to(DEV))
        ref = torch.from_numpy(g["wave_" + kind])
        r = float((y.cpu().reshape(-1) - ref).pow(2).mean().sqrt())
        print(f"full-size generator ({kind}) vs the reference's waveform: rms {r:.2e} (signal rms {float(ref.pow(2).mean().sqrt()):.3f})")
        assert y.numel() == ref.numel() and r < 1e-5


def _write_tiny_dataset(tmp_path, g):
    (tmp_path / "a").mkdir(); (tmp_path / "b").mkdir()
    src_wav, src_f0 = S.synth_clip(3 * 16000 + 77, seed=int(g["src_seed"]))
    audio_io.write_wav_pcm16(str(tmp_path / "a" / "src.wav"), src_wav, 16000)
    np.save(tmp_path / "a" / "src_f0.npy", src_f0 * float(g["f0_scale"]))
    for i in range(3):
        w, f = S.synth_clip(4 * 16000 + 5 * i, seed=int(g["pool_seed0"]) + i)
        audio_io.write_wav_pcm16(str(tmp_path / "b" / f"u{i}.wav"), w, 16000)
        np.save(tmp_path / "b" / f"u{i}_f0.npy", f)
    return str(tmp_path / "a" / "src.wav"), tmp_path / "b"


def test_e2e_tiny_golden(golden, tmp_path):
    """Files -> match_at_inference_time -> vocode, against the waveform the reference produced (g11)."""
    from knn_svc_amd.matcher import KNeighborsVC
    from knn_svc_amd.matching import match_at_inference_time
    from knn_svc_amd.vocoder import Vocoder
    from knn_svc_amd.wavlm import WavLMEncoder
    g = golden("g11_e2e")
    cfg, h = C.WAVLM_TINY, C.HIFIGAN_TINY
    sdw = S.seeded_state(S.wavlm_param_spec(cfg), seed=11)
    enc = WavLMEncoder(sdw, cfg, DEV, n_layers=2)
    srcp, poolp = _write_tiny_dataset(tmp_path, g)
    for kind, ckpt, post_opt, seed in (("mix", "mix", "no_post_opt", 63), ("mix", "mix", "post_opt_0.2", 63),
                                       ("f0", "wavlm_only", "no_post_opt", 64)):
        sdg = S.seeded_state(S.generator_param_spec(h, kind), seed)
        knn = KNeighborsVC(enc, Vocoder(sdg, h, kind, DEV), h, DEV)
        if kind == "mix":
            of, hw, _a, sf0 = match_at_inference_time(srcp, poolp, enc, knn.weighting, knn.weighting, prioritize_f0=True,
                                                      ckpt_type=ckpt, post_opt=post_opt, tgt_dataset_path=tmp_path,
                                                      duration_limit=int(g["duration_limit"]))
            y = knn.vocode(of[srcp][None], sf0[srcp][None, :, None], hw[srcp][None]).squeeze()
        else:
            of, _a, sf0 = match_at_inference_time(srcp, poolp, enc, knn.weighting, knn.weighting, prioritize_f0=True,
                                                  ckpt_type=ckpt, tgt_dataset_path=tmp_path,
                                                  duration_limit=int(g["duration_limit"]))
            y = knn.vocode(of[srcp][None], sf0[srcp][None, :, None]).squeeze()
        ref = g[f"{ckpt}__{post_opt}"]
        r = _rms(y, ref)
        print(f"e2e {ckpt} {post_opt}: waveform rms error {r:.2e} (signal rms {float(np.sqrt((ref ** 2).mean())):.3f})")
        assert y.shape[0] == ref.shape[0]
        assert r < 1e-4, (ckpt, post_opt, r)      # north-star tolerance: waveform within 1e-4 RMS
    # different matching and synthesis layer weightings (ddsp_prematch_dataset.py:349-350: two feature sets per target file — the
    # search runs on the matching features, the smoothness weights and the weighted sums on the synthesis features)
    sdg = S.seeded_state(S.generator_param_spec(h, "mix"), 63)
    knn = KNeighborsVC(enc, Vocoder(sdg, h, "mix", DEV), h, DEV)
    synth1 = torch.zeros_like(knn.weighting); synth1[1] = 1.0
    of, hw, _a, sf0 = match_at_inference_time(srcp, poolp, enc, knn.weighting, synth1, prioritize_f0=True, ckpt_type="mix",
                                              post_opt="post_opt_0.2", tgt_dataset_path=tmp_path, duration_limit=int(g["duration_limit"]))
    y = knn.vocode(of[srcp][None], sf0[srcp][None, :, None], hw[srcp][None]).squeeze()
    r = _rms(y, g["mix__post_opt_0.2__synth_layer1"])
    print(f"e2e mix post_opt_0.2, synthesis weighting on layer 1: waveform rms error {r:.2e}")
    assert r < 1e-4 and _rms(y, g["mix__post_opt_0.2"]) > 1e-3            # ... and it is not the equal-weighting result
    assert enc.layer_mix is None                                           # the shared encoder is back on the matching weighting


def test_e2e_full_architecture_golden(golden, tmp_path):
    """Fixture G11c: the REFERENCE's waveform for files -> match_at_inference_time -> vocode with the full architecture (WavLM-Large
    x six layers, full-size 'mix' generator, seeded weights, post_opt_0.2): the product's waveform against it, north-star bar."""
    from knn_svc_amd.matcher import KNeighborsVC
    from knn_svc_amd.matching import match_at_inference_time
    from knn_svc_amd.vocoder import Vocoder
    from knn_svc_amd.wavlm import WavLMEncoder
    g = golden("g11c_e2e_full")
    (tmp_path / "a").mkdir(); (tmp_path / "b").mkdir()
    src_wav, src_f0 = S.synth_clip(3 * 16000 + 40, seed=int(g["src_seed"]))
    audio_io.write_wav_pcm16(str(tmp_path / "a" / "src.wav"), src_wav, 16000)
    np.save(tmp_path / "a" / "src_f0.npy", (src_f0 * float(g["f0_scale"])).astype(np.float32))
    for i in range(3):
        w, f = S.synth_clip(5 * 16000 + 7 * i, seed=int(g["pool_seed0"]) + i)
        audio_io.write_wav_pcm16(str(tmp_path / "b" / f"u{i}.wav"), w, 16000)
        np.save(tmp_path / "b" / f"u{i}_f0.npy", f.astype(np.float32))
    enc = WavLMEncoder(S.seeded_state(S.wavlm_param_spec(C.WAVLM_LARGE, 6), seed=1), C.WAVLM_LARGE, DEV, n_layers=6)
    knn = KNeighborsVC(enc, Vocoder(S.seeded_state(S.generator_param_spec(C.HIFIGAN_V1, "mix"), 2), C.HIFIGAN_V1, "mix", DEV), C.HIFIGAN_V1, DEV)
    srcp = str(tmp_path / "a" / "src.wav")
    of, hw, _a, sf0 = match_at_inference_time(srcp, tmp_path / "b", enc, knn.weighting, knn.weighting, prioritize_f0=True,
                                              ckpt_type="mix", post_opt="post_opt_0.2", tgt_dataset_path=tmp_path)
    y = knn.vocode(of[srcp][None], sf0[srcp][None, :, None], hw[srcp][None]).squeeze()
    ref = g["wave"]
    r = _rms(y, ref)
    print(f"e2e full architecture, mix post_opt_0.2: waveform rms error vs the reference {r:.2e} (signal rms {float(np.sqrt((ref ** 2).mean())):.3f})")
    assert y.shape[0] == ref.shape[0] and r < 1e-4


def test_special_match_writes_reference_named_file(golden, tmp_path):
    from knn_svc_amd.matcher import KNeighborsVC
    from knn_svc_amd.vocoder import Vocoder
    from knn_svc_amd.wavlm import WavLMEncoder
    g = golden("g11_e2e")
    cfg, h = C.WAVLM_TINY, C.HIFIGAN_TINY
    enc = WavLMEncoder(S.seeded_state(S.wavlm_param_spec(cfg), seed=11), cfg, DEV, n_layers=2)
    knn = KNeighborsVC(enc, Vocoder(S.seeded_state(S.generator_param_spec(h, "mix"), 63), h, "mix", DEV), h, DEV)
    srcp, poolp = _write_tiny_dataset(tmp_path, g)
    y = knn.special_match(srcp, str(poolp / "u0.wav"), ckpt_type="mix", post_opt="post_opt_0.2")
    out = tmp_path / "a" / "src_to_u0_knn_mix_post_opt_0.2.wav"
    assert out.is_file()
    x, sr = audio_io.read_wav(str(out))
    assert sr == 16000 and x.shape == (1, y.numel())
    assert np.max(np.abs(x[0] - np.clip(y.cpu().numpy(), -1, 1))) < 1e-6


def test_bulk_match_dataset_mode_uses_pool_store(tmp_path):
    """Folder -> folder conversion (ddsp_matcher.py:1027-1155): output tree, subset file, duration-limit prefix,
    and the device-resident pool store (every file encoded once, identical waveforms with the store off)."""
    from knn_svc_amd import matching, pool_cache
    from knn_svc_amd.inference import output_dir_for
    from knn_svc_amd.matcher import KNeighborsVC
    from knn_svc_amd.vocoder import Vocoder
    from knn_svc_amd.wavlm import WavLMEncoder
    cfg, h = C.WAVLM_TINY, C.HIFIGAN_TINY
    enc = WavLMEncoder(S.seeded_state(S.wavlm_param_spec(cfg), seed=11), cfg, DEV, n_layers=2)
    knn = KNeighborsVC(enc, Vocoder(S.seeded_state(S.generator_param_spec(h, "mix"), 63), h, "mix", DEV), h, DEV)
    root = tmp_path / "data"
    for s, spk in enumerate(("spkA", "spkB", "spkC")):
        (root / spk).mkdir(parents=True)
        for u in range(2):
            w, f = S.synth_clip(16000 + 320 * (s + u), seed=300 + 10 * s + u)
            audio_io.write_wav_pcm16(str(root / spk / f"u{u}.wav"), w, 16000)
            np.save(root / spk / f"u{u}_f0.npy", f)
    csvp = tmp_path / "subset.csv"
    csvp.write_text("src,tgt,key,x,label\n" + "".join(f"a,b,u0/{t},x,0\n" for t in ("spkA", "spkB", "spkC")) + "a,b,u1/spkB,x,1\n")

    def run(out_dir):
        calls = {"n": 0}
        orig = enc.encode_many
        def counted(wavs, **kw):
            calls["n"] += len(wavs)
            return orig(wavs, **kw)
        enc.encode_many = counted
        try:
            written = knn.bulk_match(str(root), str(root), str(out_dir), ckpt_type="mix", post_opt="post_opt_0.2",
                                     required_subset_file=str(csvp), duration_limit=100)
        finally:
            enc.encode_many = orig
        return written, calls["n"]
    matching._POOL_CACHE = pool_cache.PoolCache()
    out1 = output_dir_for(str(root), str(root), "mix", "post_opt_0.2", 100)
    assert "duration_limit_100_data_to_data_mix_post_opt_post_opt_0.2" in out1
    w1, n1 = run(tmp_path / "o1")
    # 3 x 2 ordered speaker pairs, only utterance u0 is in the subset (label 0), one output per pair
    assert len(w1) == 6 and all(p.endswith(".wav") and "/u0/" in p for p in w1)
    assert (tmp_path / "o1" / "spkA" / "u0" / "spkB.wav").is_file()
    assert n1 == 6                                         # six files, each encoded exactly once
    matching._POOL_CACHE = pool_cache.PoolCache(budget_bytes=0)
    w2, n2 = run(tmp_path / "o2")
    assert n2 == 24                                        # the reference's behaviour: both pools per pair
    for a, b in zip(sorted(w1), sorted(w2)):
        xa, _ = audio_io.read_wav(a); xb, _ = audio_io.read_wav(b)
        assert np.array_equal(xa, xb)
    matching._POOL_CACHE = None


def test_lane_pipeline_matches_sequential_order():
    """pipeline.LanePipeline: three match bodies in flight on their own stream pairs + a tail stream give
    bit-identical results to running the items one after the other (same kernels, same inputs)."""
    from knn_svc_amd import matching, ops
    from knn_svc_amd.pipeline import LanePipeline
    g = torch.Generator().manual_seed(5)
    P = S.clustered_features(3000, 256, seed=3).to(DEV)
    Pf0 = (torch.rand(3000, generator=g) * 200 + 100).to(DEV)
    Pf0[::7] = 0
    Ph = torch.rand(3000, 49, generator=g).to(DEV)
    items = []
    for i in range(5):
        q = S.clustered_features(120 + 8 * i, 256, seed=20 + i).to(DEV)
        f = (torch.rand(q.shape[0], generator=g) * 200 + 100).to(DEV)
        f[::5] = 0
        items.append((q, f))
    body = lambda it, flags=None: matching.match_features(it[0], it[1], P, Pf0, Ph, "mix", "post_opt_0.2", nan_flags=flags)
    seq = [body(it) for it in items]
    flags = []
    tail = lambda it, h: (h[0] * 2.0, h[1], h[2])
    par = LanePipeline(DEV, lanes=3).run(items, lambda it: body(it, flags), tail)
    assert len(flags) == 5
    for f in flags:
        ops.raise_if_nan(f)
    torch.cuda.synchronize()
    for (a0, a1, a2), (b0, b1, b2) in zip(seq, par):
        assert torch.equal(a0 * 2.0, b0) and torch.equal(a1, b1) and torch.equal(a2, b2)


def test_pipeline_streams_are_measured_to_run_side_by_side():
    """Round 5: the stream scheduler no longer trusts HIP's stream -> hardware-queue mapping (the bench step moved by +-8 % with the
    number of streams the process had created before).  Every lane, tail and partner stream is taken from torch's pool by
    MEASUREMENT (pipeline.new_stream / stream_contention: a dispatch-bound probe launch on both streams at once): whatever was
    created before, a pipeline's lane and tail — and the tail's partner — end up side by side, not on one hardware queue."""
    from knn_svc_amd import matching, pipeline
    dummies = [torch.cuda.Stream() for _ in range(5)]          # noqa: F841
    for st in dummies:
        with torch.cuda.stream(st):
            torch.zeros(8, device=DEV)
    pipe = pipeline.LanePipeline(torch.device(DEV, 0), lanes=1)
    lane, tail = pipe.lanes[0], pipe.tail_streams[0]
    with torch.cuda.stream(tail):
        partner = matching._side_stream(torch.device(DEV, 0))
    c_lt = pipeline.stream_contention(lane, tail, 0)
    c_tp = pipeline.stream_contention(tail, partner, 0)
    c_lp = pipeline.stream_contention(lane, partner, 0)
    same = pipeline.stream_contention(lane, lane, 0)                # one stream against itself: strictly one after the other
    print(f"contention lane/tail {c_lt:.2f}, tail/partner {c_tp:.2f}, lane/partner {c_lp:.2f}; one stream against itself {same:.2f}; "
          f"probe log {pipeline._PROBE['log'][-6:]}")
    assert same > 1.9                                               # the probe can tell
    assert c_lt <= pipeline.CONTENTION_OK + 0.1 and c_tp <= pipeline.CONTENTION_OK + 0.1 and c_lp <= 2.4


@pytest.mark.parametrize("size", ["tiny", "large"])
def test_wavlm_one_call_equals_the_host_sequenced_forward(size, monkeypatch):
    """WavLM.extract_features (wavlm/WavLM.py:323-375) behind ONE C call (knnsvc_wavlm_encode: an opaque handle owns
    the descriptor, the call enqueues the layer sequence) against the launch-by-launch host sequence of rounds 1-4 — same kernels,
    same arguments: the same bits.  Exact-length batches, ragged batches with a device-side length mask, a general layer
    weighting, and through the hipGraph cache; tiny config (fp32 layouts between the narrow conv layers: range slots + absmax
    launches) and WavLM-Large (everything in the split layout)."""
    from knn_svc_amd.wavlm import WavLMEncoder
    cfg, nl = (C.WAVLM_TINY, 3) if size == "tiny" else (C.WAVLM_LARGE, 6)
    enc = WavLMEncoder(S.seeded_state(S.wavlm_param_spec(cfg, nl), seed=5), cfg, DEV, nl)
    g = torch.Generator().manual_seed(3)
    B, Tb = (3, 60) if size == "tiny" else (2, 150)
    wav = (torch.randn(B, 320 * Tb + 80, generator=g) * 0.1).to(DEV)
    lens = torch.tensor([Tb, Tb - 7, Tb - 31][:B], dtype=torch.int32, device=DEV)

    def both(lens_, graphs=False):
        monkeypatch.setenv("KNNSVC_WAVLM_HOST_SEQ", "1")
        ref = enc._encode_batch(wav, lens_).clone()
        monkeypatch.delenv("KNNSVC_WAVLM_HOST_SEQ")
        assert enc._handle_ok()
        one = enc._encode_batch(wav, lens_).clone()
        assert torch.equal(ref, one), float((ref - one).abs().max())
        if graphs:
            for _ in range(3):                       # first sight eager, capture, replay
                y = enc.encode_batch(wav, lens_)
            assert torch.equal(ref, y)
        return ref
    a = both(None, graphs=True)
    b = both(lens, graphs=True)
    assert torch.equal(a[0], b[0]) and not torch.equal(a[1], b[1])          # the mask does something, and only to the masked rows' batch entries
    mix = [0.0] * (nl + 1); mix[1], mix[nl] = 0.25, 0.75
    enc.set_layer_mix(mix)
    c = both(lens)
    assert not torch.equal(b, c)
    enc.set_layer_mix(None)
    assert torch.equal(both(lens), b)
    import ctypes
    from knn_svc_amd import _lib
    h = enc._handle()
    assert int(_lib.load().knnsvc_wavlm_frames(h, wav.shape[1])) == Tb


@pytest.mark.parametrize("kind", ["mix", "f0"])
def test_generator_branches_in_one_grid_equal_one_launch_per_branch(kind):
    """the three ResBlock branches of a stage run as ONE grid per step (no per-branch streams any more).  The
    full-size generators, eager and through the bucketed hipGraph (ragged length), give the waveform of the
    one-launch-per-branch form bit for bit, and creating unrelated streams first changes nothing (the round-4 headline depended
    on how many streams the process had made before)."""
    from knn_svc_amd.vocoder import Vocoder, serial_resblocks
    h = C.HIFIGAN_V1
    voc = Vocoder(S.seeded_state(S.generator_param_spec(h, kind), 2), h, kind, DEV)
    g = torch.Generator().manual_seed(17)
    dummies = [torch.cuda.Stream() for _ in range(3)]          # noqa: F841  (unrelated streams: must not matter)
    for N in (150, 137):
        c = torch.randn(N, 1024, generator=g).to(DEV)
        harm = (torch.rand(N, 49, generator=g) * 0.02).to(DEV) if kind == "mix" else None
        _, f0 = S.synth_clip(N * 320, 40 + N); f0 = torch.from_numpy(f0[:N].copy()).to(DEV)
        with serial_resblocks():
            ref = voc._forward(c, f0, harm).clone()
        host = voc._forward_host(c, f0, harm).clone()             # merged grids, sequenced from the host
        assert torch.equal(ref, host), float((ref - host).abs().max())
        assert voc._handle_ok()
        eager = voc._forward(c, f0, harm).clone()                 # the same sequence behind ONE C call (knnsvc_generator_forward)
        assert torch.equal(ref, eager), float((ref - eager).abs().max())
        for _ in range(3):                                        # first sight eager, capture, replay
            y = voc.forward(c, f0, harm)
        assert torch.equal(ref, y)
    assert any(not (isinstance(k, tuple) and "serial" in k) for k in voc._graphs)


def test_generators_on_three_tails_equal_one_generator():
    """The stream pipeline runs the tail of item i on tail stream i mod 3 when it has several lanes; the generator keeps one hipGraph
    instance (and one memory pool) per tail.  Seven utterances of three lengths through three tails — first sight (eager), capture
    and replay on every tail — give the waveforms of the same generator run on one stream."""
    from knn_svc_amd import pipeline
    from knn_svc_amd.vocoder import Vocoder, serial_resblocks
    voc = Vocoder(S.seeded_state(S.generator_param_spec(C.HIFIGAN_V1, "mix"), 2), C.HIFIGAN_V1, "mix", DEV)
    g = torch.Generator().manual_seed(11)
    items = []
    for i in range(7):
        N = (120, 95, 150)[i % 3]
        c = torch.randn(N, 1024, generator=g).to(DEV); harm = (torch.rand(N, 49, generator=g) * 0.02).to(DEV)
        _, f0 = S.synth_clip(N * 320, 30 + i); f0 = torch.from_numpy(f0[:N].copy()).to(DEV)
        items.append((c, f0, harm))
    with serial_resblocks():
        ref = [voc.forward(*it).clone() for it in items]
        seen = []
        pipe = pipeline.LanePipeline(DEV, lanes=3)
        assert len(pipe.tail_streams) == 3
        for _ in range(3):                      # eager, capture, replay on each tail
            out = pipe.run(items, lambda it: it, lambda it, h: (seen.append(pipeline.current_tail()), voc.forward(*h))[1])
    torch.cuda.synchronize()
    assert seen[:7] == [0, 1, 2, 0, 1, 2, 0] and pipeline.current_tail() == 0
    assert any(isinstance(k, tuple) and k[-1] in (1, 2) for k in voc._graphs)          # instances of the other tails exist
    for a, b in zip(ref, out):
        assert torch.equal(a, b)


def test_prematch_files_match_reference(golden, tmp_path):
    """per_spk_extract on the GPU writes the reference's files (g12: pool.npy, pool_harmonics.npy, per-utterance
    pickles with slice / nearest_nbrs / nearest_nbrs_f0_priority / amp_ratio / harmonics_best_weight_para)."""
    import pickle
    from knn_svc_amd import prematch
    from knn_svc_amd.wavlm import WavLMEncoder
    from tests.prematch_common import write_dataset
    g = golden("g12_prematch")
    cfg = C.WAVLM_TINY
    enc = WavLMEncoder(S.seeded_state(S.wavlm_param_spec(cfg), seed=11), cfg, DEV, n_layers=2)
    lay = write_dataset(tmp_path / "data", g)
    with torch.inference_mode():
        prematch.per_spk_extract(enc, DEV, tmp_path / "data", tmp_path / "cached")
        prematch.per_spk_extract(enc, DEV, tmp_path / "data", tmp_path / "cached")     # second run: existing pickles, same slices
    rows = same = 0
    for name, utts in lay.items():
        pool = np.load(tmp_path / "cached" / name / "pool.npy")
        ref_pool = g[f"{name}__pool_f16"].astype(np.float32)
        assert pool.dtype == np.float32 and pool.shape == ref_pool.shape
        assert np.array_equal(pool, pool.astype(np.float16).astype(np.float32))            # fp16-representable
        assert float(np.mean(pool != ref_pool)) < 0.02 and float(np.abs(pool - ref_pool).max()) <= 4e-3
        harm = np.load(tmp_path / "cached" / name / "pool_harmonics.npy")
        assert float(np.abs(harm - g[f"{name}__pool_harmonics"]).max()) < 1e-5
        for i in range(len(utts)):
            with open(tmp_path / "cached" / name / f"u{i}.pt", "rb") as fh:
                d = pickle.load(fh)
            assert set(d) == {"slice", "nearest_nbrs", "nearest_nbrs_f0_priority", "harmonics_best_weight_para", "amp_ratio"}
            assert tuple(d["slice"]) == tuple(g[f"{name}__u{i}__slice"])
            nn, ref_nn = d["nearest_nbrs"], g[f"{name}__u{i}__nearest_nbrs"]
            assert nn.dtype == np.int64 and nn.shape == ref_nn.shape
            s, e = d["slice"]
            assert not np.any((nn >= s) & (nn < e)), "own-utterance rows must lose against the rest of the pool here"
            ok = np.all(nn[:, :4] == ref_nn[:, :4], axis=1) & \
                np.all(d["nearest_nbrs_f0_priority"][:, :4] == g[f"{name}__u{i}__nearest_nbrs_f0_priority"][:, :4], axis=1)
            rows += len(ok); same += int(ok.sum())
            assert float(np.mean([set(a) == set(b) for a, b in zip(nn, ref_nn)])) > 0.9
            ar, ref_ar = d["amp_ratio"], g[f"{name}__u{i}__amp_ratio"]
            assert float(np.abs(ar[ok] - ref_ar[ok]).max() / np.abs(ref_ar).max()) < 1e-4
            w = d["harmonics_best_weight_para"]
            assert w.shape == ref_ar.shape and abs(float(w.sum(1).mean()) - 1) < 1e-5
            if ok.all():
                assert float(np.abs(w - g[f"{name}__u{i}__harmonics_best_weight_para"]).max()) < 5e-3
    print(f"prematch: {same}/{rows} frames with identical first-4 neighbours (plain and f0-priority)")
    assert same == rows                       # measured: every frame (514 / 514)


def test_bulk_match_pipelined_vocoder_equals_sequential(tmp_path, monkeypatch):
    """Dataset mode with several utterances per speaker: the pipelined order (match bodies on lanes, generator as the
    tail stage, one pool preparation per pair) writes the same bytes as running every stage of every utterance in
    sequence on one stream."""
    from knn_svc_amd import matching, pipeline, pool_cache
    from knn_svc_amd.matcher import KNeighborsVC
    from knn_svc_amd.vocoder import Vocoder
    from knn_svc_amd.wavlm import WavLMEncoder
    cfg, h = C.WAVLM_TINY, C.HIFIGAN_TINY
    enc = WavLMEncoder(S.seeded_state(S.wavlm_param_spec(cfg), seed=11), cfg, DEV, n_layers=2)
    knn = KNeighborsVC(enc, Vocoder(S.seeded_state(S.generator_param_spec(h, "mix"), 63), h, "mix", DEV), h, DEV)
    root = tmp_path / "data"
    for s, spk in enumerate(("spkA", "spkB")):
        (root / spk).mkdir(parents=True)
        for u in range(4):
            w, f = S.synth_clip(16000 + 320 * (3 * s + u), seed=400 + 10 * s + u)
            audio_io.write_wav_pcm16(str(root / spk / f"u{u}.wav"), w, 16000)
            np.save(root / spk / f"u{u}_f0.npy", f)
    matching._POOL_CACHE = pool_cache.PoolCache()
    w1 = knn.bulk_match(str(root), str(root), str(tmp_path / "o1"), ckpt_type="mix", post_opt="post_opt_0.2")
    assert len(w1) == 8

    def sequential(self, items, head, tail=None):
        out = []
        for it in items:
            r = head(it)
            out.append(tail(it, r) if tail is not None else r)
        return out
    monkeypatch.setattr(pipeline.LanePipeline, "run", sequential)
    w2 = knn.bulk_match(str(root), str(root), str(tmp_path / "o2"), ckpt_type="mix", post_opt="post_opt_0.2")
    for a, b in zip(sorted(w1), sorted(w2)):
        xa, _ = audio_io.read_wav(a); xb, _ = audio_io.read_wav(b)
        assert xa.shape == xb.shape and np.array_equal(xa, xb)
    matching._POOL_CACHE = None


def test_pool_store_disk_tier_skips_encoding(tmp_path):
    """A fresh process (fresh resident store, fresh encoder object with the same weights) finds every file of a previous
    dataset-mode run in the on-disk tier: nothing is encoded and the converted files are byte-identical."""
    from knn_svc_amd import matching, pool_cache
    from knn_svc_amd.matcher import KNeighborsVC
    from knn_svc_amd.vocoder import Vocoder
    from knn_svc_amd.wavlm import WavLMEncoder
    cfg, h = C.WAVLM_TINY, C.HIFIGAN_TINY
    root = tmp_path / "data"
    for s, spk in enumerate(("spkA", "spkB")):
        (root / spk).mkdir(parents=True)
        for u in range(2):
            w, f = S.synth_clip(16000 + 320 * (s + 2 * u), seed=500 + 10 * s + u)
            audio_io.write_wav_pcm16(str(root / spk / f"u{u}.wav"), w, 16000)
            np.save(root / spk / f"u{u}_f0.npy", f)
    outs, encoded = [], []
    for run in range(2):
        enc = WavLMEncoder(S.seeded_state(S.wavlm_param_spec(cfg), seed=11), cfg, DEV, n_layers=2)
        knn = KNeighborsVC(enc, Vocoder(S.seeded_state(S.generator_param_spec(h, "mix"), 63), h, "mix", DEV), h, DEV)
        matching._POOL_CACHE = pool_cache.PoolCache(disk_dir=str(tmp_path / "store"))
        n = {"n": 0}
        orig = enc.encode_many
        def counted(wavs, _o=orig, _n=n, **kw):
            _n["n"] += len(wavs)
            return _o(wavs, **kw)
        enc.encode_many = counted
        outs.append(sorted(knn.bulk_match(str(root), str(root), str(tmp_path / f"o{run}"), ckpt_type="mix", post_opt="post_opt_0.2")))
        encoded.append(n["n"])
        assert matching._POOL_CACHE.disk_writes == (4 if run == 0 else 0)
    assert encoded == [4, 0], encoded
    other = WavLMEncoder(S.seeded_state(S.wavlm_param_spec(cfg), seed=12), cfg, DEV, n_layers=2)
    assert other.weights_fingerprint() != enc.weights_fingerprint()
    for a, b in zip(*outs):
        xa, _ = audio_io.read_wav(a); xb, _ = audio_io.read_wav(b)
        assert np.array_equal(xa, xb)
    matching._POOL_CACHE = None


def test_sample_content_single_file_cli_path(golden, tmp_path):
    """BASELINE cfg 1/2's input pair (6 s excerpts + their harvest f0 caches, tests/golden/sample_content) through the
    file-based single-file path (special_match: load, encode, match, vocode, PCM_32 write) against the reference's
    waveform for the same files (g13)."""
    import shutil
    from pathlib import Path
    from knn_svc_amd.matcher import KNeighborsVC
    from knn_svc_amd.vocoder import Vocoder
    from knn_svc_amd.wavlm import WavLMEncoder
    g = golden("g13_sample")
    fx = Path(__file__).parent / "golden" / "sample_content"
    for f in fx.iterdir():
        shutil.copy(f, tmp_path / f.name)
    cfg, h = C.WAVLM_TINY, C.HIFIGAN_TINY
    enc = WavLMEncoder(S.seeded_state(S.wavlm_param_spec(cfg), seed=11), cfg, DEV, n_layers=2)
    for kind, ckpt, post_opt, seed in (("mix", "mix", "post_opt_0.2", 63), ("f0", "wavlm_only", "no_post_opt", 64)):
        knn = KNeighborsVC(enc, Vocoder(S.seeded_state(S.generator_param_spec(h, kind), seed), h, kind, DEV), h, DEV)
        y = knn.special_match(str(tmp_path / "src.wav"), str(tmp_path / "tgt.wav"), ckpt_type=ckpt, post_opt=post_opt)
        ref = g[f"{ckpt}__{post_opt}"]
        r = _rms(y, ref)
        print(f"sample_content {ckpt} {post_opt}: rms error {r:.2e} (signal rms {float(np.sqrt((ref ** 2).mean())):.3f})")
        assert y.shape == ref.shape and r < 1e-4
        out = tmp_path / f"src_to_tgt_knn_{ckpt}_{post_opt}.wav"
        assert out.is_file()
        x, sr = audio_io.read_wav(str(out))
        assert sr == 16000 and x.shape[1] == ref.shape[0]


def test_pool_sharded_single_file_path_equals_unsharded(golden, tmp_path):
    """BASELINE cfg 4 orchestration (pool files sharded over ranks, replicated queries, per-shard top-32 merged after an
    RCCL all-gather, pool side arrays all-gathered) forced through a ONE-rank process group: same waveform as the
    unsharded path.  The multi-rank arithmetic of the merge is covered by tests/test_dist_cpu.py (gloo, world 2)."""
    import os
    import torch.distributed as dist
    from knn_svc_amd import matching
    from knn_svc_amd.matcher import KNeighborsVC
    from knn_svc_amd.vocoder import Vocoder
    from knn_svc_amd.wavlm import WavLMEncoder
    cfg, h = C.WAVLM_TINY, C.HIFIGAN_TINY
    enc = WavLMEncoder(S.seeded_state(S.wavlm_param_spec(cfg), seed=11), cfg, DEV, n_layers=2)
    knn = KNeighborsVC(enc, Vocoder(S.seeded_state(S.generator_param_spec(h, "mix"), 63), h, "mix", DEV), h, DEV)
    srcp, poolp = _write_tiny_dataset(tmp_path, golden("g11_e2e"))
    common = dict(topk=4, device=DEV, prioritize_f0=True, ckpt_type="mix", post_opt="post_opt_0.2", tgt_dataset_path=tmp_path)
    from pathlib import Path
    ref = matching.match_at_inference_time(Path(srcp), poolp, enc, knn.weighting, knn.weighting, pool_sharded=False, **common)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29700 + os.getpid() % 200))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(DEV, 0))
    try:
        got = matching.match_at_inference_time(Path(srcp), poolp, enc, knn.weighting, knn.weighting, pool_sharded=True, **common)
    finally:
        dist.destroy_process_group()
    for a, b in zip(ref, got):
        for k in a:
            if a[k] is not None:
                assert torch.equal(a[k], b[k]), k
    # round 5: DIFFERENT matching / synthesis layer weightings with a sharded pool (ddsp_prematch_dataset.py:349-350, 1157, 1260: the
    # search runs on the matching features, the gathers and the smoothness weights read the synthesis ones) — was NotImplementedError
    wm = torch.zeros(cfg["encoder_layers"] + 1); wm[2] = 1.0
    ws_ = torch.zeros(cfg["encoder_layers"] + 1); ws_[1] = 1.0
    ref2 = matching.match_at_inference_time(Path(srcp), poolp, enc, wm, ws_, pool_sharded=False, **common)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(DEV, 0))
    try:
        got2 = matching.match_at_inference_time(Path(srcp), poolp, enc, wm, ws_, pool_sharded=True, **common)
    finally:
        dist.destroy_process_group()
    for a, b in zip(ref2, got2):
        for k in a:
            if a[k] is not None:
                assert torch.equal(a[k], b[k]), k
    assert not all(torch.equal(ref[0][k], ref2[0][k]) for k in ref[0])          # the second weighting changes the features that go out


def test_sharded_search_reports_nan_once_agreed():
    """The per-shard searches defer the reference's NaN exit (lib_ongaku_test.py:166-169): dist.raise_if_any_nan() reads
    the flags once and raises."""
    from knn_svc_amd import dist as kd, ops
    q = S.clustered_features(64, 64, 3, n_centres=5).to(DEV)
    p = S.clustered_features(500, 64, 4, n_centres=5)
    kd._NAN_FLAGS.clear()
    kd._hip_local_topk(q, p.to(DEV), 8, 0)
    kd.raise_if_any_nan()                                   # clean pool: nothing raised, flags drained
    assert not kd._NAN_FLAGS
    p[77, 5] = float("nan")
    kd._hip_local_topk(q, p.to(DEV), 8, 0)
    with pytest.raises(ops.KnnSvcError):
        kd.raise_if_any_nan()
    assert not kd._NAN_FLAGS


def test_prematch_cli_entry_point(tmp_path, monkeypatch):
    """`python ddsp_prematch_dataset.py --librispeech_path ... --out_path ... --prematch` (the reference's command line,
    ddsp_prematch_dataset.py:1815-1831) end to end with seeded WavLM-Large weights: files of both speakers appear with the
    reference's names and keys, a second invocation with --save_pool_only adds the pool_f0 / pool_spec arrays."""
    import pickle
    from knn_svc_amd import matching, pool_cache, prematch
    monkeypatch.setenv("KNNSVC_SEEDED_WEIGHTS", "1")
    matching._POOL_CACHE = pool_cache.PoolCache()
    root = tmp_path / "train"
    for s, spk in enumerate(("singerA", "singerB")):
        (root / spk).mkdir(parents=True)
        for u in range(3):
            w, f = S.synth_clip(16000 + 640 * (s + u), seed=700 + 10 * s + u)
            audio_io.write_wav_pcm16(str(root / spk / f"u{u}.wav"), w, 16000)
            np.save(root / spk / f"u{u}_f0.npy", f)
    out = tmp_path / "cached"
    assert prematch.main(["--librispeech_path", str(root), "--out_path", str(out), "--topk", "4", "--matching_layer", "6",
                          "--synthesis_layer", "6", "--prematch"]) == 0
    for spk in ("singerA", "singerB"):
        pool = np.load(out / spk / "pool.npy")
        harm = np.load(out / spk / "pool_harmonics.npy")
        assert pool.shape[1] == 1024 and harm.shape == (pool.shape[0], 49)
        end = 0
        for u in range(3):
            with open(out / spk / f"u{u}.pt", "rb") as fh:
                d = pickle.load(fh)
            assert d["slice"][0] == end and d["nearest_nbrs"].shape == (d["slice"][1] - d["slice"][0], 32)
            assert d["amp_ratio"].shape == d["harmonics_best_weight_para"].shape == (d["nearest_nbrs"].shape[0], 4)
            end = d["slice"][1]
        assert end == pool.shape[0]
    assert prematch.main(["--librispeech_path", str(root), "--out_path", str(out), "--save_pool_only"]) == 0
    assert (out / "singerA" / "pool_f0.npy").is_file() and np.load(out / "singerB" / "pool_spec.npy").shape[1] == 200
    matching._POOL_CACHE = None


def test_released_checkpoints_reproduce_the_reference_sample_output(tmp_path):
    """BASELINE cfg 2 with the RELEASED weights, when they are present: the reference's own sample pair (60 s each, with its
    harvest f0 caches) through ddsp_inference's single-file path, against the output the reference ships for exactly that
    call (sample_content/..._knn_mix_post_opt_0.2.wav, PCM_32).  Bar: 1e-4 RMS (north_star).  The checkpoints cannot be
    fetched offline: looked for in $KNNSVC_CKPT_DIR (a '*mix*' generator, {'generator': state_dict}, ddsp_hubconf.py:85-94)
    and $KNNSVC_WAVLM_PT or the torch.hub cache (WavLM-Large.pt, {'cfg','model'}, :113-123); skipped otherwise."""
    import shutil
    from pathlib import Path
    from knn_svc_amd import hubconf
    ckpt_dir = os.environ.get("KNNSVC_CKPT_DIR", hubconf.DEFAULT_CKPT_DIR)
    hub_pt = os.path.join(torch.hub.get_dir(), "checkpoints", "WavLM-Large.pt")
    wavlm_pt = os.environ.get("KNNSVC_WAVLM_PT") or (hub_pt if os.path.isfile(hub_pt) else None)
    if not (os.path.isdir(ckpt_dir) and hubconf.scan_checkpoint(ckpt_dir, "mix") and wavlm_pt):
        pytest.skip("released checkpoints not present (WavLM-Large.pt + a *mix* generator): nothing to compare with")
    fx = Path(__file__).parent / "golden" / "sample_content_full"
    src, tgt = "Danakil-voice_resampled_16000_cut", "Tiken_lead_07_resampled_16000_cut"
    for f in fx.iterdir():
        if "_to_" not in f.name:
            shutil.copy(f, tmp_path / f.name)
    os.environ["KNNSVC_WAVLM_PT"] = wavlm_pt
    knn = hubconf.knn_vc(ckpt_type="mix", device="cuda", local_ckpt_dir=ckpt_dir, weights="released")
    y = knn.special_match(str(tmp_path / f"{src}.wav"), str(tmp_path / f"{tgt}.wav"), ckpt_type="mix", post_opt="post_opt_0.2")
    ref, sr = audio_io.read_wav(str(fx / f"{src}_to_{tgt}_knn_mix_post_opt_0.2.wav"))
    assert sr == 16000 and y.numel() == ref.shape[1] == 960320
    r = _rms(y, ref[0])
    print(f"released checkpoints, cfg 2 sample pair: waveform rms error {r:.2e}")
    assert r < 1e-4
