"""GPU: parity against the CPU oracle at FULL dimensions AND FULL length — the shapes bench.py runs
(one 30 s chunk = T 1500 frames), not short stand-ins: the multi-key-tile path of the attention kernel, the
>= 800-distance buckets of the relative position bias, the last-tile masks, the fp64 phase prefix of the additive
synthesiser over 1500 frames, the generator's long time axes.  Reference: wavlm/modules.py:457-564,
wavlm/WavLM.py:572-714, ddsp_prematch_dataset.py:165-208, hifigan/ddsp_models.py:176-233."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from knn_svc_amd import config as C, synthetic as S

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _stats(out, ref):
    d = (out.detach().cpu().double() - ref.detach().cpu().double())
    return float(d.abs().max()), float(d.pow(2).mean().sqrt()), float(ref.abs().max()), float(ref.double().pow(2).mean().sqrt())


def test_wavlm_large_six_layers_one_full_chunk_vs_oracle():
    """WavLM-Large, 6 layers, ONE 30 s chunk (480 320 padded samples -> T = 1500) against the oracle."""
    from knn_svc_amd.wavlm import WavLMEncoder
    from oracle import wavlm_ref
    cfg = C.WAVLM_LARGE
    sd = S.seeded_state(S.wavlm_param_spec(cfg, 6), seed=1)
    w, _ = S.synth_clip(30 * 16000, 31)
    x = torch.from_numpy(np.pad(w, (0, 320)))[None]
    ref = wavlm_ref.extract_layer(sd, cfg, x, 6)
    assert ref.shape == (1, 1500, 1024)
    ref64 = wavlm_ref.extract_layer({k: v.double() for k, v in sd.items()}, cfg, x.double(), 6)     # yardstick: exact arithmetic
    out = WavLMEncoder(sd, cfg, DEV, n_layers=6).encode_batch(x.to(DEV))
    mx, rms, rmax, rrms = _stats(out, ref)
    mx64, rms64, _, _ = _stats(out, ref64)
    cmx, crms, _, _ = _stats(ref, ref64)
    print(f"WavLM-Large 6 layers, T=1500: vs oracle max|d| {mx:.2e} rms {rms:.2e}; vs fp64 max {mx64:.2e} rms {rms64:.2e} "
          f"(oracle vs fp64: max {cmx:.2e} rms {crms:.2e}; ref max {rmax:.2f} rms {rrms:.3f})")
    assert mx < 5e-5 * max(1.0, rmax) and rms < 2.5e-6 * max(1.0, rrms)
    # distance from exact arithmetic: measured 2.2 x the CPU's own fp32 evaluation (3.5e-6 vs 1.6e-6 at rms 2.8).  It
    # is set by the conv stack's GEMMs: an MFMA chain adds K/16 x 3 partial sums one after the other in fp32 (288 roundings
    # at K = 1536: 4.4e-7 per layer), MKL's blocked 16-lane partial sums are shorter chains (1.7e-7) — tools/layer_error.py
    assert rms64 <= 2.5 * crms + 1e-7
    # the features feed a cosine kNN: per-frame direction error
    cos = F.cosine_similarity(out[0].cpu().double(), ref[0].double(), dim=1)
    print(f"  min per-frame cosine {float(cos.min()):.12f}")
    assert float((1 - cos).max()) < 1e-10


def test_wavlm_large_six_layers_one_full_chunk_golden(golden):
    """Fixture G1d: the REFERENCE's WavLM-Large (six layers, seeded weights) on one full 30 s chunk — every 25th frame of the exit
    layer and all 1500 frame norms, against the GPU encoder."""
    from knn_svc_amd.wavlm import WavLMEncoder
    g = golden("g1d_wavlm_large6_full_chunk")
    sd = S.seeded_state(S.wavlm_param_spec(C.WAVLM_LARGE, 6), int(g["seed"]))
    w, _ = S.synth_clip(30 * 16000, int(g["clip_seed"]))
    x = torch.from_numpy(np.pad(w, (0, 320)))[None]
    out = WavLMEncoder(sd, C.WAVLM_LARGE, DEV, n_layers=6).encode_batch(x.to(DEV))[0].cpu()
    assert out.shape == (1500, 1024)
    ref = torch.from_numpy(g["rows"])
    mx, rms, rmax, rrms = _stats(out[::25], ref)
    nd = float((out.norm(dim=1) - torch.from_numpy(g["norms"])).abs().max())
    cos = F.cosine_similarity(out[::25].double(), ref.double(), dim=1)
    print(f"WavLM-Large 6 layers, T=1500 vs the reference's rows: max|d| {mx:.2e} rms {rms:.2e} (ref max {rmax:.2f} rms {rrms:.3f}); "
          f"largest frame-norm difference {nd:.2e}; min per-frame cosine {float(cos.min()):.12f}")
    assert mx < 5e-5 * max(1.0, rmax) and rms < 2.5e-6 * max(1.0, rrms)
    assert nd < 2e-4 and float((1 - cos).max()) < 1e-10


def test_attention_full_length_vs_oracle():
    """attention2_kernel at E=1024 / H=16 / T=1500 (12 query blocks x many key tiles per head, distances beyond the
    last log bucket, ragged last tile since 1500 % 128 != 0) vs F.scaled_dot_product_attention with the gated bias."""
    from knn_svc_amd import ops
    from oracle import wavlm_ref
    cfg = dict(C.WAVLM_LARGE, encoder_layers=1)
    H, E, T, B = 16, 1024, 1500, 1
    sd = S.seeded_state([s for s in S.wavlm_param_spec(cfg) if s[0].startswith("encoder.layers.0.self_attn")], 3)
    g = torch.Generator().manual_seed(2)
    xn = torch.randn(T, B, E, generator=g)
    p = "encoder.layers.0.self_attn."
    gate_ref = wavlm_ref.gate(sd, cfg, 0, xn)
    w8, b8 = sd[p + "grep_linear.weight"], sd[p + "grep_linear.bias"]
    w2 = torch.stack([w8[:4].sum(0), w8[4:].sum(0)]).contiguous()
    b2 = torch.stack([b8[:4].sum(), b8[4:].sum()])
    xbt = xn.transpose(0, 1).reshape(B * T, E).contiguous().to(DEV)
    gate = ops.wavlm_gate(xbt, H, w2.to(DEV), b2.to(DEV), sd[p + "grep_a"].reshape(-1).to(DEV))
    pb = wavlm_ref.position_bias(sd, cfg, T)
    # scale the projections up so that the scores are not flat (seeded N(0, 0.02) weights give near-uniform softmax)
    q = 6.0 * F.linear(xn, sd[p + "q_proj.weight"], sd[p + "q_proj.bias"])
    k = 6.0 * F.linear(xn, sd[p + "k_proj.weight"], sd[p + "k_proj.bias"])
    v = F.linear(xn, sd[p + "v_proj.weight"], sd[p + "v_proj.bias"])
    sh = lambda t: t.reshape(T, B * H, 64).transpose(0, 1).reshape(B, H, T, 64)
    ref = F.scaled_dot_product_attention(sh(q), sh(k), sh(v), attn_mask=gate_ref * pb[None])
    ref = ref.permute(0, 2, 1, 3).reshape(B * T, E)
    # yardstick: the same attention in fp64 — how far is the reference's own fp32 arithmetic from exact?
    ref64 = F.scaled_dot_product_attention(sh(q).double(), sh(k).double(), sh(v).double(), attn_mask=(gate_ref * pb[None]).double())
    ref64 = ref64.permute(0, 2, 1, 3).reshape(B * T, E)
    lut = wavlm_ref.rel_bucket_table(T, 320, 800)
    assert int(lut.max()) == 319 and int(lut.min()) == 0                      # both saturated log buckets are reached
    table = sd[p + "relative_attention_bias.weight"][lut].T.contiguous()
    qkv = torch.cat([q, k, v], -1).transpose(0, 1).reshape(B * T, 3 * E).contiguous().to(DEV)
    out = ops.wavlm_attention(qkv, gate, table.to(DEV), B, T, H)
    mx, rms, rmax, rrms = _stats(out, ref)
    mx64, rms64, _, _ = _stats(out, ref64)
    cmx, crms, _, _ = _stats(ref, ref64)
    print(f"attention T=1500: vs oracle max|d| {mx:.2e} rms {rms:.2e}; vs fp64 max {mx64:.2e} rms {rms64:.2e} "
          f"(oracle vs fp64: max {cmx:.2e} rms {crms:.2e}; ref max {rmax:.3f} rms {rrms:.4f})")
    # as close to exact arithmetic as the reference's own fp32 evaluation (x 1.5 margin), and close to the oracle
    assert rms64 <= 1.5 * crms + 1e-8 and mx64 <= 1.5 * cmx + 1e-7
    assert rms < 5e-6 * max(1.0, rrms) and mx < 5e-5 * max(1.0, rmax)
    # first and last query rows see the extreme relative distances
    for rows in (slice(0, 4), slice(T - 4, T)):
        assert float((out[rows].cpu().double() - ref64[rows]).abs().max()) <= 1.5 * cmx + 1e-7


@pytest.mark.parametrize("kind", ["mix", "f0"])
def test_vocoder_full_size_300_frames_vs_oracle(kind):
    """Full-size generator, both variants — 'mix' (hifigan/ddsp_models.py:405-493: harmonic excitation, 22.9 M parameters) and the
    f0-only one the wavlm_only checkpoints use (hifigan/ddsp_models_f0.py:106-216, 320-381: sine excitation, BASELINE cfg 1) —,
    300 frames = 6 s = 96 000 samples, vs the CPU oracle."""
    from knn_svc_amd.vocoder import Vocoder
    from oracle import vocoder_ref
    h = C.HIFIGAN_V1
    sd = S.seeded_state(S.generator_param_spec(h, kind), 2)
    g = torch.Generator().manual_seed(3)
    N = 300
    c = torch.randn(N, 1024, generator=g)
    _, f0 = S.synth_clip(N * 320, 5); f0 = torch.from_numpy(f0[:N].copy())
    harm = torch.rand(N, 49, generator=g) * 0.02 if kind == "mix" else None
    ref = vocoder_ref.synthesizer(sd, h, kind, c[None], f0[None, :, None], None if harm is None else harm[None])[0, 0]
    y = Vocoder(sd, h, kind, DEV).forward(c.to(DEV), f0.to(DEV), None if harm is None else harm.to(DEV))
    mx, rms, rmax, rrms = _stats(y, ref)
    print(f"full generator ({kind}), 300 frames: rms {rms:.2e} max|d| {mx:.2e} (ref rms {rrms:.3f})")
    assert y.numel() == N * 320 and rms < 2e-6 and mx < 5e-5      # north_star bar: 1e-4 RMS


def test_additive_synth_1500_frames_gliding_f0_vs_oracle():
    """get_bulk_dsp_choral over a whole 30 s chunk with a gliding f0 and unvoiced gaps: the fp64 phase prefix runs
    over 480 000 samples (DESIGN 'known sensitivity'), harmonics cross Nyquist as f0 rises."""
    from knn_svc_amd import ops
    from oracle import synth_ref
    N = 1500
    _, f0 = S.synth_clip(N * 320, 77)
    f0 = torch.from_numpy(f0[:N].copy())
    f0 = torch.where(f0 > 0, f0 * (1.0 + 0.9 * torch.linspace(0, 1, N)), f0)        # glide up to ~800 Hz: k*f0 crosses 8 kHz
    g = torch.Generator().manual_seed(7)
    amp = torch.rand(N, 49, generator=g) * 0.05
    amp[::7] *= -0.3                                                                  # bicubic overshoot region / sign changes
    ref = synth_ref.additive_synth(f0[None, :, None], amp[None])[0, :, 0]
    pw = torch.randn(32, 3, generator=g); pb = torch.randn(32, generator=g)
    cond = torch.empty(N * 320, 32, device=DEV)
    exc = ops.additive_synth(f0.to(DEV), amp.to(DEV), pw.to(DEV), pb.to(DEV), cond, 32, want_exc=True)
    mx, rms, rmax, rrms = _stats(exc, ref)
    print(f"additive synth 1500 frames: max|d| {mx:.2e} rms {rms:.2e} (ref max {rmax:.3f})")
    dd = (exc.cpu() - ref).abs()
    worst = torch.topk(dd, 6).indices
    print("  worst samples:", [(int(i), int(i) // 320, round(float(f0[int(i) // 320]), 2), f"{float(dd[i]):.1e}") for i in worst],
          "samples above 1e-6:", int((dd > 1e-6).sum()))
    assert mx < 2e-5 and rms < 2e-6
    # the error must not grow along the phase integrator: last second as good as the first
    d = (exc.cpu() - ref).abs()
    assert float(d[-16000:].max()) < 2e-5 and float(d[:16000].max()) < 2e-5
    ref_cond = F.conv1d(ref[None, None], pw[:, None, :], pb, padding=1)[0].T
    assert float((cond.cpu() - ref_cond).abs().max()) < 1e-4


def test_wavlm_bucketed_ragged_chunks_equal_exact_length():
    """Dataset mode: utterances of different lengths share ONE (batch, bucket) shape — each is zero-padded up to its
    bucket and masked with WavLM's own padding-mask semantics (wavlm/WavLM.py:311-321, 353, 574-575: padded frames are
    zeroed in front of the positional conv and excluded as attention keys).  Rows below a chunk's own frame count must be
    BIT-IDENTICAL to encoding the chunk alone at its exact length, and match the oracle."""
    from knn_svc_amd.wavlm import WavLMEncoder
    from oracle import wavlm_ref
    cfg = C.WAVLM_LARGE
    sd = S.seeded_state(S.wavlm_param_spec(cfg, 3), seed=1)
    enc = WavLMEncoder(sd, cfg, DEV, n_layers=3)
    lens = [81700, 88000, 94401, 95999, 321, 16000 * 30 + 5000]        # 255 / 275 / 295 / 299 frames: one 300-frame bucket
    wavs = [torch.from_numpy(S.synth_clip(n, 40 + i)[0]).to(DEV) for i, n in enumerate(lens)]
    got = enc.encode_many(wavs, pow2_batches=True)
    # the 5.x s utterances fall into the same 300-frame bucket and were encoded as one masked batch
    frames = [enc.n_frames(l + 320 - l % 320) if l <= 480000 else None for l in lens]
    assert len({enc.bucket_frames(f) for f in frames[:4]}) == 1 and frames[0] != frames[2]
    enc.use_graphs = False
    for w, out in zip(wavs, got):
        # exact-length reference path: every chunk alone, its own padded length, no mask
        parts = []
        from knn_svc_amd.wavlm import chunk_plan
        for (s_, l, p) in chunk_plan(w.numel()):
            buf = torch.zeros(1, l + p, device=DEV); buf[0, :l] = w[s_:s_ + l]
            parts.append(enc.encode_batch(buf)[0])
        exact = torch.cat(parts, 0)
        assert out.shape == exact.shape and torch.equal(out, exact), (w.numel(), float((out - exact).abs().max()))
    ref = wavlm_ref.full_features(sd, cfg, wavs[1].cpu(), 3)
    assert float((got[1].cpu() - ref).abs().max()) < 1e-4 * float(ref.abs().max())
    # and through the graphs: second sight captures, third replays — same bits
    enc.use_graphs = True
    again = [enc.encode_many(wavs, pow2_batches=True) for _ in range(3)][-1]
    assert all(torch.equal(a, b) for a, b in zip(again, got))
    assert any(k[2] for k in enc._graphs) and len(enc._graphs) <= 6          # masked buckets were captured; few shapes


def test_vocoder_bucket_graph_equals_exact_length():
    """Generator frame-count buckets: one hipGraph per 25-frame bucket; every launch inside takes its lengths from a
    device-side frame count (rows past the valid length read as the convolutions' zero padding).  Different lengths of one
    bucket through the same graph must equal the eager exact-length run bit for bit, and match the oracle."""
    from knn_svc_amd.vocoder import Vocoder
    from oracle import vocoder_ref
    for kind, hcfg in (("mix", C.HIFIGAN_V1), ("f0", C.HIFIGAN_TINY)):
        sd = S.seeded_state(S.generator_param_spec(hcfg, kind), 2)
        voc = Vocoder(sd, hcfg, kind, DEV)
        g = torch.Generator().manual_seed(3)
        hub = hcfg.get("hubert_dim", 1024)
        outs = {}
        for N in (103, 111, 125, 103, 111, 125, 101):          # one bucket (125): eager first sights, capture, replays
            c = torch.randn(N, hub, generator=torch.Generator().manual_seed(N)).to(DEV)
            _, f0 = S.synth_clip(N * 320, N); f0 = torch.from_numpy(f0[:N].copy()).to(DEV)
            harm = (torch.rand(N, 49, generator=torch.Generator().manual_seed(N + 1)) * 0.02).to(DEV) if kind == "mix" else None
            y = voc.forward(c, f0, harm)
            voc.use_graphs = False
            ye = voc.forward(c, f0, harm)
            voc.use_graphs = True
            assert y.numel() == N * 320 and torch.equal(y, ye), (kind, N, float((y - ye).abs().max()))
            outs[N] = (c, f0, harm, y)
        assert list(voc._graphs) == [125]
        if kind == "mix":
            c, f0, harm, y = outs[111]
            ref = vocoder_ref.synthesizer(sd, hcfg, kind, c.cpu()[None], f0.cpu()[None, :, None], harm.cpu()[None])[0, 0]
            assert float((y.cpu() - ref).pow(2).mean().sqrt()) < 2e-6


@pytest.mark.parametrize("ckpt_type,post_opt,src_seconds", [("mix", "post_opt_0.2", 30), ("wavlm_only", "no_post_opt", 60)] +
                         ([("mix", "post_opt_0.2", 60)] if __import__("os").environ.get("KNNSVC_SLOW_TESTS") == "1" else []))
def test_cfg1_cfg2_sample_pair_at_full_size_vs_oracle(tmp_path, monkeypatch, ckpt_type, post_opt, src_seconds):
    """BASELINE cfg 2 (ckpt_type=mix, post_opt_0.2) and cfg 1's arguments (wavlm_only, no_post_opt; its `--device cpu` is the
    oracle's side here) on the reference's own 60 s sample pair (Danakil -> Tiken, 3001 x 3001 frames,
    ddsp_matcher.py:937-1023 special_match): WavLM-Large (6 layers) and the FULL generators — the additive-synth 'mix' one and
    the sine-excitation 'f0' one (hifigan/ddsp_models_f0.py) — with seeded weights, through the product entry points
    (hubconf.knn_vc + KNeighborsVC.special_match, files in -> file out).  The CPU oracle is given the GPU's features of both
    clips (the encoder has its own full-size oracle tests) and restates everything behind them: 3001 x 3001 search, f0 shift,
    re-rank, both concat re-selections and both Adam loops (mix) or the uniform weights (wavlm_only), weighted sums, additive
    synth / sine source, generator.  Waveform within the north-star tolerance (1e-4 RMS).
    The 'mix' case runs the first 30 s of the source (1500 frames) against the FULL 60 s pool by default: the oracle's two
    frame-sequential re-selections and two Adam loops over 3001 frames take 2.5 minutes of host time; KNNSVC_SLOW_TESTS=1 adds the
    full 60 s source (measured in round 5: rms 2.5e-5).  The 'wavlm_only' case is the full 3001 x 3001."""
    import shutil
    from pathlib import Path
    from knn_svc_amd import audio_io, hubconf, matching
    from oracle import pipeline_ref, vocoder_ref
    fx = Path(__file__).parent / "golden" / "sample_content_full"
    src, tgt = "Danakil-voice_resampled_16000_cut", "Tiken_lead_07_resampled_16000_cut"
    for name in (src, tgt):
        shutil.copy(fx / f"{name}.wav", tmp_path / f"{name}.wav")
        shutil.copy(fx / f"{name}_f0.npy", tmp_path / f"{name}_f0.npy")
    n_frames = 3001
    if src_seconds < 60:                                              # the head of the source clip and of its f0 track
        x, sr = audio_io.read_wav(str(fx / f"{src}.wav"))
        audio_io.write_wav_pcm16(str(tmp_path / f"{src}.wav"), x[0][:src_seconds * 16000], sr)
        np.save(tmp_path / f"{src}_f0.npy", np.load(fx / f"{src}_f0.npy")[:src_seconds * 50 + 1])
        n_frames = src_seconds * 50                                   # 480 000 samples -> one padded 30 s chunk -> 1500 frames
    monkeypatch.setenv("KNNSVC_SEEDED_WEIGHTS", "1")
    matching._POOL_CACHE = None
    knn = hubconf.knn_vc(ckpt_type=ckpt_type, device="cuda", weights="seeded")
    y = knn.special_match(str(tmp_path / f"{src}.wav"), str(tmp_path / f"{tgt}.wav"), ckpt_type=ckpt_type, post_opt=post_opt).cpu()
    out_file = tmp_path / f"{src}_to_{tgt}_knn_{ckpt_type}_{post_opt}.wav"
    assert out_file.is_file() and y.numel() == n_frames * 320
    # the GPU's features of both clips (the pool store still holds them: nothing is encoded twice)
    qp, _s, _a, _sp, qf0, _qh = matching.get_complete_spk_pool(tmp_path / f"{src}.wav", knn.wavlm, device=DEV)
    pp, _s, _a, _sp, pf0, ph = matching.get_complete_spk_pool(tmp_path / f"{tgt}.wav", knn.wavlm, device=DEV)
    c = lambda d: torch.cat(list(d.values())).cpu()
    query, pool = dict(feats=c(qp), f0=c(qf0)), dict(feats=c(pp), f0=c(pf0), harm=c(ph))
    assert query["feats"].shape == (n_frames, 1024) and pool["feats"].shape == (3001, 1024)
    f0only = "wavlm_only" in ckpt_type
    of, hw, sf0 = pipeline_ref.match(query, pool, ckpt_type, "no_post_opt" if f0only else post_opt)[:3]
    kind = hubconf.generator_kind(ckpt_type)
    sdg = S.seeded_state(S.generator_param_spec(C.HIFIGAN_V1, kind), seed=2)
    ref = vocoder_ref.synthesizer(sdg, C.HIFIGAN_V1, kind, of[None], sf0[None, :, None], None if f0only else hw[None]).reshape(-1)
    rms = float((y.double() - ref.double()).pow(2).mean().sqrt())
    sig = float(ref.double().pow(2).mean().sqrt())
    print(f"sample pair ({src_seconds} s of the source vs the 60 s pool), {ckpt_type} / {post_opt}: waveform rms vs oracle {rms:.2e} "
          f"(signal rms {sig:.3f}, {kind} generator, {n_frames} frames)")
    assert rms < 1e-4, rms
    x, sr = audio_io.read_wav(str(out_file))                          # the file the call wrote: save_audio's scaling of the same samples
    yy = y.numpy().astype(np.float64)
    assert sr == 16000 and np.abs(x[0].astype(np.float64) - yy / max(1.0, float(np.abs(yy).max()))).max() < 1e-6
    matching._POOL_CACHE = None
