"""GPU: the f16x2 (fp32-on-fp16-matrix-cores) path under activation ranges the seeded N(0, s) weights never produce —
outlier channels (x 10^3..10^4), tiny-rms tensors, weights that push LayerNorm / FFN / attention / generator
activations past the fixed-scale limit of 4094.  Nothing here may come out NaN or inaccurate, and nothing needs a manual
re-run in another mode: operand scales are derived on the device from range slots (knnsvc_absmax, out_absmax -> x_absmax),
the encoder's split-layout plan is decided from analytic bounds of the weights, and out-of-range attention layers run the
bf16x3 kernel.  Released WavLM-Large / generator checkpoints are known for outlier channels."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from knn_svc_amd import config as C, synthetic as S

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _rel(out, ref):
    out, ref = out.detach().cpu().double(), ref.detach().cpu().double()
    return float((out - ref).pow(2).mean().sqrt() / (ref.pow(2).mean().sqrt() + 1e-300)), float((out - ref).abs().max() / (ref.abs().max() + 1e-300))


def _heavy(M, K, seed, outliers=(3, 500, 901), mag=2.0e4):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(M, K, generator=g)
    for j, c in enumerate(outliers):
        x[:, c % K] *= mag / (1 + j)                 # a few channels 10^3..10^4 times larger than the rest
    return x


@pytest.mark.parametrize("M,K,N", [(1500, 1024, 1024), (300, 4096, 1024), (129, 64, 130)])
def test_linear_outlier_channels_with_range_slot(M, K, N):
    from knn_svc_amd import ops
    x = _heavy(M, K, 1)
    g = torch.Generator().manual_seed(2)
    w = torch.randn(N, K, generator=g) * 0.03
    b = torch.randn(N, generator=g)
    ref = x.double() @ w.double().T + b.double()
    xd, wd, bd = x.to(DEV), ops.attach_split(w.to(DEV)), b.to(DEV)
    assert float(x.abs().max()) > 4094 * 4                        # far outside the fixed-scale range
    slot = ops.absmax(xd)
    assert float(slot.max()) == float(x.abs().max())
    out_slot = ops.new_slot(DEV)
    out = ops.linear(xd, wd, bd, x_absmax=slot, out_absmax=out_slot)
    rms, mx = _rel(out, ref)
    print(f"linear with outlier channels ({M}x{K}x{N}): rel rms {rms:.2e} max {mx:.2e}")
    assert bool(torch.isfinite(out).all()) and rms < 1.5e-6 and mx < 5e-6      # fp32-GEMM accuracy (fp32 accumulation over K)
    # the epilogue's fold of max|out| is a bound of what was stored (rows past M may add |bias|: still a bound)
    true_max = float(out.abs().max())
    assert true_max <= float(out_slot.max()) <= max(true_max, float(b.abs().max())) * (1 + 1e-6)
    # the fixed default scale cannot hold these activations: documented as loud (NaN), never silently wrong
    bad = ops.linear(xd, wd, bd)
    assert not bool(torch.isfinite(bad).all())


@pytest.mark.parametrize("scale", [1e-6, 3e-3, 7e4])
def test_linear_tiny_and_huge_rms_with_range_slot(scale):
    from knn_svc_amd import ops
    g = torch.Generator().manual_seed(5)
    x = torch.randn(700, 1024, generator=g) * scale
    w = torch.randn(512, 1024, generator=g) * 0.05
    ref = x.double() @ w.double().T
    out = ops.linear(x.to(DEV), ops.attach_split(w.to(DEV)), x_absmax=ops.absmax(x.to(DEV)))
    rms, mx = _rel(out, ref)
    print(f"linear at activation scale {scale:g}: rel rms {rms:.2e} max {mx:.2e}")
    assert rms < 6e-7 and mx < 3e-6                    # the same error at every scale: the operand scale follows the data


def test_range_slot_results_do_not_depend_on_the_bound():
    """Power-of-two operand scales are exact wherever both fp16 pieces of an element stay normal; only the low pieces of
    very small elements (fp16 subnormals) depend on the scale.  A tight slot and a loose bound (x 8) therefore agree to
    ~1e-7 of the output range, and the fixed scale 16 is never closer to fp64 than the slot-driven scale."""
    from knn_svc_amd import ops
    g = torch.Generator().manual_seed(9)
    x = torch.randn(1000, 1024, generator=g)
    w = torch.randn(768, 1024, generator=g) * 0.02
    ref = x.double() @ w.double().T
    xd, wd = x.to(DEV), ops.attach_split(w.to(DEV))
    a = ops.linear(xd, wd)
    b = ops.linear(xd, wd, x_absmax=ops.absmax(xd))
    c = ops.linear(xd, wd, x_absmax=torch.full((ops.SLOT_W,), 8.0 * float(x.abs().max()), device=DEV))
    assert float((b - c).abs().max()) < 2e-7 * float(b.abs().max())
    ea, eb = _rel(a, ref)[0], _rel(b, ref)[0]
    print(f"rel rms vs fp64: fixed scale 16 {ea:.3e}, slot-driven {eb:.3e}")
    assert eb <= ea * 1.02 and float((a - b).abs().max()) < 1e-5


@pytest.mark.parametrize("fs", [2.0 ** 15, 2.0 ** -15, 300.0])
def test_knn_feature_scale_invariance(fs):
    """WavLM features have no a-priori range: the kNN picks its operand scales from max|q| / max|pool| on the device."""
    from knn_svc_amd import ops
    from oracle import knn_ref
    q = S.clustered_features(200, 1024, 41) * fs
    p = S.clustered_features(4096, 1024, 42) * fs
    idx, dist = ops.knn_topk(q.to(DEV), p.to(DEV), 32)
    ref_i, ref_d = knn_ref.knn_topk(q, p, 32)
    st = knn_ref.topk_agreement(ref_i, idx.cpu(), knn_ref.cosine_dist_f64(q, p), tau=5e-7)
    print(f"kNN at feature scale {fs:g}:", st)
    assert st["unexplained"] == 0 and st["top4"] >= 0.99 and float((dist.cpu() - ref_d).abs().max()) < 5e-6
    if fs in (2.0 ** 15, 2.0 ** -15):          # a power-of-two rescaling changes no bit of the dots, norms or distances
        i1, d1 = ops.knn_topk((q / fs).to(DEV), (p / fs).to(DEV), 32)
        assert torch.equal(i1, idx) and torch.equal(d1, dist)


def _scaled_state(sd, factors):
    sd = {k: v.clone() for k, v in sd.items()}
    for k, f in factors.items():
        assert k in sd, k
        sd[k] = sd[k] * f
    return sd


def test_wavlm_large_with_outlier_weights_takes_the_wide_paths():
    """LayerNorm gains / FFN / q-k projections scaled so that layer 1's FFN hidden and layer 0's keys exceed 4094: the
    range plan must route those tensors around the fixed-scale split layout, and the result must still match the oracle."""
    from knn_svc_amd.wavlm import WavLMEncoder
    from oracle import wavlm_ref
    cfg = C.WAVLM_LARGE
    base = S.seeded_state(S.wavlm_param_spec(cfg, 3), seed=1)
    sd = _scaled_state(base, {
        "encoder.layers.1.final_layer_norm.weight": 60.0, "encoder.layers.1.fc1.weight": 40.0,     # FFN hidden up to ~7000
        "encoder.layers.1.fc2.weight": 1.0 / 2400.0,                                                # keep the stream sane
        # (x 200, not more: q, k and v share ONE fused weight matrix and its split uses one per-tensor power-of-two scale, so rows
        #  more than ~2^15 smaller than the largest row lose their low fp16 piece — DESIGN.md "known limits")
        "encoder.layers.0.self_attn.k_proj.weight": 200.0, "encoder.layers.0.self_attn.q_proj.weight": 1.0 / 200.0,     # key bound 10 000
        "encoder.layers.0.self_attn.q_proj.bias": 1.0 / 200.0,
        "encoder.layers.2.self_attn_layer_norm.weight": 1500.0, "encoder.layers.2.self_attn.v_proj.weight": 1.0 / 1500.0,   # LN output up to ~6000
        "encoder.layers.2.self_attn.q_proj.weight": 1.0 / 1500.0, "encoder.layers.2.self_attn.k_proj.weight": 1.0 / 1500.0,
    })
    w0, _ = S.synth_clip(32000, 11)
    x = torch.from_numpy(np.pad(w0, (0, 320)))[None]
    ref = wavlm_ref.extract_layer(sd, cfg, x, 3)
    ref64 = wavlm_ref.extract_layer({k: v.double() for k, v in sd.items()}, cfg, x.double(), 3)
    enc = WavLMEncoder(sd, cfg, DEV, n_layers=3)
    pl = enc.plan["layers"]
    print("range plan:", pl, {k: (v if not isinstance(v, dict) else {a: round(b, 1) for a, b in v.items()})
                              for k, v in enc.plan["bounds"].items() if k.startswith("layer")})
    assert not pl[1]["h"] and not pl[0]["attn_f16"] and not pl[2]["xn"]
    assert pl[0]["xn"] and pl[0]["h"] and pl[1]["xn2"] and pl[2]["h"] and pl[1]["attn_f16"]   # untouched tensors keep the split layout
    out = enc.encode_batch(x.to(DEV))
    assert bool(torch.isfinite(out).all())
    rms, mx = _rel(out, ref)
    rms64, mx64 = _rel(out, ref64)
    crms, cmx = _rel(ref, ref64)
    print(f"WavLM-Large with outlier weights: vs oracle rel rms {rms:.2e} max {mx:.2e}; vs fp64 {rms64:.2e} / {mx64:.2e} "
          f"(oracle vs fp64 {crms:.2e} / {cmx:.2e})")
    assert rms64 <= 3.0 * crms + 1e-7 and rms < 2e-5            # within 3 x of the reference's own fp32 distance from exact
    # the seeded state itself plans "everything split" (the fast path is what the other tests and bench.py run)
    enc0 = WavLMEncoder(base, cfg, DEV, n_layers=3)
    assert all(all(v.values()) for v in enc0.plan["layers"]) and all(enc0.plan["conv"]) and enc0.plan["feats"]


@pytest.mark.parametrize("gain", [3000.0, 1e-4])
def test_generator_with_large_and_tiny_activations(gain):
    """Full-size 'mix' generator with every internal activation `gain` x larger / smaller: the network before tanh is
    positively homogeneous (convolutions + leaky ReLUs), so scaling every bias and the two entry layers (lin_pre,
    sin_prenet) by `gain` and conv_post by 1/gain leaves the waveform unchanged mathematically — while the GEMM inputs sit
    at ~10^3 x / 10^-4 x their usual magnitude (far beyond / below the fixed-scale window)."""
    from knn_svc_amd.vocoder import Vocoder
    from oracle import vocoder_ref
    h = C.HIFIGAN_V1
    base = S.seeded_state(S.generator_param_spec(h, "mix"), 2)
    factors = {k: gain for k in base if k.endswith(".bias")}
    factors.update({"dec.lin_pre.weight": gain, "sin_prenet.weight": gain, "dec.conv_post.weight": 1.0 / gain})
    sd = _scaled_state(base, factors)
    g = torch.Generator().manual_seed(3)
    N = 40
    c = torch.randn(N, 1024, generator=g) * 3.0
    _, f0 = S.synth_clip(N * 320, 5); f0 = torch.from_numpy(f0[:N].copy())
    harm = torch.rand(N, 49, generator=g) * 0.02
    ref0 = vocoder_ref.synthesizer(base, h, "mix", c[None], f0[None, :, None], harm[None])[0, 0]
    ref = vocoder_ref.synthesizer(sd, h, "mix", c[None], f0[None, :, None], harm[None])[0, 0]
    y = Vocoder(sd, h, "mix", DEV).forward(c.to(DEV), f0.to(DEV), harm.to(DEV))
    assert bool(torch.isfinite(y).all())
    rms = float((y.cpu() - ref).pow(2).mean().sqrt())
    rms0 = float((y.cpu() - ref0).pow(2).mean().sqrt())
    print(f"generator with activations x {gain:g}: rms vs oracle {rms:.2e}, vs the unscaled network {rms0:.2e} "
          f"(signal rms {float(ref.pow(2).mean().sqrt()):.3f})")
    assert rms < 2e-6 and rms0 < 2e-6                                     # north_star bar: 1e-4 RMS


def _heavy_tailed(sd, seed, frac=0.004, gain=12.0, skip=("bias", "layer_norm", "norm")):
    """Seeded weights with heavy tails: a fraction of every weight matrix's entries multiplied by +-`gain` (the Gaussian state
    has no entry beyond ~5 sigma; released checkpoints do)."""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for k, v in sd.items():
        v = v.clone()
        if v.dim() >= 2 and not any(s in k for s in skip):
            m = torch.rand(v.shape, generator=g) < frac
            v[m] = v[m] * gain
        out[k] = v
    return out


def test_e2e_full_architecture_with_heavy_tailed_weights_and_outlier_channels_vs_oracle(tmp_path):
    """VERDICT r3 do-this 8: every model test ran on seeded N(0, s) weights.  Here WavLM-Large (6 layers) and the full 'mix'
    generator carry heavy-tailed weights (0.4 % of every matrix x 12) AND outlier channels the way released WavLM checkpoints
    have them (two residual-stream channels driven ~50 x above the rest through fc2 / the LayerNorm gains that follow, and
    compensated downstream), files in -> waveform out (match_at_inference_time + vocode, mix, post_opt_0.2), against the CPU
    oracle's whole pipeline (encoder, pool, kNN, f0 shift / re-rank, both concat re-selections, both Adam loops, weighted sums,
    additive synth, generator) on the same files: north-star tolerance 1e-4 RMS."""
    from knn_svc_amd import audio_io
    from knn_svc_amd.matcher import KNeighborsVC
    from knn_svc_amd.matching import match_at_inference_time
    from knn_svc_amd.vocoder import Vocoder
    from knn_svc_amd.wavlm import WavLMEncoder
    from oracle import pipeline_ref
    cfg, h = C.WAVLM_LARGE, C.HIFIGAN_V1
    sdw = _heavy_tailed(S.seeded_state(S.wavlm_param_spec(cfg, 6), seed=1), seed=7)
    out_ch = (37, 611)                                   # outlier channels of the residual stream from layer 2 on
    for c in out_ch:
        sdw["encoder.layers.1.fc2.weight"][c] *= 50.0
        for l in range(2, 6):                            # the LayerNorms behind them keep the sub-layers' inputs sane
            sdw[f"encoder.layers.{l}.self_attn_layer_norm.weight"][c] /= 50.0
            sdw[f"encoder.layers.{l}.final_layer_norm.weight"][c] /= 50.0
    sdg = _heavy_tailed(S.seeded_state(S.generator_param_spec(h, "mix"), seed=2), seed=8, frac=0.002, gain=8.0, skip=("bias",))
    (tmp_path / "a").mkdir(); (tmp_path / "b").mkdir()
    src_w, src_f = S.synth_clip(3 * 16000 + 40, seed=31)
    audio_io.write_wav_pcm16(str(tmp_path / "a" / "src.wav"), src_w, 16000)
    np.save(tmp_path / "a" / "src_f0.npy", (src_f * 1.2).astype(np.float32))
    pool_w, pool_f = [], []
    for i in range(3):
        w, f = S.synth_clip(5 * 16000 + 7 * i, seed=40 + i)
        audio_io.write_wav_pcm16(str(tmp_path / "b" / f"u{i}.wav"), w, 16000)
        np.save(tmp_path / "b" / f"u{i}_f0.npy", f.astype(np.float32))
    srcp = str(tmp_path / "a" / "src.wav")
    enc = WavLMEncoder(sdw, cfg, DEV, n_layers=6)
    knn = KNeighborsVC(enc, Vocoder(sdg, h, "mix", DEV), h, DEV)
    of, hw, _a, sf0 = match_at_inference_time(srcp, tmp_path / "b", enc, knn.weighting, knn.weighting, prioritize_f0=True,
                                              ckpt_type="mix", post_opt="post_opt_0.2", tgt_dataset_path=tmp_path)
    y = knn.vocode(of[srcp][None], sf0[srcp][None, :, None], hw[srcp][None]).squeeze().cpu()
    assert bool(torch.isfinite(y).all())
    # the oracle reads what the product read: 16-bit files and the stored f0 tracks
    rd = lambda p: torch.from_numpy(audio_io.read_wav(str(p))[0][0].astype(np.float32))
    ref = pipeline_ref.convert(sdw, cfg, sdg, h, "mix", rd(tmp_path / "a" / "src.wav"), torch.from_numpy(np.load(tmp_path / "a" / "src_f0.npy")),
                               [rd(tmp_path / "b" / f"u{i}.wav") for i in range(3)],
                               [torch.from_numpy(np.load(tmp_path / "b" / f"u{i}_f0.npy")) for i in range(3)],
                               "mix", "post_opt_0.2", n_layers=6)
    assert y.shape == ref.shape
    rms = float((y.double() - ref.double()).pow(2).mean().sqrt())
    feat_max = float(torch.cat([v for v in matching_pool_feats(enc, tmp_path / "b")]).abs().max())
    print(f"heavy-tailed e2e: waveform rms vs oracle {rms:.2e} (signal rms {float(ref.double().pow(2).mean().sqrt()):.3f}); "
          f"largest |feature| {feat_max:.1f}; range plan {enc.plan['layers']}")
    assert rms < 1e-4, rms


def matching_pool_feats(enc, folder):
    from knn_svc_amd import matching
    mp = matching.get_complete_spk_pool(folder, enc, device=DEV)[0]
    return list(mp.values())
