"""GPU: the product-level entry points for many sources against one pool (BASELINE cfg 3 / cfg 5) — every
utterance converted in a batch, through the stream pipeline, must come out as it does converted alone, and (cfg 3 size) as the
CPU oracle converts it."""
import os

import numpy as np
import pytest
import torch

from knn_svc_amd import audio_io, config as C, synthetic as S

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _tiny_vc(kind="mix"):
    from knn_svc_amd.matcher import KNeighborsVC
    from knn_svc_amd.vocoder import Vocoder
    from knn_svc_amd.wavlm import WavLMEncoder
    cfg, h = C.WAVLM_TINY, C.HIFIGAN_TINY
    enc = WavLMEncoder(S.seeded_state(S.wavlm_param_spec(cfg), seed=11), cfg, DEV, n_layers=2)
    return KNeighborsVC(enc, Vocoder(S.seeded_state(S.generator_param_spec(h, kind), 63 if kind == "mix" else 64), h, kind, DEV), h, DEV)


def _write_pool(d, n=4, secs=3):
    d.mkdir(parents=True)
    for i in range(n):
        w, f = S.synth_clip(secs * 16000 + 37 * i, seed=900 + i)
        audio_io.write_wav_pcm16(str(d / f"t{i}.wav"), w, 16000)
        np.save(d / f"t{i}_f0.npy", f)


@pytest.mark.parametrize("ckpt_type,kind", [("mix", "mix"), ("wavlm_only", "f0")])
def test_many_to_one_equals_special_match_per_source(tmp_path, ckpt_type, kind):
    """KNeighborsVC.many_to_one (serving.TargetVoice + BatchConverter: pool built once, grouped kNN searches, lane pipeline,
    generator as the tail stage) writes, for every source, the samples ``special_match`` writes for that source alone — both
    generator variants, ragged source lengths incl. one shorter than a chunk bucket."""
    from knn_svc_amd.matching import match_at_inference_time
    vc = _tiny_vc(kind)
    pool = tmp_path / "tgt"
    _write_pool(pool)
    srcd = tmp_path / "src"; srcd.mkdir()
    lens = [16000 * 2 + 11, 16000 * 3, 9000, 16000 * 2 + 11, 16000 + 641, 40000]
    files = []
    for i, n in enumerate(lens):
        w, f = S.synth_clip(n, seed=700 + i)
        p = srcd / f"s{i}.wav"
        audio_io.write_wav_pcm16(str(p), w, 16000); np.save(srcd / f"s{i}_f0.npy", f * 1.2)
        files.append(str(p))
    post_opt = "post_opt_0.2"
    out = vc.many_to_one(files, str(pool), str(tmp_path / "out"), ckpt_type=ckpt_type, post_opt=post_opt)
    assert len(out) == len(files)
    for p, o in zip(files, out):
        stem = os.path.basename(p).split(".")[0]
        assert os.path.basename(o) == f"{stem}_to_tgt_knn_{ckpt_type}_{post_opt}.wav"
        # the single-source path (special_match's body, ddsp_matcher.py:937-995, with a folder as the target)
        if kind == "mix":
            of, hw, _a, sf0 = match_at_inference_time(p, pool, vc.wavlm, vc.weighting, vc.weighting, prioritize_f0=True,
                                                      ckpt_type=ckpt_type, post_opt=post_opt, tgt_dataset_path=tmp_path)
            y = vc.vocode(of[p][None], sf0[p][None, :, None], hw[p][None]).squeeze()
        else:
            of, _a, sf0 = match_at_inference_time(p, pool, vc.wavlm, vc.weighting, vc.weighting, prioritize_f0=True,
                                                  ckpt_type=ckpt_type, tgt_dataset_path=tmp_path)
            y = vc.vocode(of[p][None], sf0[p][None, :, None]).squeeze()
        x, sr = audio_io.read_wav(o)
        assert sr == 16000 and x.shape == (1, y.numel())
        ref = audio_io.to_pcm32(y.detach().cpu().numpy()[None])
        got = audio_io.to_pcm32(x)
        assert np.array_equal(ref, got), (stem, int(np.abs(ref.astype(np.int64) - got.astype(np.int64)).max()))


def test_request_queue_batches_and_returns_each_request_its_own_waveform(tmp_path):
    """serving.RequestQueue: requests submitted from several threads are drained into batches (dynamic batching) and every
    Future gets the waveform of ITS request — equal to converting that request alone; in-memory requests with and without an
    f0 track (Harvest on the GPU when it is missing)."""
    import threading
    from knn_svc_amd import serving
    vc = _tiny_vc()
    pool = tmp_path / "tgt"
    _write_pool(pool)
    tv = serving.TargetVoice(vc, str(pool))
    conv = serving.BatchConverter(vc, tv, "mix", "post_opt_0.2")
    reqs = []
    for i in range(9):
        w, f = S.synth_clip(16000 + 2000 * i + 7, seed=800 + i)
        reqs.append((w, None if i % 4 == 3 else (f * 1.1).astype(np.float32)))
    alone = [conv.convert([r])[0].cpu() for r in reqs]
    rq = serving.RequestQueue(conv, max_batch=4, max_wait_ms=200.0)
    futs = [None] * len(reqs)

    def client(i):
        futs[i] = rq.submit(reqs[i])
    ths = [threading.Thread(target=client, args=(i,)) for i in range(len(reqs))]
    [t.start() for t in ths]; [t.join() for t in ths]
    got = [f.result(timeout=30) for f in futs]
    rq.close()
    assert sum(rq.batches) == len(reqs) and max(rq.batches) <= 4 and len(rq.batches) < len(reqs), rq.batches
    for a, b in zip(alone, got):
        assert torch.equal(a, b)
    with pytest.raises(RuntimeError):
        rq.submit(reqs[0])


def test_request_queue_a_bad_request_fails_alone_and_one_deadline_per_batch(tmp_path):
    """(a) a malformed request (an f0 track of the wrong length) and a poisonous one (NaN samples) fail ONLY their
    own Futures, at the door — the other requests of the same batch are converted and equal their stand-alone conversions; a
    NaN source that is handed to the converter directly raises the reference's "containing nan" instead of taking the process
    down (round 3: NaN costs left the concat re-selection without a ranking and the next frame gathered through garbage); (b) the batching window is one deadline per batch, set by its first request: a trickle of
    arrivals spaced closer than max_wait cannot hold the first request for max_batch x max_wait."""
    import time
    from knn_svc_amd import ops, serving
    vc = _tiny_vc()
    pool = tmp_path / "tgt"
    _write_pool(pool)
    tv = serving.TargetVoice(vc, str(pool))
    conv = serving.BatchConverter(vc, tv, "mix", "post_opt_0.2")
    reqs = []
    for i in range(4):
        w, f = S.synth_clip(16000 + 1500 * i + 11, seed=840 + i)
        reqs.append((w, (f * 1.1).astype(np.float32)))
    alone = [conv.convert([r])[0].cpu() for r in reqs]
    bad_f0 = (reqs[1][0], reqs[1][1][:-7])
    nan_w = reqs[2][0].copy(); nan_w[5000:5100] = np.nan
    rq = serving.RequestQueue(conv, max_batch=8, max_wait_ms=400.0)
    # a stereo [2, L] array must not be flattened into one 2L-sample mono clip, and a NaN waveform WITHOUT an f0 track
    # must be refused before Harvest runs on it — both are checked on the raw input
    stereo = (np.stack([reqs[0][0], reqs[0][0]]), reqs[0][1])
    futs = [rq.submit(r) for r in (reqs[0], bad_f0, (nan_w, reqs[2][1]), reqs[3], stereo, (nan_w, None))]
    with pytest.raises(ValueError):
        futs[1].result(timeout=60)
    with pytest.raises(ValueError):
        futs[2].result(timeout=60)
    with pytest.raises(ValueError, match="mono"):
        futs[4].result(timeout=60)
    with pytest.raises(ValueError, match="NaN"):
        futs[5].result(timeout=60)
    assert torch.equal(futs[0].result(timeout=60), alone[0]) and torch.equal(futs[3].result(timeout=60), alone[3])
    assert rq.batches == [6] and rq.isolated == 0, (rq.batches, rq.isolated)
    assert torch.equal(conv.convert([(reqs[0][0][None], reqs[0][1])])[0].cpu(), alone[0])       # [1, L]: mono with a channel axis
    rq.close()
    with pytest.raises(ops.KnnSvcError):                 # not through the queue: the whole batch is enqueued, then the flag raises
        conv.convert([reqs[0], (nan_w, reqs[2][1]), reqs[3]])
    torch.cuda.synchronize()
    assert torch.equal(conv.convert([reqs[3]])[0].cpu(), alone[3])                 # ... and the device is still fine
    # (b) five requests 120 ms apart with max_wait 300 ms: the first batch closes 300 ms after ITS first request
    rq = serving.RequestQueue(conv, max_batch=8, max_wait_ms=300.0)
    t0 = time.monotonic()
    futs = []
    for i in range(5):
        futs.append(rq.submit(reqs[i % 4]))
        time.sleep(0.12)
    [f.result(timeout=60) for f in futs]
    rq.close()
    assert len(rq.batches) >= 2 and rq.batches[0] <= 4, rq.batches          # per-arrival timers would have made it one batch of 5


def test_cfg5_share_through_the_product_entry_equals_per_item_path():
    """BASELINE cfg 5, one GPU's share at full size: 32 x 30 s sources against a resident 60-minute pool (180 000 frames),
    WavLM-Large (6 layers) + the full generator with seeded weights, mix, post_opt_0.2, through serving.BatchConverter (grouped
    searches on the fused screen + refine route, three lanes, generator tail).  Sampled sources are converted once more ALONE
    (one search of their own 1500 frames — the dot-matrix route —, no pipeline): the same waveform up to the rounding of the
    batch-wide operand scales (bit for bit on most sources)."""
    from knn_svc_amd import ops, serving
    from knn_svc_amd.matcher import KNeighborsVC
    from knn_svc_amd.vocoder import Vocoder
    from knn_svc_amd.wavlm import WavLMEncoder
    enc = WavLMEncoder(S.seeded_state(S.wavlm_param_spec(C.WAVLM_LARGE, 6), seed=1), C.WAVLM_LARGE, DEV, 6)
    voc = Vocoder(S.seeded_state(S.generator_param_spec(C.HIFIGAN_V1, "mix"), seed=2), C.HIFIGAN_V1, "mix", DEV)
    vc = KNeighborsVC(enc, voc, C.HIFIGAN_V1, DEV)
    n = 30 * C.SAMPLE_RATE
    tv = serving.TargetVoice.from_clips(vc, [S.synth_clip(n, seed=5000 + i) for i in range(120)])
    assert tv.frames == 180000
    srcs = [S.synth_clip(n, seed=7000 + i) for i in range(32)]
    reqs = [(w, (f * 1.3).astype(np.float32)) for w, f in srcs]
    conv = serving.BatchConverter(vc, tv, "mix", "post_opt_0.2")
    fused0, dot0 = ops.KNN_ROUTE_COUNTS["fused"], ops.KNN_ROUTE_COUNTS["dot"]
    ys = conv.convert(reqs)
    assert ops.KNN_ROUTE_COUNTS["fused"] > fused0, "the grouped searches of a 32-source batch take the fused route"
    assert len(ys) == 32 and all(y.numel() == n for y in ys)
    exact = 0
    for i in (0, 13, 31):
        alone = conv.convert([reqs[i]])[0]
        d = (alone - ys[i]).double()
        rms, mx = float(d.pow(2).mean().sqrt()), float(d.abs().max())
        exact += int(mx == 0.0)
        print(f"source {i}: batch vs alone max |diff| {mx:.2e}, rms {rms:.2e}")
        # Same kernels, same order; what a batch changes is the power-of-two operand scale the f16x2 GEMMs derive from the range
        # slot (max |x| over everything encoded / searched together): exact for all but the elements whose low fp16 piece goes
        # subnormal, i.e. fp32-rounding-level differences (measured: 0 on most sources, 6e-7 max on one).  Bar: the north-star
        # waveform tolerance, two orders of magnitude above that.
        assert rms < 1e-6 and mx < 1e-4, (i, rms, mx)
    assert exact == 3, "bit-identical since the re-selection kernel runs alone on its CU and the encoder is batch-invariant"
    # (round 3: a source on its own took the dot-matrix route, so this also compared the two routes; since round 4 a 30 s source
    #  takes the fused route as well — with another epoch plan than the grouped search: 6 row tiles instead of 12 or 24)
    assert ops.KNN_ROUTE_COUNTS["dot"] >= dot0


def test_cfg3_size_bulk_match_sampled_utterances_vs_oracle(tmp_path):
    """BASELINE cfg 3 at its pool size: dataset mode (``bulk_match``, ddsp_matcher.py:1027-1155) with --dur_limit 600 — a
    10-minute target pool (30 000 frames) per speaker pair, utterances of 5-10 s — WavLM-Large (6 layers) + full generator,
    seeded weights, mix, post_opt_0.2.  The CPU oracle cannot encode 20 minutes of audio in test time, so it is given the
    GPU's features (the encoder has its own full-size oracle tests) and restates everything behind them — kNN, f0 shift and
    re-rank, both concat re-selections, both Adam loops, weighted sums, additive synth, generator — for SAMPLED utterances:
    the written waveform must agree within the north-star tolerance (1e-4 RMS)."""
    from knn_svc_amd import matching
    from knn_svc_amd.matcher import KNeighborsVC
    from knn_svc_amd.vocoder import Vocoder
    from knn_svc_amd.wavlm import WavLMEncoder
    from oracle import pipeline_ref, vocoder_ref
    sdg = S.seeded_state(S.generator_param_spec(C.HIFIGAN_V1, "mix"), seed=2)
    enc = WavLMEncoder(S.seeded_state(S.wavlm_param_spec(C.WAVLM_LARGE, 6), seed=1), C.WAVLM_LARGE, DEV, 6)
    vc = KNeighborsVC(enc, Vocoder(sdg, C.HIFIGAN_V1, "mix", DEV), C.HIFIGAN_V1, DEV)
    root = tmp_path / "data"
    rng = np.random.default_rng(3)
    for s, (spk, n_utt) in enumerate((("spkA", 6), ("spkB", 90))):          # spkB: ~11 minutes, cut at 600 s by the limit
        (root / spk).mkdir(parents=True)
        for u in range(n_utt):
            secs = float(rng.uniform(5.0, 10.0)) if s == 1 else 5.0 + 0.1 * u          # (short sources: the oracle's generator runs on the CPU)
            w, f = S.synth_clip(int(secs * 16000), seed=1000 * s + u)
            audio_io.write_wav_pcm16(str(root / spk / f"u{u:03d}.wav"), w, 16000)
            np.save(root / spk / f"u{u:03d}_f0.npy", (f * (1.25 if s == 0 else 1.0)).astype(np.float32))
    csvp = tmp_path / "subset.csv"          # only spkA -> spkB
    csvp.write_text("src,tgt,key,x,label\n" + "".join(f"a,b,u{u:03d}/spkB,x,0\n" for u in range(6)))
    out_dir = tmp_path / "out"
    written = vc.bulk_match(str(root), str(root), str(out_dir), ckpt_type="mix", post_opt="post_opt_0.2",
                            required_subset_file=str(csvp), duration_limit=600)
    assert len(written) == 6
    # the pool the run used, from the store (every file was encoded once), as the oracle's input
    mp, _s, _a, _sp, f0p, hp = matching.get_complete_spk_pool(root / "spkB", enc, device=DEV, duration_limit=600)
    P = torch.cat(list(mp.values())).cpu(); Pf0 = torch.cat(list(f0p.values())).cpu(); Ph = torch.cat(list(hp.values())).cpu()
    assert 30000 <= P.shape[0] < 30000 + 501
    qp, _s, _a, _sp, qf0p, _h = matching.get_complete_spk_pool(root / "spkA", enc, device=DEV)
    for u in (4,):
        key = str(root / "spkA" / f"u{u:03d}.wav")
        query = dict(feats=qp[key].cpu(), f0=qf0p[key].cpu())
        pool = dict(feats=P, f0=Pf0, harm=Ph)
        of, hw, sf0 = pipeline_ref.match(query, pool, "mix", "post_opt_0.2")[:3]
        ref = vocoder_ref.synthesizer(sdg, C.HIFIGAN_V1, "mix", of[None], sf0[None, :, None], hw[None]).reshape(-1).numpy().astype(np.float64)
        ref = ref / max(1.0, float(np.abs(ref).max()))                     # save_audio's scaling (lib_ongaku_test.py:102-112)
        x, sr = audio_io.read_wav(str(out_dir / "spkA" / f"u{u:03d}" / "spkB.wav"))
        assert sr == 16000 and x.shape[1] == ref.size
        rms = float(np.sqrt(np.mean((x[0].astype(np.float64) - ref) ** 2)))
        print(f"cfg-3-size utterance u{u:03d}: waveform rms vs oracle {rms:.2e} (signal rms {float(np.sqrt(np.mean(ref ** 2))):.3f})")
        assert rms < 1e-4, (u, rms)


def test_pipelined_batch_is_run_to_run_deterministic_and_equals_the_unpipelined_path():
    """The same batch through serving.BatchConverter three times (grouped fused searches on the kNN stream, three match lanes with
    their partner streams, the generator on the tail stream — everything overlapping) gives the same samples every time, and every
    source equals its conversion alone.  Round 3 found this NOT to hold: the frame-sequential re-selection kernel returned slightly
    different costs for a few frames whenever another kernel's workgroup shared its CU (csrc/select.hip: it now takes the CU's whole
    LDS), on top of two batch-dependent choices in the encoder (GEMM kernel by tile count, an operand scale from a batch-wide range
    slot).  Full-size models, 8 x 30 s sources against a 20-minute pool."""
    from knn_svc_amd import serving
    from knn_svc_amd.matcher import KNeighborsVC
    from knn_svc_amd.vocoder import Vocoder
    from knn_svc_amd.wavlm import WavLMEncoder
    enc = WavLMEncoder(S.seeded_state(S.wavlm_param_spec(C.WAVLM_LARGE, 6), seed=1), C.WAVLM_LARGE, DEV, 6)
    voc = Vocoder(S.seeded_state(S.generator_param_spec(C.HIFIGAN_V1, "mix"), seed=2), C.HIFIGAN_V1, "mix", DEV)
    vc = KNeighborsVC(enc, voc, C.HIFIGAN_V1, DEV)
    n = 30 * C.SAMPLE_RATE
    tv = serving.TargetVoice.from_clips(vc, [S.synth_clip(n, seed=5000 + i) for i in range(40)])
    reqs = [(w, (f * 1.3).astype(np.float32)) for w, f in (S.synth_clip(n, seed=7000 + i) for i in range(8))]
    conv = serving.BatchConverter(vc, tv, "mix", "post_opt_0.2")
    runs = [[y.clone() for y in conv.convert(reqs)] for _ in range(4)]      # eager first sights, capture, replays
    for r in (2, 3):
        bad = [(i, float((runs[1][i] - runs[r][i]).abs().max())) for i in range(8) if not torch.equal(runs[1][i], runs[r][i])]
        assert not bad, ("pipelined runs differ", r, bad)
    alone = [conv.convert([reqs[i]])[0] for i in range(8)]
    bad = [(i, float((runs[3][i] - alone[i]).abs().max())) for i in range(8) if not torch.equal(runs[3][i], alone[i])]
    assert not bad, ("batch differs from the per-source path", bad)
