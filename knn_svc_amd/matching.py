"""Feature-pool building and frame matching on the GPU.

Host-side mirror of the reference's ``get_complete_spk_pool`` and
``match_at_inference_time`` (ddsp_prematch_dataset.py:301-414, 1074-1459): same
arguments, same returned dict-of-tensors keyed by ``str(path)``, same quirks
(k hard-coded to 32 -> first 4, ``--dur_limit`` in seconds overshooting by one file,
``post_opt`` parsing, ``prioritize_f0`` must be True, pool rebuilt per call).
All arithmetic is done by libknnsvc_hip.so kernels on device-resident tensors; the
only host work is file I/O and the chunk bookkeeping.
"""
from __future__ import annotations

import contextlib
import os
from pathlib import Path

import numpy as np
import torch

from . import audio_io, config as C, dist as kdist, features, ops, pipeline, pool_cache
from .wavlm import WavLMEncoder, chunk_plan

AUDIO_EXT = {".flac", ".wav", ".mp3"}


def parse_post_opt(post_opt: str):
    """(concat_weight, run_adam): float of the last '_' token, 'extra' -> 0.3, else -1 (off);
    the Adam stage is skipped iff 'no_post_opt' is a substring (ddsp_prematch_dataset.py:1273-1279, 1356)."""
    tail = post_opt.split("_")[-1]
    try:
        w = float(tail)
    except ValueError:
        w = 0.3 if tail == "extra" else -1
    return w, ("no_post_opt" not in post_opt)


def list_audio(path) -> list:
    """A single audio file, or every audio file under a folder in sorted rglob order (:313-319)."""
    path = Path(path)
    if os.path.isfile(path) and os.path.splitext(path)[-1] in AUDIO_EXT:
        files = [path]
    else:
        files = sorted([p for p in path.rglob("**/*") if p.suffix.lower() in AUDIO_EXT])
        assert len(files) != 0, [f"directory not containing any audio {path}"]
    # The reference decodes every listed container through torchaudio (:332).  Here .wav and .flac are decoded natively and
    # .mp3 only when soundfile is importable: a pool that holds a file nobody can decode is refused NOW, by name, not in the
    # middle of a run after the other files have been encoded.
    bad = [str(p) for p in files if not audio_io.can_decode(p.suffix)]
    if bad:
        raise RuntimeError(f"{len(bad)} audio file(s) under {path} are in a container this build cannot decode without the "
                           f"'soundfile' package (.wav and .flac are native): {bad[:5]}{' ...' if len(bad) > 5 else ''} — "
                           "transcode them to .wav / .flac or install soundfile")
    return files


def load_utterance(pth):
    """-> (wav float32 [L] mono 16 kHz on the host, f0 float32 array).  Channel mean for multi-channel
    input (:333-335); f0 from ``<stem>_f0.npy`` (:373-382)."""
    x, sr = audio_io.load_audio(str(pth))
    if x.shape[0] > 1:
        x = x.mean(axis=0, keepdims=True)
    if sr != C.SAMPLE_RATE:      # on the GPU (features.resample); the reference resamples with torchaudio on the host (:338-341)
        x = features.resample(torch.from_numpy(np.ascontiguousarray(x[0], dtype=np.float32)).cuda(), sr,
                              C.SAMPLE_RATE).cpu().numpy()[None]
    f0_path = os.path.splitext(str(pth))[0] + "_f0.npy"
    if not os.path.isfile(f0_path):
        # same bookkeeping as the reference (:376-379): warn, compute, write the cache next to the audio.  The reference runs
        # pyworld.harvest on the host here (:121-128); this build runs Harvest on the GPU (csrc/harvest.hip, fp64, pinned on
        # the reference's own shipped tracks).
        xg = torch.from_numpy(np.ascontiguousarray(x[0], dtype=np.float32)).cuda()
        print(f"WARNING: {f0_path} not exists, generating...")
        f0_new = ops.f0_harvest(xg).cpu().numpy()
        np.save(f0_path, f0_new)
    f0 = np.asarray(np.load(f0_path, allow_pickle=True), dtype=np.float32)
    return np.ascontiguousarray(x[0], dtype=np.float32), f0


def frames_of(n_samples: int, enc: WavLMEncoder) -> int:
    return sum(enc.n_frames(l + p) for (_s, l, p) in chunk_plan(n_samples))


def side_features(wav_gpu: torch.Tensor, f0_host: np.ndarray, T: int):
    """STFT magnitude -> harmonic amplitudes for one utterance (:361-404).  Returns (f0 [T], harm [T,49], spec [T,200])."""
    assert wav_gpu.numel() >= C.HOP * T
    spec = features.stft_mag(wav_gpu)
    assert spec.shape[0] >= T
    spec = spec[:T].contiguous()
    assert abs(len(f0_host) - T) <= 1 and len(f0_host) >= T, [len(f0_host), T]
    if isinstance(f0_host, torch.Tensor):           # already resident on the device
        f0 = f0_host[:T].contiguous()
    else:
        f0 = torch.from_numpy(np.ascontiguousarray(f0_host[:T])).to(wav_gpu.device)
    harm = ops.harmonic_amps(spec, f0, C.N_HARM)
    return f0, harm, spec


def side_features_many(wavs_gpu, f0s_host, Ts):
    """``side_features`` for a list of utterances in four launches (features.stft_harm_batch) instead of four per file.
    -> list of (f0 [T], harm [T,49], spec [T,200]); identical values."""
    if wavs_gpu and not wavs_gpu[0].is_cuda:          # (CPU tensors: the gloo tests inject their stand-in at ``side_features``)
        return [side_features(w, f0, T) for w, f0, T in zip(wavs_gpu, f0s_host, Ts)]
    f0s = []
    for w, f0, T in zip(wavs_gpu, f0s_host, Ts):
        assert w.numel() >= C.HOP * T
        assert abs(len(f0) - T) <= 1 and len(f0) >= T, [len(f0), T]
        f0s.append(f0[:T].contiguous() if isinstance(f0, torch.Tensor)
                   else torch.from_numpy(np.ascontiguousarray(f0[:T])).to(w.device, non_blocking=True))
    res = features.stft_harm_batch(wavs_gpu, f0s, Ts, n_harm=C.N_HARM)
    return [(f0, harm, spec) for f0, (spec, harm) in zip(f0s, res)]


def disk_identity(wavlm) -> tuple:
    """What an on-disk pool-store entry belongs to besides its audio file: the encoder's WEIGHTS, its exit layer and the layer
    weighting the features were mixed under (set_layer_mix bumps ``uid`` for the in-memory tier; the disk tier outlives the
    process, so it needs the weighting itself — without it a second weighting read back the first one's features)."""
    return (wavlm.weights_fingerprint(), wavlm.n_layers, getattr(wavlm, "layer_mix", None))


def get_complete_spk_pool(path, wavlm: WavLMEncoder, match_weights=None, synth_weights=None, device="cuda",
                          duration_limit=None, vad_trigger_level=0, shard_files=False, gather=False):
    """Per-file dicts (matching_pool, synth_pool, audio_synth_pool, spec_synth_pool, f0_pool, harmonics_pool),
    like the reference.  matching == synth features under the encoder's CURRENT layer mix (both weightings are the same one-hot on
    the live path; match_at_inference_time calls this once per weighting when they differ);
    ``audio_synth_pool`` is kept as None values: the live path never reads it (audio_out_feats_weighted = None,
    ddsp_prematch_dataset.py:1368).  Per-file results are kept in the device-resident pool store
    (knn_svc_amd/pool_cache.py) so that dataset mode encodes every file once instead of once per speaker pair.
    ``shard_files`` (one process per GPU): every rank walks the same file list and duration limit, but encodes and
    returns only its contiguous share of the kept files (dist.contiguous_share) — the pool shard of BASELINE cfg 4.
    ``gather`` (with shard_files): the shares are all-gathered afterwards (features and f0 only — what a QUERY pool
    needs), so every rank returns the complete per-file dicts although it encoded only its share."""
    dev = wavlm.device
    files = list_audio(path)
    cache = _pool_cache()
    tag = (wavlm.uid, wavlm.n_layers)
    dtag = disk_identity(wavlm) if cache.disk_dir else None     # on-disk tier: content identity
    dkeys = {}
    kept, keys, Ts = [], [], []
    loaded = {}                       # index -> (wav host, f0 host) for files that miss the cache
    hits = {}                         # index -> cache entry: held here, a later put() may evict it from the store
    dur = 0.0
    for i, pth in enumerate(files):
        key = pool_cache.file_key(pth, tag)
        if dtag is not None:
            dkeys[key] = pool_cache.file_key(pth, dtag)
        ent = cache.get(key, dkeys.get(key), dev)
        if ent is not None:
            T = ent["feats"].shape[0]
            hits[i] = ent
        else:
            w, f0 = load_utterance(pth)
            key = pool_cache.file_key(pth, tag)                   # load_utterance may have just written <stem>_f0.npy,
            if dtag is not None:                                  # which is part of the file's identity
                dkeys[key] = pool_cache.file_key(pth, dtag)
            T = frames_of(len(w), wavlm)
            loaded[i] = (w, f0)                                   # host arrays: only this rank's share goes to the device
        kept.append(str(pth)); keys.append(key); Ts.append(T)
        dur += T * C.HOP / C.SAMPLE_RATE
        if duration_limit is not None and dur >= duration_limit:
            break
    lo, hi = kdist.contiguous_share(len(kept)) if shard_files else (0, len(kept))
    loaded = {i: (torch.from_numpy(v[0]).to(dev), v[1]) for i, v in loaded.items() if lo <= i < hi}
    miss = sorted(loaded)
    feats = wavlm.encode_many([loaded[i][0] for i in miss], pow2_batches=True) if miss else []
    sides = side_features_many([loaded[i][0] for i in miss], [loaded[i][1] for i in miss], [Ts[i] for i in miss])
    for i, ft, (f0, harm, spec) in zip(miss, feats, sides):
        assert ft.shape[0] == Ts[i]
        hits[i] = dict(feats=ft, f0=f0, harm=harm, spec=spec)
        cache.put(keys[i], hits[i], dkeys.get(keys[i]))
    matching, synth, audio, specs, f0p, harmp = {}, {}, {}, {}, {}, {}
    if shard_files and gather and kdist.world()[1] > 1:
        # query pool: every rank needs every file's features and f0 (replicated kNN queries); rank order = file order
        mine = [hits[i] for i in range(lo, hi)]
        E = wavlm.E
        cat = lambda k, shape: (torch.cat([e[k] for e in mine], 0).contiguous() if mine
                                else torch.empty(shape, device=dev, dtype=torch.float32))
        counts = kdist.shard_rows(sum(Ts[lo:hi]), dev)
        all_f = kdist.all_gather_rows_var(cat("feats", (0, E)), counts)
        all_f0 = kdist.all_gather_rows_var(cat("f0", (0,)), counts)
        r0 = 0
        for key, T in zip(kept, Ts):
            matching[key] = synth[key] = all_f[r0:r0 + T]
            f0p[key] = all_f0[r0:r0 + T]
            audio[key] = specs[key] = harmp[key] = None
            r0 += T
        return matching, synth, audio, specs, f0p, harmp
    for i, key in enumerate(kept):
        if not lo <= i < hi:
            continue
        ent = hits[i]
        matching[key] = ent["feats"]; synth[key] = ent["feats"]; audio[key] = None
        specs[key] = ent["spec"]; f0p[key] = ent["f0"]; harmp[key] = ent["harm"]
    return matching, synth, audio, specs, f0p, harmp


_POOL_CACHE = None


def _pool_cache():
    global _POOL_CACHE
    if _POOL_CACHE is None:
        _POOL_CACHE = pool_cache.PoolCache()
    return _POOL_CACHE


_SIDE = {}


def _side_stream(device) -> "torch.cuda.Stream":
    """Partner stream of the CURRENT stream (one per (device, stream)): concurrent match_features calls on
    different streams (pipeline.LanePipeline) must not meet on one shared side stream."""
    idx = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
    key = (idx, torch.cuda.current_stream(idx).cuda_stream)
    if key not in _SIDE:
        from .pipeline import new_stream
        cur = torch.cuda.current_stream(idx)
        # a partner has to run BESIDE its stream — and beside the lanes, tails and other partners: measured at creation
        # (pipeline.new_stream, overlap_with="all"), not left to the stream -> hardware-queue mapping
        from . import pipeline
        vetted = [st for (d_, _k, _i, _p), st in pipeline._VETTED.items() if d_ == idx]
        if cur.cuda_stream == 0 or any(cur.cuda_stream == st.cuda_stream for st in vetted):
            # the partner of the default stream or of a measured (single-lane) pipeline stream: measured too
            _SIDE[key] = new_stream(idx, priority=cur.priority, kind="partner", overlap_with="all", must=[cur] + vetted)
        else:
            _SIDE[key] = new_stream(idx, priority=cur.priority, kind="partner")
    return _SIDE[key]


def prepare_pool(matching_list, split=True):
    """Per-pool quantities that do not depend on the query (row norms, the split image the kNN GEMM reads): computed
    once per target pool in dataset mode instead of once per utterance.  ``split=False``: norms only (the neighbours
    come from the pool-sharded search)."""
    P = matching_list
    stats = ops.row_norms(P)
    return dict(stats=stats, split=ops.prepare_knn_pool(P, C.KNN_K, stats) if split else None)


# query frames per grouped search in dataset mode.  Measured (tools/knn_group_ab.sh; cfg 5 share, xRT): 1500 -> 2321, 3000 -> 2349,
# 8192 -> 1884, everything in one search -> 1864; cfg 3 is flat (1331-1351).  Groups large enough for the fused screen + refine route
# (>= 4096 rows) LOSE inside a pipeline: its one-block-per-CU, 128 KB-LDS launches leave no room next to them for the single-workgroup
# recurrences and the generator of the other items.  3000 frames keeps the searches on the 128x128 kernel and ahead of the lanes.
KNN_GROUP_FRAMES = int(os.environ.get("KNNSVC_KNN_GROUP_FRAMES", "3000"))
_KNN_STREAMS = {}


def _knn_stream(device) -> "torch.cuda.Stream":
    """The stream the grouped kNN searches of dataset mode run on (ahead of the lanes that consume their results)."""
    key = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
    if key not in _KNN_STREAMS:
        from .pipeline import new_stream
        _KNN_STREAMS[key] = new_stream(device, kind="knn")
    return _KNN_STREAMS[key]


# CUs a grouped search's screening kernel may occupy inside the dataset-mode pipeline (it is persistent: a block walks tiles); the
# rest stays free for the single-workgroup recurrences and the generator of the items in flight.  0 = all.
KNN_PIPE_BLOCKS = int(os.environ.get("KNNSVC_KNN_PIPE_BLOCKS", "192"))


def batched_knn(q_all, matching_list, prep, max_blocks=0):
    """Top-32 of the stacked frames of many query utterances against one prepared pool -> (idx [sum Nq, 32], flag: NaN /
    overflow bits, read by the caller at the end)."""
    idx, _d, fl = ops.knn_topk(q_all, matching_list, C.KNN_K, p_stats=prep["stats"], prepared=prep["split"],
                               check_nan=False, return_flag=True, max_blocks=max_blocks)
    return idx, fl


def grouped_knn(items, query_pool, matching_list, prep, flags):
    """Top-32 of every item's frames against one prepared pool, searched in GROUPS of at least KNN_GROUP_FRAMES query frames
    one after the other on a stream of their own: group g + 1 is searched while the match bodies and the generator of group g
    run, instead of one search of everything in front of the whole pipeline (32 x 30 s sources against a 60-minute pool:
    16.1 -> 12.8 ms per source).  -> (nn: item -> [Nq, 32] indices, ready: item -> event a consumer
    on another stream has to wait for; empty without a kNN stream).  Results do not depend on the grouping."""
    nn, nn_ready = {}, {}
    groups, cur_g, cur_n = [], [], 0
    for it in items:
        cur_g.append(it); cur_n += query_pool[it].shape[0]
        if cur_n >= KNN_GROUP_FRAMES:
            groups.append(cur_g); cur_g, cur_n = [], 0
    if cur_g:
        groups.append(cur_g)
    main = torch.cuda.current_stream(matching_list.device) if matching_list.is_cuda else None
    ks = _knn_stream(matching_list.device) if main is not None and len(groups) > 1 else None
    if ks is not None:
        ks.wait_stream(main)
    prod = ks if ks is not None else main          # the stream the searches are enqueued on (None: CPU stand-ins in the gloo tests)
    for grp in groups:
        ctx = torch.cuda.stream(ks) if ks is not None else contextlib.nullcontext()
        with ctx:
            q_all = torch.cat([query_pool[it] for it in grp], 0).contiguous()
            idx_all, fl = batched_knn(q_all, matching_list, prep, max_blocks=KNN_PIPE_BLOCKS if len(groups) > 1 else 0)
            parts = [t.contiguous() for t in idx_all.split([query_pool[it].shape[0] for it in grp])]
            # ALWAYS an ordering token, also for a single group searched on the caller's own stream: the lists are consumed on
            # lane streams, and a tensor handed out without one is a race waiting for a caller that does not happen to wait
            ev = prod.record_event() if prod is not None else None
        if fl is not None:
            flags.append(fl)
        for it, t in zip(grp, parts):
            nn[it] = t
            if ev is not None:
                nn_ready[it] = ev
    return nn, nn_ready


def ordering_token(device):
    """Event on the current stream: the token for neighbour lists produced on it (pool-sharded searches)."""
    return torch.cuda.current_stream(device).record_event()


def wait_for_neighbours(nn_item, ready_event, device):
    """Make the current stream wait for a neighbour list produced on another stream.  Every device-resident list comes with
    the event of its producing stream (grouped_knn, ordering_token); a list without one is refused rather than read early."""
    if nn_item is None:
        return
    if ready_event is None:
        if nn_item.is_cuda:
            raise RuntimeError("wait_for_neighbours: device-resident neighbour list without an ordering event")
        return
    st = torch.cuda.current_stream(device)
    st.wait_event(ready_event)
    nn_item.record_stream(st)


def match_features(query_seq, query_f0, matching_list, matching_f0, harmonics_list, ckpt_type, post_opt,
                   return_debug=False, nn32=None, nan_flags=None, pool_prep=None, synth_list=None):
    """The per-query body (ddsp_prematch_dataset.py:1189-1450) on device tensors.  ``synth_list``: the pool under the SYNTHESIS
    layer weighting when it differs from the matching one (:349-350, 1157): the search and the concat re-selection run on
    ``matching_list``, the smoothness weights and the weighted sums on ``synth_list`` (:1260, 1347-1350).  ``nn32`` may carry
    neighbours already found by the pool-sharded search (knn_svc_amd.dist.sharded_knn).  ``nan_flags``: a
    list that receives the kNN NaN flag instead of the host checking it here (the caller then calls
    ``ops.raise_if_nan`` on each entry once everything is enqueued — keeps a stream pipeline free of syncs)."""
    q = query_seq.contiguous()
    P = matching_list
    qn, qs = ops.row_norms(q)
    pn, ps = pool_prep["stats"] if pool_prep is not None else ops.row_norms(P)
    nan_flag = None
    if nn32 is None:
        # NaN check deferred to the end of the launch sequence (one host sync instead of a split stream)
        nn32, _, nan_flag = ops.knn_topk(q, P, C.KNN_K, q_stats=(qn, qs), p_stats=(pn, ps), check_nan=False,
                                         return_flag=True, prepared=pool_prep["split"] if pool_prep is not None else None)
    cw, run_adam = parse_post_opt(post_opt)
    with_harm = "wavlm_only" not in ckpt_type and "no_harm_no_amp" not in ckpt_type
    # The WavLM-feature branch (concat re-selection -> Adam -> weighted sum) and the pitched branch
    # (f0 shift -> re-rank -> concat re-selection -> Adam on harmonics) only share nn32, and each is a
    # chain of single-workgroup latency-bound kernels: run them on two HIP streams.
    main = torch.cuda.current_stream()
    side = _side_stream(q.device)
    side.wait_stream(main)
    it1 = it2 = None
    w = w2 = harm_w = None
    with torch.cuda.stream(side):
        qmed, pmed = ops.log_f0_median(query_f0), ops.log_f0_median(matching_f0)
        shifted = ops.shift_f0(query_f0, qmed, pmed)
        ranked = ops.f0_rerank(nn32, shifted, matching_f0)
        idx2 = ranked[:, :C.KNN_USE].contiguous()
        if cw != -1:
            idx2 = ops.concat_reselect(idx2, q, qn, P, pn, shifted, matching_f0, concat_weight=cw)
        if with_harm:
            if run_adam:
                w2, it2 = ops.smooth_weights(idx2, harmonics_list, 1000.0, return_iters=True)
            harm_w = ops.weighted_gather(idx2, w2, harmonics_list)
    idx = nn32[:, :C.KNN_USE].contiguous()
    if cw != -1:
        idx = ops.concat_reselect(idx, q, qn, P, pn, concat_weight=cw)
    Ps = synth_list if synth_list is not None else P
    if run_adam:
        w, it1 = ops.smooth_weights(idx, Ps, 0.1, return_iters=True)
    out_feats = ops.weighted_gather(idx, w, Ps)
    main.wait_stream(side)
    for t in (shifted, idx2, harm_w, w2):
        if t is not None:
            t.record_stream(main)
    if nan_flag is not None:
        if nan_flags is not None:
            nan_flags.append(nan_flag)
        else:
            try:
                ops.raise_if_nan(nan_flag)
            except ops.KnnOverflow:         # the fused route's candidate buffer overflowed: once more on the dot-matrix route
                with ops.fused_off():
                    return match_features(query_seq, query_f0, matching_list, matching_f0, harmonics_list, ckpt_type, post_opt,
                                          return_debug=return_debug, pool_prep=pool_prep, synth_list=synth_list)
    if return_debug:
        return out_feats, harm_w, shifted, dict(nn32=nn32, idx_wavlm=idx, w_wavlm=w, idx_harm=idx2, w_harm=w2,
                                                iters_wavlm=it1 if it1 is not None else 0,
                                                iters_harm=it2 if it2 is not None else 0)
    return out_feats, harm_w, shifted


def _mix_of(weights, wavlm):
    """A layer weighting as the encoder takes it: None for the one-hot on its exit layer (the live path), else the tuple."""
    if weights is None:
        return None
    w = [float(v) for v in torch.as_tensor(weights).reshape(-1).float().cpu().tolist()]
    if any(v != 0.0 for v in w[wavlm.n_layers + 1:]):
        raise NotImplementedError(f"layer weighting uses layers beyond the {wavlm.n_layers} the encoder was loaded with "
                                  "(load it with n_layers = the last weighted layer)")
    w = (w + [0.0] * (wavlm.n_layers + 1))[:wavlm.n_layers + 1]
    if sum(1 for v in w if v != 0.0) == 1 and w[wavlm.n_layers] == 1.0:
        return None
    return tuple(w)


def match_at_inference_time(src_wav_file, ref_wav_file, wavlm: WavLMEncoder, match_weights=None, synth_weights=None,
                            topk: int = 4, device="cuda", prioritize_f0=False, ckpt_type="wavlm_only",
                            src_dataset_path=None, tgt_dataset_path=None, cache_dir=None, required_subset=None,
                            post_opt="no_post_opt", duration_limit=None, vocode_fn=None, waves_out=None,
                            pool_sharded=None, share_items=False):
    """Same contract as the reference function (ddsp_prematch_dataset.py:1074).  ``topk`` is accepted and
    ignored (k = 32 -> 4 is hard-coded upstream, :1203,1246,1398); ``cache_dir`` is ignored (the reference
    force-disables it, :1086-1087).

    Build extension (BASELINE cfg 5, many sources against one pool): ``vocode_fn(out_feats, shifted_f0, harm)`` is
    enqueued as the pipeline's tail stage right behind each item's match body and its waveform stored in
    ``waves_out[item]`` — the generator of item i then runs underneath the kNN / recurrences of items i+1.., instead
    of after all of them.  It must not synchronise with the host.

    ``pool_sharded`` (default: environment KNNSVC_POOL_SHARD=1; needs a process group, one process per GPU; BASELINE
    cfg 4): EVERY rank must make this call with the same arguments.  The target pool's files are encoded in contiguous
    shares over the ranks, every rank searches the (replicated) query frames in its own shard and the pool's features /
    f0 / harmonics are all-gathered once (rank order = file order, so rows mean what they mean on one GPU).
      * ``share_items=False`` (single file): the per-shard top-32 lists are all-gathered and merged on every rank; every
        rank then holds the same neighbours, runs the same later stages and returns the same dicts.
      * ``share_items=True`` (dataset mode, ``bulk_match``): the source speaker's files are encoded in contiguous shares
        too (features all-gathered), the query items are dealt round-robin over the ranks, ONE all-to-all hands each
        rank the per-shard lists of its own items only, and each rank runs the match bodies (and ``vocode_fn``) of its
        own items: the returned dicts hold this rank's items."""
    import torch.distributed as tdist
    if pool_sharded is None:         # by environment: only when there is more than one rank to shard over
        pool_sharded = os.environ.get("KNNSVC_POOL_SHARD") == "1" and kdist.world()[1] > 1
    pool_sharded = bool(pool_sharded) and tdist.is_available() and tdist.is_initialized()   # explicit True: any group, even 1 rank
    share_items = bool(share_items) and pool_sharded
    assert prioritize_f0, "prioritize_f0=False is unsupported by the reference (ddsp_prematch_dataset.py:1375)"
    if "wavlm_only" not in ckpt_type and "no_harm_no_amp" not in ckpt_type and "mix" not in ckpt_type:
        raise NotImplementedError(ckpt_type)
    # the layer weighting (ddsp_prematch_dataset.py:349-350).  matching == synth features on the live path (ddsp_matcher.py:88-89,
    # 319: both the one-hot on layer 6).  Two DIFFERENT weightings mean two feature sets per target file: the pool is then encoded
    # once per weighting (the reference mixes both from one forward; this case is never taken live — correctness, not speed)
    mix_m, mix_s = _mix_of(match_weights, wavlm), _mix_of(synth_weights, wavlm)
    wavlm.set_layer_mix(mix_m)
    if src_dataset_path is None:
        assert os.path.isfile(src_wav_file)
    query_pool, _, _, _, query_f0_pool, _ = get_complete_spk_pool(src_wav_file, wavlm, device=device,
                                                                  shard_files=share_items, gather=share_items)
    if tgt_dataset_path is None:
        assert os.path.isfile(ref_wav_file)
    matching_pool, _synth, _audio, _spec, f0_pool, harm_pool = get_complete_spk_pool(
        ref_wav_file, wavlm, device=device, duration_limit=duration_limit, shard_files=pool_sharded)
    keys = list(matching_pool)
    E = wavlm.E
    cat = lambda pool, shape: (torch.cat([pool[k] for k in keys], 0).contiguous() if keys
                               else torch.empty(shape, device=wavlm.device, dtype=torch.float32))
    matching_list, matching_f0, harmonics_list = cat(matching_pool, (0, E)), cat(f0_pool, (0,)), cat(harm_pool, (0, C.N_HARM))
    synth_list = None
    if mix_m != mix_s:
        wavlm.set_layer_mix(mix_s)
        try:
            # (a sharded pool: this rank's share again, under the synthesis weighting; all-gathered below like the matching features —
            #  the search only ever sees the matching features, the gathers and the smoothness weights read the synthesis ones)
            _m, synth_pool, _a2, _s2, _f2, _h2 = get_complete_spk_pool(ref_wav_file, wavlm, device=device, duration_limit=duration_limit,
                                                                       shard_files=pool_sharded)
        finally:
            wavlm.set_layer_mix(mix_m)
        assert list(synth_pool) == keys
        synth_list = cat(synth_pool, (0, E))
        assert synth_list.shape == matching_list.shape
    shard = None
    if pool_sharded:
        shard = matching_list                                      # this rank's rows, searched locally
        counts = kdist.shard_rows(shard.shape[0], shard.device)
        if min(counts) < C.KNN_K:
            raise ops.KnnSvcError(f"pool shards {counts}: every rank needs at least {C.KNN_K} pool frames")
        matching_list = kdist.all_gather_rows_var(shard, counts)
        matching_f0 = kdist.all_gather_rows_var(matching_f0, counts)
        harmonics_list = kdist.all_gather_rows_var(harmonics_list, counts)
        if synth_list is not None:
            synth_list = kdist.all_gather_rows_var(synth_list, counts)

    out_c, harm_c, audio_c, f0_c = {}, {}, {}, {}
    items = [item for item in query_pool
             if required_subset is None or
             os.path.basename(item).split(".")[0] + "/" + os.path.basename(ref_wav_file) in required_subset]
    all_items = items

    def run():
        items = all_items
        nn = {}
        nn_ready = {}                 # item -> event of the kNN-stream search that produced nn[item]
        if shard is not None and share_items:
            # one search of ALL items' frames in every shard, one all-to-all: each rank gets the lists of its own items
            rank, ws = kdist.world()
            owned = [[it for j, it in enumerate(items) if j % ws == r] for r in range(ws)]      # == dist.my_share, per rank
            order = [it for part in owned for it in part]
            if order:
                q_all = torch.cat([query_pool[it] for it in order], 0).contiguous()
                rows = [sum(query_pool[it].shape[0] for it in part) for part in owned]
                mine_idx = kdist.sharded_knn_owned(q_all, rows, shard, C.KNN_K, counts=counts)[0]
                r0 = 0
                for it in owned[rank]:
                    n = query_pool[it].shape[0]
                    nn[it] = mine_idx[r0:r0 + n].contiguous()
                    r0 += n
            kdist.raise_if_any_nan()
            items = owned[rank]
            if matching_list.is_cuda:
                tok = ordering_token(matching_list.device)
                nn_ready.update({it: tok for it in nn})
        elif shard is not None:          # collectives first, in item order on every rank; the match bodies then need none
            for item in items:
                nn[item] = kdist.sharded_knn(query_pool[item].contiguous(), shard, C.KNN_K, replicated=True, counts=counts)[0]
            kdist.raise_if_any_nan()
            if matching_list.is_cuda:
                tok = ordering_token(matching_list.device)
                nn_ready.update({it: tok for it in nn})
        # the per-item bodies are independent chains of mostly single-workgroup kernels: three at a time, each on
        # its own pair of streams (pipeline.LanePipeline); the NaN flags of their kNN searches are read once at the end
        flags = []
        prep = prepare_pool(matching_list, split=shard is None) if len(items) > 1 else None
        if shard is None and len(items) > 1:
            # searches over the frames of SEVERAL items at a time (the reference searches 20 rows at a time,
            # ddsp_prematch_dataset.py:1195-1206; rows are independent): a [~3000, 1024] x [Np, 1024] product runs the matrix cores
            # 3-4 x as efficiently as one ~300-row search per utterance
            g_nn, g_ready = grouped_knn(items, query_pool, matching_list, prep, flags)
            nn.update(g_nn); nn_ready.update(g_ready)
        def body(item):
            wait_for_neighbours(nn.get(item), nn_ready.get(item), matching_list.device)      # a group search on the kNN stream
            extra = dict(synth_list=synth_list) if synth_list is not None else {}
            return match_features(query_pool[item], query_f0_pool[item], matching_list, matching_f0,
                                  harmonics_list, ckpt_type, post_opt, nan_flags=flags, pool_prep=prep, nn32=nn.get(item), **extra)
        # match bodies in flight at once (each is a chain of single-workgroup recurrences: more lanes = more of them side by side)
        lanes = min(int(os.environ.get("KNNSVC_MATCH_LANES", "3")), len(items)) if matching_list.is_cuda else 1   # (CPU tensors: injected kernels in the gloo tests)
        if vocode_fn is not None and len(items) > 0:
            assert waves_out is not None
            tail = lambda item, r: r + (vocode_fn(r[0], r[2], r[1]),)
            if matching_list.is_cuda:
                results = pipeline.LanePipeline(matching_list.device, max(1, lanes)).run(items, body, tail)
            else:
                results = [tail(i, body(i)) for i in items]
            for item, r in zip(items, results):
                waves_out[item] = r[3]
            results = [r[:3] for r in results]
        else:
            results = pipeline.LanePipeline(matching_list.device, lanes).run(items, body) if lanes > 1 else [body(i) for i in items]
        for f in flags:
            ops.raise_if_nan(f)
        return items, results

    # a candidate-buffer overflow of the fused kNN route (reported by the deferred flags) repeats the searches and the bodies on
    # the dot-matrix route; under a process group every rank sees it together (dist.raise_if_any_nan)
    items, results = ops.retry_on_overflow(run)
    for item, (of, hw, sf0) in zip(items, results):
        out_c[item] = of; audio_c[item] = None; f0_c[item] = sf0
        if hw is not None:
            harm_c[item] = hw
    if "wavlm_only" in ckpt_type or "no_harm_no_amp" in ckpt_type:
        return out_c, audio_c, f0_c
    return out_c, harm_c, audio_c, f0_c
