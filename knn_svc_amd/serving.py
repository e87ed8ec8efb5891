"""Many-to-one batched conversion and a request queue (BASELINE cfg 5: many concurrent source clips against one
resident target pool).  The reference has no equivalent: its loops convert one utterance at a time and rebuild the target pool on
every call (ddsp_matcher.py:1073-1150, ddsp_prematch_dataset.py:1086-1134).  Everything here is host-side orchestration of the same
kernels, in the same order per utterance, as ``special_match`` / ``bulk_match`` — an utterance converted in a batch equals the same
utterance converted alone up to the rounding of the batch-wide power-of-two operand scales of the f16x2 GEMMs (range slots): bit
for bit with the small test models and on most full-size sources, <= 1e-6 otherwise (``tests/test_gpu_product.py``).

  TargetVoice      the pool of one target speaker, resident in HBM: features / f0 / harmonics, row norms, the f16x2 image the
                   kNN reads — built once (get_complete_spk_pool), reused by every request.
  BatchConverter   one batch of sources through the pipeline: batched encoder chunks (hipGraph buckets), batched f0 / side
                   features, kNN searches over the frames of several sources at a time on a stream of their own
                   (matching.grouped_knn), match bodies on lane streams, the generator as the tail stage (pipeline.LanePipeline).
                   No host synchronisation between the first launch and the last; the deferred kNN flags are read once.
  RequestQueue     dynamic batching for a serving process: ``submit`` returns a Future; one worker thread drains the queue into
                   batches (up to ``max_batch`` requests, waiting at most ``max_wait_ms`` for the batch to fill) and runs them
                   through a BatchConverter.  One worker per GPU: a hipGraph owns its buffers, so the same encoder / generator
                   graph must not be replayed from two host threads.

Several GPUs (one process per GPU): sources are independent — deal them over the ranks (``dist.my_share``) and give every rank
the same TargetVoice; when the pool itself is too large or too slow to encode on one GPU, use
``match_at_inference_time(pool_sharded=True, share_items=True)`` (``bulk_match`` with KNNSVC_POOL_SHARD=1), which shards the pool
rows and merges the per-shard lists over RCCL.
"""
from __future__ import annotations

import os
import queue
import threading
from concurrent.futures import Future
from pathlib import Path

import numpy as np
import torch

from . import audio_io, config as C, ops, pipeline
from . import matching as M


class TargetVoice:
    """The resident pool of one target speaker (a file, or a folder of files; ``duration_limit`` in seconds as --dur_limit)."""

    def __init__(self, vc, ref_path, duration_limit=None):
        self.ref_path = str(ref_path)
        self.ref_id = os.path.basename(self.ref_path).split(".")[0]
        with torch.inference_mode():
            vc.wavlm.set_layer_mix(M._mix_of(vc.weighting, vc.wavlm))
            mp, _s, _a, _spec, f0p, hp = M.get_complete_spk_pool(Path(ref_path), vc.wavlm, device=vc.device,
                                                                 duration_limit=duration_limit)
            keys = list(mp)
            self.files = keys
            self.feats = torch.cat([mp[k] for k in keys], 0).contiguous()
            self.f0 = torch.cat([f0p[k] for k in keys], 0).contiguous()
            self.harm = torch.cat([hp[k] for k in keys], 0).contiguous()
            if self.feats.shape[0] < C.KNN_K:
                raise ops.KnnSvcError(f"target pool has {self.feats.shape[0]} frames; the search needs at least {C.KNN_K}")
            self.prep = M.prepare_pool(self.feats)           # row norms + the split image of the kNN GEMM: once per voice

    @classmethod
    def from_clips(cls, vc, clips, name: str = "target"):
        """The same pool from audio already in memory: clips = [(wav [L] float32 16 kHz mono, f0 [L // 320 + 1]), ...]."""
        self = cls.__new__(cls)
        self.ref_path, self.ref_id, self.files = name, name, [f"{name}#{i}" for i in range(len(clips))]
        dev = vc.device
        with torch.inference_mode():
            vc.wavlm.set_layer_mix(M._mix_of(vc.weighting, vc.wavlm))
            g = lambda a: (a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))).to(dev)
            feats, f0s, harms = [], [], []
            for b in range(0, len(clips), 32):
                ws = [g(w).reshape(-1) for w, _ in clips[b:b + 32]]
                fs = vc.wavlm.encode_many(ws, max_batch=32, pow2_batches=True)
                sides = M.side_features_many(ws, [f for _, f in clips[b:b + 32]], [ft.shape[0] for ft in fs])
                feats += fs; f0s += [t[0] for t in sides]; harms += [t[1] for t in sides]
            from .wavlm import cat_rows
            self.feats = cat_rows(feats).contiguous()
            self.f0, self.harm = (torch.cat(x, 0).contiguous() for x in (f0s, harms))
            if self.feats.shape[0] < C.KNN_K:
                raise ops.KnnSvcError(f"target pool has {self.feats.shape[0]} frames; the search needs at least {C.KNN_K}")
            self.prep = M.prepare_pool(self.feats)
        return self

    @property
    def frames(self) -> int:
        return int(self.feats.shape[0])


class BatchConverter:
    def __init__(self, vc, target: TargetVoice, ckpt_type: str = "mix", post_opt: str = "post_opt_0.2", lanes: int | None = None,
                 max_encode_batch: int = 32):
        if "wavlm_only_original" in ckpt_type:
            raise NotImplementedError("wavlm_only_original needs hifigan/models.py, absent upstream")
        self.vc, self.target, self.ckpt_type, self.post_opt = vc, target, ckpt_type, post_opt
        self.f0only = "wavlm_only" in ckpt_type or "no_harm_no_amp" in ckpt_type
        self.lanes = lanes if lanes is not None else int(os.environ.get("KNNSVC_MATCH_LANES", "3"))
        self.max_encode_batch = max_encode_batch

    def _load(self, src, check=False):
        """A request is a path, or (wav [L] float32 16 kHz mono as array / tensor, f0 [L // 320 + 1] or None).
        -> (wav on the device, f0 host array or device tensor)."""
        dev = self.vc.device
        if isinstance(src, (str, os.PathLike)):
            w, f0 = M.load_utterance(src)                    # resamples, reads or computes + caches <stem>_f0.npy
            return torch.from_numpy(w).to(dev), f0
        w, f0 = src
        w = w if isinstance(w, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(w, dtype=np.float32))
        # checked on the RAW input, before anything flattens it or runs on it: a [2, L] stereo array would otherwise be converted
        # as one 2L-sample mono clip, and Harvest would run on NaN samples before the finiteness check
        if w.dim() == 2 and 1 in w.shape:
            w = w.reshape(-1)                                # [1, L] / [L, 1]: mono with a channel axis (torchaudio.load's shape)
        if w.dim() != 1:
            raise ValueError(f"request: expected a mono waveform [L], got shape {tuple(w.shape)}")
        w = w.to(dev).float()
        if check and not bool(torch.isfinite(w).all()):      # (one small host read per request: a server checks its inputs at the door)
            raise ValueError("request: the waveform contains NaN or infinite samples")
        if f0 is None:
            f0 = ops.f0_harvest(w)                           # Harvest on the GPU (csrc/harvest.hip), as load_utterance does
        elif not isinstance(f0, torch.Tensor):
            f0 = np.ascontiguousarray(f0, dtype=np.float32)
        return w, f0

    def load_checked(self, src):
        """_load + the checks that do not need the encoder: a request that fails them fails alone (RequestQueue), before it can
        take a batch down with it.  -> (wav, f0)."""
        w, f0 = self._load(src, check=True)          # shape and finiteness are checked on the raw input, before Harvest runs on it
        if isinstance(src, (str, os.PathLike)) and not bool(torch.isfinite(w).all()):
            raise ValueError("request: the waveform contains NaN or infinite samples")
        T = M.frames_of(int(w.shape[0]), self.vc.wavlm)
        if T < 1:
            raise ValueError(f"request: expected a mono waveform of at least one frame, got shape {tuple(w.shape)}")
        if not (len(f0) >= T and len(f0) - T <= 1):
            raise ValueError(f"request: f0 has {len(f0)} frames for a waveform of {T} frames (expected {T} or {T + 1})")
        return w, f0

    @torch.inference_mode()
    def convert(self, sources, loaded=None) -> list:
        """-> one waveform tensor [T_i * 320] per source, in order.  ``loaded``: the sources already through load_checked."""
        if len(sources) == 0:
            return []
        vc, tg = self.vc, self.target
        dev = vc.device
        vc.wavlm.set_layer_mix(M._mix_of(vc.weighting, vc.wavlm))      # the shared encoder may have been left in another mix
        loaded = loaded if loaded is not None else [self._load(s) for s in sources]
        wavs = [w for w, _ in loaded]
        feats = vc.wavlm.encode_many(wavs, max_batch=self.max_encode_batch, pow2_batches=True)
        f0s = []
        for (w, f0), ft in zip(loaded, feats):
            T = ft.shape[0]
            if not (abs(len(f0) - T) <= 1 and len(f0) >= T):
                raise ValueError(f"f0 has {len(f0)} frames for {T} feature frames")
            f0s.append(f0[:T].contiguous().to(dev) if isinstance(f0, torch.Tensor)
                       else torch.from_numpy(np.ascontiguousarray(f0[:T])).to(dev, non_blocking=True))
        items = list(range(len(sources)))
        qpool = dict(enumerate(feats))
        # the reference does not forward post_opt to the f0-only generators (ddsp_matcher.py:970, 1102-1110)
        post_opt = "no_post_opt" if self.f0only else self.post_opt
        voc = (lambda c, f0, h: vc._vocode_async(c, f0)) if self.f0only else vc._vocode_async

        def run():
            flags = []
            if len(items) > 1:
                nn, ready = M.grouped_knn(items, qpool, tg.feats, tg.prep, flags)
            else:
                nn, ready = {}, {}

            def body(i):
                M.wait_for_neighbours(nn.get(i), ready.get(i), dev)
                return M.match_features(qpool[i], f0s[i], tg.feats, tg.f0, tg.harm, self.ckpt_type, post_opt,
                                        nan_flags=flags, pool_prep=tg.prep, nn32=nn.get(i))
            tail = lambda i, r: voc(r[0], r[2], r[1])
            n_lanes = max(1, min(self.lanes, len(items)))
            ys = pipeline.LanePipeline(dev, n_lanes).run(items, body, tail)
            peak = torch.stack([y.abs().max() for y in ys])
            for f in flags:
                ops.raise_if_nan(f)                          # one host read per search, after everything is enqueued
            vc._check_finite(peak)
            return ys
        return ops.retry_on_overflow(run)

    def convert_files(self, src_files, converted_audio_dir=None) -> list:
        """Paths in, files out: ``<dir or the source's folder>/<src>_to_<ref>_knn_<ckpt_type>_<post_opt>.wav`` (the single-file
        naming, ddsp_matcher.py:1015), PCM_32 16 kHz mono.  -> the written paths."""
        ys = self.convert([str(p) for p in src_files])
        out = []
        for p, y in zip(src_files, ys):
            d = converted_audio_dir if converted_audio_dir is not None else str(Path(p).parent)
            Path(d).mkdir(parents=True, exist_ok=True)
            name = os.path.basename(str(p)).split(".")[0] + "_to_" + self.target.ref_id + f"_knn_{self.ckpt_type}_{self.post_opt}.wav"
            out.append(audio_io.save_audio(os.path.join(d, name), y.detach().cpu().numpy(), sample_rate=16000))
        return out


class RequestQueue:
    """Dynamic batching in front of a BatchConverter.  ``submit(src)`` -> Future of the waveform (a CPU tensor)."""

    def __init__(self, converter: BatchConverter, max_batch: int = 32, max_wait_ms: float = 5.0):
        self.conv, self.max_batch, self.max_wait = converter, int(max_batch), float(max_wait_ms) / 1e3
        self._q = queue.Queue()
        self._stop = False
        self.batches = []                                    # sizes of the batches run so far (observability / tests)
        self.isolated = 0                                    # batches that failed as a whole and were re-run request by request
        self._th = threading.Thread(target=self._loop, name="knnsvc-batcher", daemon=True)
        self._th.start()

    def submit(self, src) -> Future:
        if self._stop:
            raise RuntimeError("RequestQueue is closed")
        fut = Future()
        self._q.put((src, fut))
        return fut

    def _loop(self):
        dev = self.conv.vc.device
        if dev.type == "cuda" and dev.index is not None:     # a new host thread starts on device 0
            torch.cuda.set_device(dev)
        import time
        while True:
            first = self._q.get()
            if first is None:
                return
            batch = [first]
            deadline = time.monotonic() + self.max_wait       # ONE deadline per batch, set by its first request: a trickle of
            try:                                              # arrivals cannot hold it longer than max_wait_ms
                while len(batch) < self.max_batch:
                    nxt = self._q.get(timeout=max(0.0, deadline - time.monotonic()))
                    if nxt is None:
                        self._q.put(None)                    # finish this batch, then stop
                        break
                    batch.append(nxt)
            except queue.Empty:
                pass
            self.batches.append(len(batch))
            self._run(batch)

    def _run(self, batch):
        """One batch; a request that cannot be loaded, or whose data makes the conversion fail (a NaN source, an f0 track of the
        wrong length), fails ALONE: the others of its batch are converted without it."""
        good, loaded = [], []
        for s, fut in batch:
            try:
                loaded.append(self.conv.load_checked(s))
                good.append((s, fut))
            except BaseException as e:
                fut.set_exception(e)
        if not good:
            return
        try:
            ys = self.conv.convert([s for s, _ in good], loaded=loaded)
            for (_s, fut), y in zip(good, ys):
                fut.set_result(y.detach().cpu())
            return
        except BaseException as e:
            if len(good) == 1:
                good[0][1].set_exception(e)
                return
        self.isolated += 1
        for (s, fut), ld in zip(good, loaded):                # the batch failed as a whole: find the offender(s) one by one
            try:
                fut.set_result(self.conv.convert([s], loaded=[ld])[0].detach().cpu())
            except BaseException as e:
                fut.set_exception(e)

    def close(self):
        self._stop = True
        self._q.put(None)
        self._th.join()
