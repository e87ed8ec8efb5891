"""Training-pool generation ("prematch") on the GPU — host mirror of the reference's
``per_spk_extract`` and of its command line (ddsp_prematch_dataset.py:1464-1772, 1776-1812).

Per speaker folder (any leaf folder under ``ls_path`` that holds .wav/.flac files) it writes
``pool.npy`` (layer-6 features rounded through fp16) and ``pool_harmonics.npy`` — plus ``pool_f0.npy`` /
``pool_spec.npy`` when ``save_pool_only`` — and per utterance a pickled dict ``<utt>.pt`` with ``slice``,
``nearest_nbrs`` [T,32] (self-kNN, the utterance's own rows forced to distance 1), ``nearest_nbrs_f0_priority``,
``amp_ratio`` [T,4] and ``harmonics_best_weight_para`` [T,4]: the files ``hifigan/ddsp_meldataset.py:473-499``
reads back.  Same paths, same keys, same dtypes (int64 / float32 numpy arrays, protocol = HIGHEST).

Differences from the reference, all deliberate:
* the reference passes ``(ls_path, device)`` into ``get_complete_spk_pool``'s ``(device, duration_limit)``
  parameters (:1490 vs :301) and cannot run as committed; the call is made with the intended arguments;
* speaker folders are visited in sorted order (the reference iterates a ``set``);
* the gathers whose results the reference throws away (``out_feats``, ``audio_out_feats``, the weighted
  harmonic sum, :1637-1674) are not computed; ``audio_synth_pool`` is therefore never built;
* under ``torch.distributed`` (one process per GPU) the speaker folders are dealt round-robin over the ranks;
* every utterance of a speaker is a kNN launch against the speaker's pool whose split image (f16x2 operand of
  the matrix-core GEMM) is built once per speaker, utterances run on the stream scheduler's lanes, and the
  Adam loop runs on the device without host round trips.
"""
from __future__ import annotations

import argparse
import os
import pickle
from pathlib import Path

import numpy as np
import torch

from . import config as C, dist as kdist, ops, pipeline
from .matching import get_complete_spk_pool


def speaker_folders(ls_path) -> list:
    """Leaf folders that contain .wav / .flac files (:1469-1473; .mp3 is not globbed there), sorted."""
    ls_path = Path(ls_path)
    files = list(ls_path.glob("**/*.wav")) + list(ls_path.glob("**/*.flac"))
    return sorted(set(f.parent for f in files))


def match_speaker(matching_pool: dict, spec_pool: dict, f0_pool: dict, harm_pool: dict, lanes: int = 3) -> dict:
    """The per-speaker body on device tensors.  Returns dict(pool, pool_harmonics, pool_f0, pool_spec, items) with
    items[i] = dict(slice, nearest_nbrs, nearest_nbrs_f0_priority, amp_ratio, harmonics_best_weight_para, iters)."""
    keys = list(matching_pool)
    starts = [0]
    for k in keys:
        starts.append(starts[-1] + matching_pool[k].shape[0])
    pool_h = ops.round_f16(torch.cat([matching_pool[k] for k in keys], 0))          # synth_list == matching_list_1 (:1509, 1561)
    spec_l = torch.cat([spec_pool[k] for k in keys], 0).contiguous()
    f0_l = torch.cat([f0_pool[k] for k in keys], 0).contiguous()
    harm_l = torch.cat([harm_pool[k] for k in keys], 0).contiguous()
    p_stats = ops.row_norms(pool_h)
    prepared = ops.prepare_knn_pool(pool_h, C.KNN_K)      # split image of the pool: once per speaker

    def run():
        flags = []

        def body(i):
            key = keys[i]
            q = ops.round_f16(matching_pool[key])                                       # each 20-row block is rounded (:1596)
            nn, _, flag = ops.knn_topk(q, pool_h, C.KNN_K, p_stats=p_stats, check_nan=False, return_flag=True,
                                       mask=(starts[i], starts[i + 1]), prepared=prepared)
            flags.append(flag)
            nn_f0 = ops.f0_rerank(nn, f0_pool[key], f0_l)                               # the frame's OWN f0, unshifted (:1634)
            idx4 = nn_f0[:, :C.KNN_USE].contiguous()
            ar = ops.amp_ratio(spec_pool[key], spec_l, idx4)
            w, it = ops.smooth_weights(idx4, harm_l, 1000.0, return_iters=True, row_scale=ar)
            return dict(slice=(starts[i], starts[i + 1]), nearest_nbrs=nn, nearest_nbrs_f0_priority=nn_f0, amp_ratio=ar,
                        harmonics_best_weight_para=w, iters=it)

        n_l = max(1, min(lanes, len(keys)))
        if n_l > 1:
            items = pipeline.LanePipeline(pool_h.device, n_l).run(range(len(keys)), body)
        else:
            items = [body(i) for i in range(len(keys))]
        for f in flags:
            ops.raise_if_nan(f)
        return items

    items = ops.retry_on_overflow(run)
    return dict(pool=pool_h, pool_harmonics=harm_l, pool_f0=f0_l, pool_spec=spec_l, items=items, keys=keys)


def per_spk_extract(wavlm, device, ls_path, out_path, synth_weights=None, match_weights=None, save_pool_only=False):
    """Same contract as the reference function (:1464)."""
    ls_path, out_path = Path(ls_path), Path(out_path)
    all_folders = speaker_folders(ls_path)
    folders = kdist.my_share(all_folders)          # one process per GPU: speakers are independent, no collective
    for i, folder in enumerate(folders):
        matching_pool, _synth, _audio, spec_pool, f0_pool, harm_pool = get_complete_spk_pool(
            folder, wavlm, match_weights, synth_weights, device)
        cache = out_path / folder.relative_to(ls_path)
        os.makedirs(cache, exist_ok=True)
        if save_pool_only:
            keys = list(matching_pool)
            pool_h = ops.round_f16(torch.cat([matching_pool[k] for k in keys], 0))
            res = dict(pool=pool_h, pool_harmonics=torch.cat([harm_pool[k] for k in keys], 0),
                       pool_f0=torch.cat([f0_pool[k] for k in keys], 0), pool_spec=torch.cat([spec_pool[k] for k in keys], 0),
                       keys=keys, items=None)
        else:
            res = match_speaker(matching_pool, spec_pool, f0_pool, harm_pool)
        np.save(str(cache / "pool.npy"), res["pool"].cpu().numpy())
        np.save(str(cache / "pool_harmonics.npy"), res["pool_harmonics"].cpu().numpy())
        start = 0
        for k, item in enumerate(res["keys"]):
            end = start + matching_pool[item].shape[0]
            target = out_path / Path(item).relative_to(ls_path).with_suffix(".pt")
            os.makedirs(target.parent, exist_ok=True)
            if os.path.isfile(target):
                with open(target, "rb") as fh:
                    feats = pickle.load(fh)
                assert tuple(feats["slice"]) == (start, end), (feats["slice"], (start, end))
            else:
                feats = {"slice": (start, end)}
            if save_pool_only:
                np.save(str(cache / "pool_f0.npy"), res["pool_f0"].cpu().numpy())
                np.save(str(cache / "pool_spec.npy"), res["pool_spec"].cpu().numpy())
            else:
                it = res["items"][k]
                assert it["slice"] == (start, end)
                feats["nearest_nbrs"] = it["nearest_nbrs"].cpu().numpy()
                feats["nearest_nbrs_f0_priority"] = it["nearest_nbrs_f0_priority"].cpu().numpy()
                feats["harmonics_best_weight_para"] = it["harmonics_best_weight_para"].cpu().numpy()
                feats.pop("best_weights", None)
                feats["amp_ratio"] = it["amp_ratio"].cpu().numpy()
            with open(target, "wb") as fh:
                pickle.dump(feats, fh, protocol=pickle.HIGHEST_PROTOCOL)
            start = end
        print(i, "/", len(folders), "/".join(str(folder).split("/")[-3:]), flush=True)


def main(argv=None):
    """``python ddsp_prematch_dataset.py --librispeech_path … --out_path … --prematch`` (:1815-1831)."""
    ap = argparse.ArgumentParser(description="Compute matched wavlm features for a librispeech dataset")
    ap.add_argument("--librispeech_path", required=True, type=str)
    ap.add_argument("--seed", default=123, type=int)
    ap.add_argument("--out_path", required=True, type=str)
    ap.add_argument("--device", default="cuda", type=str)
    ap.add_argument("--topk", type=int, default=4)
    ap.add_argument("--matching_layer", type=int, default=6)
    ap.add_argument("--synthesis_layer", type=int, default=6)
    ap.add_argument("--prematch", action="store_true", help="prematch")
    ap.add_argument("--resume", action="store_true")
    ap.add_argument("--include_cross_nbrs", type=bool, default=False)
    ap.add_argument("--save_pool_only", action="store_true", help="(build extension) only write the pool_*.npy files")
    a = ap.parse_args(argv)
    if a.matching_layer != a.synthesis_layer:
        raise NotImplementedError("matching and synthesis layers must be the same exit layer")
    from .hubconf import wavlm_large
    np.random.seed(a.seed)
    torch.manual_seed(a.seed)
    wavlm = wavlm_large(pretrained=True, progress=True, device="cuda" if a.device == "cpu" else a.device,
                        n_layers=a.matching_layer)
    onehot = torch.zeros(wavlm.cfg["encoder_layers"] + 1)
    onehot[a.matching_layer] = 1
    with torch.inference_mode():
        per_spk_extract(wavlm, a.device, Path(a.librispeech_path), Path(a.out_path), onehot[:, None], onehot[:, None],
                        save_pool_only=a.save_pool_only)
    print("All done!", flush=True)
    return 0
