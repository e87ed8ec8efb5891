"""Oracle: training-pool generation ("prematch"), reference ddsp_prematch_dataset.py:1464-1772
``per_spk_extract`` on in-memory per-utterance features.  Test infrastructure only — nothing on the
product path imports this module.

What per_spk_extract writes per speaker folder (everything else it computes is discarded):
  pool.npy            concatenated layer-6 features, rounded through fp16 (:1509, 1530)
  pool_harmonics.npy  concatenated harmonic amplitudes (:1532)
  (save_pool_only)    pool_f0.npy, pool_spec.npy (:1590-1591)
and per utterance a pickled dict `<utt>.pt` with
  slice                        (start, end) rows of the utterance inside the pool (:1582)
  nearest_nbrs                 [T,32] self-kNN with the utterance's own rows forced to distance 1 (:1596-1617)
  nearest_nbrs_f0_priority     stable re-sort by |log2 f0| distance to the frame's own f0 (:1634)
  amp_ratio                    [T,4] L1(spec[t]) / (L1(spec_pool[nbr]) + 1e-5) over the first 4 f0-priority nbrs (:1657-1660)
  harmonics_best_weight_para   [T,4] compute_weight_with_amp on the harmonics with that amp_ratio (:1665)

Reference quirk: as committed, per_spk_extract passes (ls_path, device) into get_complete_spk_pool's
(device, duration_limit) slots (:1490 vs :301) and therefore raises before doing any work.  The golden vectors were
produced by the reference function with that one call mapped back (tests/gen_golden.py, gen_prematch)."""
from __future__ import annotations

import torch

from . import knn_ref, select_ref, smooth_ref


def round_f16(x: torch.Tensor) -> torch.Tensor:
    return x.half().float()


def self_knn(item_feats: torch.Tensor, pool_h: torch.Tensor, start: int, end: int, k: int = 32, rows: int = 20):
    """20 query rows at a time; the utterance's own pool rows get distance exactly 1 before topk (:1596-1617)."""
    out = []
    for s in range(0, len(item_feats), rows):
        d = knn_ref.cosine_dist(round_f16(item_feats[s:s + rows]), pool_h)
        d[:, start:end] = 1
        out.append(d.topk(k=k, dim=-1, largest=False).indices)
    return torch.cat(out, 0)


def amp_ratio(spec_item: torch.Tensor, spec_pool: torch.Tensor, idx4: torch.Tensor) -> torch.Tensor:
    orig = spec_item.norm(dim=1, p=1)
    g = spec_pool[idx4.reshape(-1)].reshape(idx4.shape[0], idx4.shape[1], spec_pool.shape[-1])
    return orig[:, None] / (g.norm(dim=-1, p=1) + 1e-5)


def extract_speaker(utts: list, max_iter: int = 100000) -> dict:
    """utts: per-utterance dicts {feats [T,D], spec [T,200], f0 [T], harm [T,49]} in pool order
    (oracle.pipeline_ref.utterance_features).  Returns {pool, pool_harmonics, pool_f0, pool_spec, items: [...]}."""
    starts = [0]
    for u in utts:
        starts.append(starts[-1] + len(u["feats"]))
    pool_h = round_f16(torch.cat([u["feats"] for u in utts], 0))
    spec_l = torch.cat([u["spec"] for u in utts], 0)
    f0_l = torch.cat([u["f0"] for u in utts], 0)
    harm_l = torch.cat([u["harm"] for u in utts], 0)
    items = []
    for k, u in enumerate(utts):
        s, e = starts[k], starts[k + 1]
        nn = self_knn(u["feats"], pool_h, s, e)
        nn_f0 = select_ref.rerank_by_f0(u["f0"], f0_l, nn)
        idx4 = nn_f0[:, :4].clone()
        ar = amp_ratio(u["spec"], spec_l, idx4)
        w = smooth_ref.smooth_weights(idx4, harm_l, 1000.0, max_iter=max_iter, row_scale=ar)
        items.append(dict(slice=(s, e), nearest_nbrs=nn, nearest_nbrs_f0_priority=nn_f0, amp_ratio=ar,
                          harmonics_best_weight_para=w))
    return dict(pool=pool_h, pool_harmonics=harm_l, pool_f0=f0_l, pool_spec=spec_l, items=items)
