"""Oracle: neighbour post-processing — f0 shift, f0-compatibility re-rank, greedy
concatenation-cost re-selection (reference ddsp_prematch_dataset.py:954-1016,
1224-1233, 1273-1279; lib_ongaku_test.py:270-369).  Test infrastructure only."""
from __future__ import annotations

import torch

from .knn_ref import cosine_dist


def parse_post_opt(post_opt: str):
    """(concat_weight, run_adam) — ddsp_prematch_dataset.py:1273-1279, 1356."""
    tail = post_opt.split("_")[-1]
    try:
        w = float(tail)
    except ValueError:
        w = 0.3 if tail == "extra" else -1
    return w, ("no_post_opt" not in post_opt)


def shift_query_f0(query_f0: torch.Tensor, pool_f0: torch.Tensor) -> torch.Tensor:
    """Median log-f0 alignment on voiced frames; torch.median = lower median
    (ddsp_prematch_dataset.py:1224-1233)."""
    qm = torch.median(torch.log(query_f0[query_f0 != 0]))
    pm = torch.median(torch.log(pool_f0[pool_f0 != 0]))
    out = query_f0.clone()
    v = query_f0 != 0
    out[v] = torch.exp(torch.log(query_f0[v]) + pm - qm)
    return out


def rerank_by_f0(shifted_f0: torch.Tensor, pool_f0: torch.Tensor, nn_idx: torch.Tensor) -> torch.Tensor:
    """Stable ascending sort of each row's neighbours by |log2(f_nbr+1e-5) - log2(f_q+1e-5)|
    (ddsp_prematch_dataset.py:954-1016)."""
    nf = pool_f0[nn_idx.reshape(-1)].reshape(nn_idx.shape)
    key = torch.abs(torch.log2(nf + 1e-5) - torch.log2(shifted_f0[:, None] + 1e-5))
    order = torch.sort(key, dim=1, descending=False, stable=True).indices
    return nn_idx.gather(1, order)


def concat_reselect(idx: torch.Tensor, q: torch.Tensor, pool: torch.Tensor,
                    shifted_f0: torch.Tensor | None = None, pool_f0: torch.Tensor | None = None,
                    concat_weight: float = 0.2) -> torch.Tensor:
    """Frame-sequential greedy re-selection (lib_ongaku_test.py:270-369).

    Frame 0 keeps its row.  For frame i the candidates are its own k
    neighbours followed by (previous selection + 1, clamped to the pool end);
    duplicates stay.  cost = w * median_over_prev(concat) + match (+ |dlog2 f0|),
    with the variant-specific thresholding of the concat matrix; the k
    smallest costs (torch.topk order) become the selection.  In the f0
    variant w drops to 0 *permanently* the first time the source step
    baseline reaches 0.08."""
    k = idx.shape[1]
    n_pool = len(pool)
    sel = [idx[0]]
    use_f0 = shifted_f0 is not None
    if use_f0:
        lf_q = torch.log2(shifted_f0.to(q) + 1e-5)
        lf_p = torch.log2(pool_f0.to(pool) + 1e-5)
    w = concat_weight
    for i in range(1, len(q)):
        extra = torch.clamp(sel[-1] + 1, max=n_pool - 1)
        cand = torch.cat([idx[i], extra])
        cf = pool[cand]
        match = cosine_dist(q[i][None], cf)                    # [1, 2k]
        cc = cosine_dist(pool[sel[-1]], cf)                    # [k, 2k]
        base = cosine_dist(q[i - 1][None], q[i][None])[0, 0] * 2
        if use_f0:
            pitch = torch.abs(lf_p[cand][None] - lf_q[i])
            if base < 0.08:
                cc = torch.where(cc < 5 * base, torch.zeros_like(cc), cc)
            else:
                w = 0
            total = w * torch.median(cc, dim=0, keepdim=True).values + match + pitch
        else:
            cc = torch.where(cc > base, 1.5 * cc - base, cc)
            total = w * torch.median(cc, dim=0, keepdim=True).values + match
        pick = total.topk(k=k, dim=-1, largest=False).indices[0]
        sel.append(cand[pick])
    return torch.stack(sel)
