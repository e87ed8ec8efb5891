"""Oracle: cosine-distance kNN (reference lib_ongaku_test.py:148-175 and the
driver loop ddsp_prematch_dataset.py:1195-1210).  Test infrastructure only."""
from __future__ import annotations

import numpy as np
import torch


def cosine_dist(q: torch.Tensor, p: torch.Tensor) -> torch.Tensor:
    """One call of fast_cosine_dist on <= 20 query rows:
        d = 1 - ((-cdist(q,p)^2 + |q|^2 + |p|^2) / 2) / (|q| |p|)
    torch.cdist picks its mm route when either side has > 25 rows
    (sqrt(clamp_min([-2q, |q|^2, 1] @ [p, 1, |p|^2]^T, 1e-30))) and the direct
    sqrt(sum((x-y)^2)) route otherwise — both are reached on the path (pool
    vs 8 candidates).  NaN => the reference exits; here it raises."""
    qn = torch.norm(q, p=2, dim=-1)
    pn = torch.norm(p, p=2, dim=-1)
    dot = -torch.cdist(q[None], p[None], p=2)[0] ** 2 + qn[:, None] ** 2 + pn[None] ** 2
    dot = dot / 2
    d = 1 - dot / (qn[:, None] * pn[None])
    if torch.isnan(d).any():
        raise FloatingPointError("containing nan")
    return d


def cosine_dist_all(q: torch.Tensor, p: torch.Tensor, rows: int = 20) -> torch.Tensor:
    """fast_cosine_dist's internal 20-row stepping (lib_ongaku_test.py:154-175)."""
    return torch.cat([cosine_dist(q[s:s + rows], p) for s in range(0, len(q), rows)], 0)


def knn_topk(q: torch.Tensor, pool: torch.Tensor, k: int = 32, rows: int = 20):
    """Ascending-distance top-k per query row, computed 20 rows at a time
    (ddsp_prematch_dataset.py:1195-1210).  Returns (idx int64 [Nq,k], dist f32 [Nq,k])."""
    idx, val = [], []
    for s in range(0, len(q), rows):
        t = cosine_dist(q[s:s + rows], pool).topk(k=k, dim=-1, largest=False)
        idx.append(t.indices)
        val.append(t.values)
    return torch.cat(idx, 0), torch.cat(val, 0)


def cosine_dist_f64(q: torch.Tensor, p: torch.Tensor) -> np.ndarray:
    """Mathematically exact (fp64) cosine distance; used to measure whether an index
    mismatch sits inside an fp32 rounding gap (SURVEY.md §7 hard part 1)."""
    qd = q.double().numpy()
    pd = p.double().numpy()
    qn = np.linalg.norm(qd, axis=1)
    pn = np.linalg.norm(pd, axis=1)
    return 1.0 - (qd @ pd.T) / (qn[:, None] * pn[None])


def topk_agreement(idx_a: torch.Tensor, idx_b: torch.Tensor, dist_f64: np.ndarray, tau: float = 5e-7):
    """Parity statistics between two [Nq,k] index sets.

    Returns dict with: exact row-match rate for the first 4 and all k columns
    (ordered), set-match rate, and ``max_gap`` = the largest fp64 distance
    inversion any mismatch implies (a mismatch is *explained* when the two
    candidates' exact distances differ by <= tau, i.e. they sit inside one
    fp32 rounding gap of the reference formula)."""
    a = idx_a.numpy().astype(np.int64)
    b = idx_b.numpy().astype(np.int64)
    nq, k = a.shape
    top4 = float(np.mean(np.all(a[:, :4] == b[:, :4], axis=1)))
    allk = float(np.mean(np.all(a == b, axis=1)))
    sets = float(np.mean([set(a[i]) == set(b[i]) for i in range(nq)]))
    da = np.take_along_axis(dist_f64, a, axis=1)
    db = np.take_along_axis(dist_f64, b, axis=1)
    # position-wise exact-distance difference: identical rankings give 0
    max_gap = float(np.max(np.abs(da - db)))
    unexplained = int(np.sum(np.abs(da - db) > tau))
    return dict(top4=top4, allk=allk, sets=sets, max_gap=max_gap, unexplained=unexplained)
