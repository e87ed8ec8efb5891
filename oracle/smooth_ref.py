"""Oracle: concatenation-smoothness weight optimisation (reference
ddsp_prematch_dataset.py:574-680 ``compute_wavlm_weight``, :807-924
``compute_extended_weight`` and :684-804 ``compute_weight_with_amp``).  Test infrastructure only."""
from __future__ import annotations

import torch


def smooth_weights(idx: torch.Tensor, pool: torch.Tensor, scale: float, max_iter: int = 100000,
                   return_iters: bool = False, row_scale: torch.Tensor | None = None):
    """theta in R^{N x k} from 0; w = softmax(theta); E_s[t] = sum_k w[t,k] pool[clamp(idx[t,k]+s)]
    for s in {-1,0,+1}; loss = mean_t scale*MSE(E_-1[t+1], E_0[t]) + mean_t scale*MSE(E_0[t+1], E_+1[t]).
    Adam(lr .1, betas .9/.999, eps 1e-8, amsgrad).  Every iteration: remember the best theta;
    stop at t % 100 == 1 when the best loss moved < 1e-5 since the previous checkpoint, or
    after 1000 consecutive non-improving iterations, or at ``max_iter``.  Returns softmax(best).

    scale = 0.1 for WavLM features (wavlm_phase_mae, :460-461), 1000 for harmonics
    (phase_mae, :449-457).  The harmonic variant's extra tanh scaling branch is the
    identity because scaling_max == scaling_min == 1 (:836-837, 862).

    ``row_scale`` [N,k] is compute_weight_with_amp's amp_ratio (:684-713): every gathered row
    (shift -1, 0, +1) of candidate k of frame t is multiplied by row_scale[t,k] first; that
    function uses phase_mae, i.e. scale = 1000 (:744-754)."""
    n_pool = len(pool)
    gathered = {}
    for s in (-1, 0, 1):
        j = torch.clamp(idx + s, 0, n_pool - 1)
        gathered[s] = pool[j.reshape(-1)].reshape(idx.shape[0], idx.shape[1], pool.shape[-1])
        if row_scale is not None:
            gathered[s] = gathered[s] * row_scale[:, :, None]
    theta = torch.zeros(idx.shape, dtype=torch.float32, requires_grad=True)
    opt = torch.optim.Adam([theta], lr=1e-1, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, amsgrad=True)
    min_loss = 20000
    conv_min = 20000
    best = theta.detach().clone()
    since_improve = 0
    it = 0
    for t in range(max_iter):
        it = t
        # the reference evaluates softmax(theta) once per shift (three autograd nodes); the
        # gradient accumulation order that follows from it is part of the bit-level behaviour
        e = {s: torch.sum(gathered[s] * torch.softmax(theta, dim=1)[..., None], dim=1) for s in (-1, 0, 1)}
        term_a = scale * torch.mean((e[-1][1:] - e[0][:-1]) ** 2, dim=-1)
        term_b = scale * torch.mean((e[0][1:] - e[1][:-1]) ** 2, dim=-1)
        loss = torch.mean(term_a) + torch.mean(term_b)
        if t % 100 == 1:
            if abs(min_loss - conv_min) < 1e-5:
                break
            conv_min = min_loss
        if loss < min_loss:
            min_loss = loss.item()
            best = theta.detach().clone()
            since_improve = 0
        else:
            since_improve += 1
        if since_improve >= 1000:
            break
        opt.zero_grad()
        loss.backward()
        opt.step()
    out = torch.softmax(best, dim=1)
    return (out, it) if return_iters else out
