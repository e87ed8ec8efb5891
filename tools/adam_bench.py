"""Smoothness-weight loop alone (1500 frames): microseconds per Adam iteration with the stopping rules off (2000 iterations),
and the natural iteration count / a checksum of the weights with them on."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from knn_svc_amd import ops, synthetic as S
nq, npool = int(sys.argv[1]) if len(sys.argv) > 1 else 1500, 30000
sm = lambda x: (x + torch.roll(x, 1, 0) + torch.roll(x, 2, 0)) / 3
for name, pool, scale in (("wavlm", sm(S.clustered_features(npool, 1024, 2, n_centres=60)).cuda(), 0.1),
                          ("harm", sm(torch.rand(npool, 49, generator=torch.Generator().manual_seed(3)) * 0.05).cuda(), 1000.0)):
    idx = torch.randint(0, npool, (nq, 4), generator=torch.Generator().manual_seed(5)).cuda()
    w, it = ops.smooth_weights(idx, pool, scale, return_iters=True)
    print(f"{name}: natural run {int(it)} iterations, checksum {float(w.double().sum()):.9f} {float((w.double() ** 2).sum()):.9f}")
    N = 2000
    for _ in range(2): ops.smooth_weights(idx, pool, scale, max_iter=-N)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(3): ops.smooth_weights(idx, pool, scale, max_iter=-N)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    print(f"{name}: {ms:.3f} ms for {N} forced iterations (incl. the Gram pass) = {ms / N * 1e3:.2f} us / iteration")
