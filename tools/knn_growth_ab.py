import sys, os
sys.path.insert(0, "/root/repo")
import torch
from knn_svc_amd import ops, synthetic as S
nq, npool = 24000, 180000
q = S.clustered_features(nq, 1024, 1).cuda(); p = S.clustered_features(npool, 1024, 2).cuda()
qs, ps = ops.row_norms(q), ops.row_norms(p)
for g in (4, 8, 16, 4, 8):
    ops.KNN_EPOCH_GROWTH = g
    for _ in range(2): ops.knn_topk(q, p, 32, q_stats=qs, p_stats=ps, check_nan=False)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(5): ops.knn_topk(q, p, 32, q_stats=qs, p_stats=ps, check_nan=False)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(f"growth {g}: epochs {ops.knn_epochs(nq, npool)}: {ms:.3f} ms {2.0 * nq * npool * 1024 / ms / 1e9:.1f} TFLOP/s", flush=True)
