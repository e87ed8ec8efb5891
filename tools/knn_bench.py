"""kNN distance + top-32 micro-benchmark: query frames/s and achieved TFLOP/s (HIP events)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from knn_svc_amd import ops, synthetic as S
for nq, npool in ((1500, 30000), (300, 30000), (3000, 30000), (6000, 30000), (3000, 180000), (24000, 180000)):
    q = S.clustered_features(nq, 1024, 1).cuda(); p = S.clustered_features(npool, 1024, 2).cuda()
    qs, ps = ops.row_norms(q), ops.row_norms(p)
    for _ in range(2): ops.knn_topk(q, p, 32, q_stats=qs, p_stats=ps, check_nan=False)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    n = 5
    for _ in range(n): ops.knn_topk(q, p, 32, q_stats=qs, p_stats=ps, check_nan=False)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print(f"Nq={nq:6d} Np={npool:7d}: {ms:8.3f} ms  {nq / ms * 1e3:12.0f} query frames/s  {2.0 * nq * npool * 1024 / ms / 1e9:7.1f} TFLOP/s"
          f"  (fused route: {'on' if ops.knn_fused_on() and nq >= ops.KNN_FUSED_MIN_Q and npool >= ops.KNN_FUSED_MIN_P else 'off'})")
