"""One kNN search size in a loop (for rocprofv3 --kernel-trace --stats): python tools/knn_prof_one.py NQ NP [iters]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from knn_svc_amd import ops, synthetic as S
nq, npool = int(sys.argv[1]), int(sys.argv[2])
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 20
q = S.clustered_features(nq, 1024, 1).cuda(); p = S.clustered_features(npool, 1024, 2).cuda()
for _ in range(iters):
    qs, ps = ops.row_norms(q), ops.row_norms(p)
    ops.knn_topk(q, p, 32, q_stats=qs, p_stats=ps, check_nan=False)
torch.cuda.synchronize()
