"""Debug aid: fused (epoch) route vs dot-matrix route for several grid caps / sizes; prints mismatching rows."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from knn_svc_amd import ops, synthetic as S
P = S.clustered_features(40000, 1024, 5, n_centres=50).cuda()
ps = ops.row_norms(P)
for nq in (300, 900, 1500, 2600):
    q = S.clustered_features(nq, 1024, 40, n_centres=50).cuda()
    qs = ops.row_norms(q)
    os.environ["KNNSVC_KNN_FUSED"] = "0"
    i0, d0 = ops.knn_topk(q, P, 32, q_stats=qs, p_stats=ps)
    os.environ["KNNSVC_KNN_FUSED"] = "1"
    for mb in (0, 192, 64, 8):
        for rep in range(2):
            i1, d1, f = ops.knn_topk(q, P, 32, q_stats=qs, p_stats=ps, check_nan=False, return_flag=True, max_blocks=mb)
            bad = (i0 != i1).any(1) | (d0 != d1).any(1)
            nb = int(bad.sum())
            msg = f"nq={nq} max_blocks={mb} rep={rep}: epochs={ops.knn_epochs(nq, 40000, mb or 256)} flag={int(f.item())} bad rows={nb}"
            if nb:
                r = int(bad.nonzero()[0])
                msg += f" first={r} want={i0[r, :6].tolist()} got={i1[r, :6].tolist()} dwant={d0[r, :3].tolist()} dgot={d1[r, :3].tolist()}"
                msg += f" rows={bad.nonzero().flatten()[:12].tolist()}"
            print(msg, flush=True)
