"""concat_reselect at the north-star size (1500 frames vs a 30 000-row pool, D = 1024): ms per call; with a library built with
-DKN_CONCAT_PROF (KNNSVC_LIB=...) the kernel prints where a frame's cycles go."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from knn_svc_amd import ops, synthetic as S
q = S.clustered_features(1500, 1024, 1, n_centres=80).cuda(); p = S.clustered_features(30000, 1024, 2, n_centres=80).cuda()
idx, _ = ops.knn_topk(q, p, 32)
qn, _ = ops.row_norms(q); pn, _ = ops.row_norms(p)
idx4 = idx[:, :4].contiguous()
for use_f0 in (False, True):
    f0q = (torch.rand(1500) * 200 + 100).cuda() if use_f0 else None
    f0p = (torch.rand(30000) * 200 + 100).cuda() if use_f0 else None
    for _ in range(2): out = ops.concat_reselect(idx4, q, qn, p, pn, f0q, f0p)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(5): out = ops.concat_reselect(idx4, q, qn, p, pn, f0q, f0p)
    e1.record(); torch.cuda.synchronize()
    print(f"use_f0={use_f0}: {e0.elapsed_time(e1) / 5:.3f} ms per call = {e0.elapsed_time(e1) / 5 / 1500 * 1e3:.2f} us/frame")
