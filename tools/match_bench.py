"""match stage only (1500 queries vs 30000 pool rows, temporally smooth synthetic features)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from knn_svc_amd import synthetic as S
from knn_svc_amd.matching import match_features
dev = "cuda"
nq, npool = 1500, 30000
q = S.clustered_features(nq, 1024, 1, n_centres=60); p = S.clustered_features(npool, 1024, 2, n_centres=60)
sm = lambda x: (x + torch.roll(x, 1, 0) + torch.roll(x, 2, 0)) / 3
q, p = sm(q).to(dev), sm(p).to(dev)
_, f0 = S.synth_clip(npool * 320, 3); pf0 = torch.from_numpy(f0[:npool].copy()).to(dev)
_, f0 = S.synth_clip(nq * 320, 4); qf0 = torch.from_numpy(f0[:nq].copy() * 1.2).to(dev)
harm = (torch.rand(npool, 49) * 0.05).to(dev)
for _ in range(2): out = match_features(q, qf0, p, pf0, harm, "mix", "post_opt_0.2", return_debug=True)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record()
for _ in range(5): out = match_features(q, qf0, p, pf0, harm, "mix", "post_opt_0.2", return_debug=True)
e1.record(); torch.cuda.synchronize()
print(f"match_features: {e0.elapsed_time(e1) / 5:.3f} ms; adam iters {int(out[3]['iters_wavlm'])}/{int(out[3]['iters_harm'])}")
