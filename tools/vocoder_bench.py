"""Vocoder-only timing: graph replay ms per 30 s, and an eager per-launch table (HIP events)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from knn_svc_amd import config as C, ops, synthetic as S
from knn_svc_amd.vocoder import Vocoder
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
dev = "cuda"
voc = Vocoder(S.seeded_state(S.generator_param_spec(C.HIFIGAN_V1, "mix"), 2), C.HIFIGAN_V1, "mix", dev)
g = torch.Generator().manual_seed(0)
c = torch.randn(N, 1024, generator=g).to(dev); harm = (torch.rand(N, 49, generator=g) * 0.02).to(dev)
_, f0 = S.synth_clip(N * 320, 5); f0 = torch.from_numpy(f0[:N].copy()).to(dev)
for _ in range(2): y = voc.forward(c, f0, harm)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record()
for _ in range(5): y = voc.forward(c, f0, harm)
e1.record(); torch.cuda.synchronize()
print(f"graph replay: {e0.elapsed_time(e1) / 5:.3f} ms per {N} frames")
recs = []
orig = ops.conv_gemm
def wrapped(x, w, out, **kw):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); r = orig(x, w, out, **kw); b.record()
    recs.append((a, b, kw["m"], kw["n"], kw["cin"], kw.get("taps", 1), kw.get("dil", 1), kw.get("stride", 1), bool(kw.get("convt_u"))))
    return r
ops.conv_gemm = wrapped
voc.use_graphs = False
voc.forward(c, f0, harm); recs.clear()
voc.forward(c, f0, harm); torch.cuda.synchronize()
agg = {}
for a, b, m, n, cin, taps, dil, st, ct in recs:
    k = (m, n, cin, taps, st, ct); t = agg.setdefault(k, [0, 0.0]); t[0] += 1; t[1] += a.elapsed_time(b)
tot = 0
for k, (cnt, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    fl = 2.0 * k[0] * k[1] * k[2] * k[3] * cnt
    tot += ms
    print(f"m={k[0]:7d} n={k[1]:5d} cin={k[2]:5d} taps={k[3]:3d} stride={k[4]:2d} convT={int(k[5])}  x{cnt:3d}  {ms:7.3f} ms  {fl / ms / 1e9:7.1f} TFLOP/s")
print("sum of conv kernels", round(tot, 3), "ms")
