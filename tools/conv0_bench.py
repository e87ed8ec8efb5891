"""conv0 + LayerNorm + GELU (first WavLM feature-extractor layer) at the bench's batch: ms per launch and write rate."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from knn_svc_amd import ops
B, L, C, k, st = 21, 480080, 512, 10, 5
g = torch.Generator().manual_seed(0)
x = torch.randn(B, L, generator=g).cuda(); w = (torch.randn(C, k, generator=g) * 0.3).cuda()
ga = torch.rand(C, generator=g).cuda() + 0.5; be = torch.randn(C, generator=g).cuda() * 0.1
for split in (False, True):
    for _ in range(3): y = ops.wavlm_conv0(x, w, ga, be, k, st, out_split=split)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(10): y = ops.wavlm_conv0(x, w, ga, be, k, st, out_split=split)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"split={split}: {ms:.3f} ms  {y.numel() * 4 / ms / 1e9:.2f} TB/s written  checksum {float(y.double().abs().sum()):.9e} bits {int(y.view(torch.int32).to(torch.int64).sum())}")
