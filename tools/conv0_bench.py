import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from knn_svc_amd import ops
B, L = 21, 480320
x = torch.randn(B, L, device="cuda") * 0.1; w = torch.randn(512, 10, device="cuda"); g = torch.randn(512, device="cuda"); b = torch.randn(512, device="cuda")
for _ in range(2): y = ops.wavlm_conv0(x, w, g, b, 10, 5)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): y = ops.wavlm_conv0(x, w, g, b, 10, 5)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
print(f"conv0+LN+GELU 21 x 30 s: {ms:.3f} ms, write {y.numel() * 4 / ms / 1e9:.2f} TB/s")
