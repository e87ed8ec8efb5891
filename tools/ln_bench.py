import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from knn_svc_amd import ops
rows, dim = int(sys.argv[1]), int(sys.argv[2])
x = torch.randn(rows, dim, device="cuda"); g = torch.randn(dim, device="cuda"); b = torch.randn(dim, device="cuda")
out = torch.empty_like(x)
for gelu in (False, True):
    for _ in range(2): ops.layernorm(x, g, b, gelu=gelu, out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): ops.layernorm(x, g, b, gelu=gelu, out=out)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"rows={rows} dim={dim} gelu={gelu}: {ms:.3f} ms  {2 * rows * dim * 4 / ms / 1e9:.2f} TB/s")
