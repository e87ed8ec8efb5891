"""Time the gated rel-pos attention kernel at the north-star shape (21 x 1500 frames, 16 heads)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from knn_svc_amd import ops
B, T, H = 21, 1500, 16
torch.manual_seed(0)
qkv = torch.randn(B, T, 3 * H * 64, device="cuda")
gate = torch.rand(B, T, H, device="cuda") * 2
table = torch.randn(H, 2 * T - 1, device="cuda")
for _ in range(2): y = ops.wavlm_attention(qkv, gate, table, B, T, H)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): y = ops.wavlm_attention(qkv, gate, table, B, T, H)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
fl = 4.0 * B * H * T * T * 64
print(f"attention {B}x{T}x{H}: {ms:.3f} ms, {fl / ms / 1e9:.1f} TF/s fp32-equivalent")
# reference check on one batch row in fp64
q, k, v = qkv[0].double().view(T, 3, H, 64).unbind(1)
s = torch.einsum("thd,shd->hts", q * 0.125, k)
idx = torch.arange(T, device="cuda")
rel = (idx[None, :] - idx[:, None] + T - 1)
s = s + gate[0].double().t()[:, :, None] * table.double()[:, rel]
o = torch.einsum("hts,shd->thd", s.softmax(-1), v).reshape(T, H * 64)
yy = y.view(B, T, H * 64); print("max abs err vs fp64:", (yy[0].double() - o).abs().max().item(), "rms", ((yy[0].double() - o).pow(2).mean().sqrt() / o.pow(2).mean().sqrt()).item())
