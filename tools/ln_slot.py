import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from knn_svc_amd import ops
torch.manual_seed(0)
T, dim = 4001, 512
one = torch.randn(T, dim) * 2 + 0.3
g = torch.randn(dim); b = torch.randn(dim)
ref = torch.nn.functional.layer_norm(one.double(), (dim,), g.double(), b.double(), 1e-5)
outs = []
for shift in range(4):
    x = torch.cat([torch.randn(shift, dim), one]).contiguous().cuda()
    y = ops.layernorm(x, g.cuda(), b.cuda(), gelu=False)[shift:].cpu()
    outs.append(y)
    e = (y.double() - ref).abs()
    print(f"shift {shift}: max err vs fp64 {float(e.max()):.3e}, mean err {float(e.mean()):.3e}")
for s in range(1, 4):
    d = (outs[s] != outs[0]).any(1)
    print(f"shift {s} vs 0: rows differing {int(d.sum())} / {T}")
x1 = torch.cat([one, one]).contiguous().cuda()
y1 = ops.layernorm(x1, g.cuda(), b.cuda(), gelu=False).cpu()
print("same data, slot r vs slot r+1:", int((y1[:T] != y1[T:]).any(1).sum()))
