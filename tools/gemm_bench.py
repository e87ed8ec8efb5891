"""Micro-benchmark of knnsvc_conv_gemm on one linear shape (M, N, K): prints TFLOP/s (HIP events)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from knn_svc_amd import ops
M, N, K = (int(v) for v in (sys.argv[1:4] if len(sys.argv) >= 4 else (31500, 4096, 1024)))
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 20
x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") / K ** 0.5; b = torch.randn(N, device="cuda")
out = torch.empty(M, N, device="cuda")
if os.environ.get("SPLIT", "1") == "1":
    ops.attach_split(w)
    print("split:", "f16x2" if hasattr(w, "_w2") else "bf16x3" if hasattr(w, "_w3") else "none")
A2 = os.environ.get("A2", "0") == "1"          # feed pre-split activations (x_split): the DMA-fed kernel
xin = ops.split_pack(x) if A2 else x
# epilogue variants: ACT=gelu, RESID=1 (residual operand), OSPLIT=1 (split-layout output)
kw = dict(act=ops.ACT_GELU if os.environ.get("ACT") == "gelu" else ops.ACT_NONE,
          resid=torch.randn(M, N, device="cuda") if os.environ.get("RESID") == "1" else None,
          out_split=os.environ.get("OSPLIT") == "1")
_lin = ops.linear
ops_linear = lambda xi, ww, bb, out, x_split: _lin(xi, ww, bb, out=out, x_split=x_split, **kw)
for _ in range(int(os.environ.get("WARM", "3"))):
    ops_linear(xin, w, b, out=out, x_split=A2)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record()
for _ in range(iters):
    ops_linear(xin, w, b, out=out, x_split=A2)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / iters
print(f"M={M} N={N} K={K}: {ms:.3f} ms  {2.0 * M * N * K / ms / 1e9:.1f} TFLOP/s")

print("kernel:", ops.last_conv_kernel())
if kw["act"] or kw["resid"] is not None or kw["out_split"]:
    sys.exit(0)
ref = (x[:256].double().cpu() @ w.double().cpu().T + b.double().cpu())
err = (out[:256].double().cpu() - ref).abs().max().item()
print(f"max |err| vs fp64 on 256 rows: {err:.3e}  (|ref| max {ref.abs().max().item():.2f})")
