"""Micro-benchmark of one channel-last conv1d through knnsvc_conv_gemm (the generator's ResBlock shapes):
    python tools/conv_bench.py M CIN N TAPS DIL [resid] [iters]
prints ms, TFLOP/s and GB/s of algorithmic traffic (x + out (+ resid)); which kernel ran comes from the library."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from knn_svc_amd import ops
M, CIN, N, TAPS, DIL = (int(v) for v in sys.argv[1:6])
resid = len(sys.argv) > 6 and sys.argv[6] == "1"
iters = int(sys.argv[7]) if len(sys.argv) > 7 else 20
torch.manual_seed(0)
x = torch.randn(M, CIN, device="cuda"); w = torch.randn(N, TAPS * CIN, device="cuda") / (TAPS * CIN) ** 0.5
b = torch.randn(N, device="cuda"); out = torch.empty(M, N, device="cuda"); r = torch.randn(M, N, device="cuda") if resid else None
ops.attach_split(w)
kw = dict(m=M, n=N, cin=CIN, taps=TAPS, dil=DIL, pad=(TAPS * DIL - DIL) // 2, t_in=M, bias=b, a_slope=0.1)
if resid: kw.update(resid=r, ldr=N)
for _ in range(3): ops.conv_gemm(x, w, out, **kw)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record()
for _ in range(iters): ops.conv_gemm(x, w, out, **kw)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / iters
by = 4.0 * (M * CIN + M * N * (2 if resid else 1))
print(f"{ops.last_conv_kernel():6s} m={M} cin={CIN} n={N} taps={TAPS} dil={DIL} resid={int(resid)}: {ms * 1e3:8.1f} us  "
      f"{2.0 * M * N * CIN * TAPS / ms / 1e9:7.1f} TFLOP/s  {by / ms / 1e6:7.1f} GB/s  checksum {float(out.double().sum()):.9e}")
