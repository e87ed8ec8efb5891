"""Same GEMM launch on random and on all-zero operands: a large gap means the loop is held back by the clock the chip can
sustain under this load (data-dependent power), not by its instruction schedule.  python tools/gemm_zero.py [M N K]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from knn_svc_amd import ops
M, N, K = (int(v) for v in (sys.argv[1:4] if len(sys.argv) >= 4 else (31500, 4096, 1024)))
for name, fill in (("random", None), ("zeros", 0.0), ("random", None)):
    x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") / K ** 0.5
    if fill is not None: x.zero_(); w.zero_()
    b = torch.zeros(N, device="cuda"); out = torch.empty(M, N, device="cuda")
    ops.attach_split(w); xin = ops.split_pack(x)
    for _ in range(50): ops.linear(xin, w, b, out=out, x_split=True)       # ~50 ms of load before timing
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(100): ops.linear(xin, w, b, out=out, x_split=True)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 100
    print(f"{name:7s} M={M} N={N} K={K}: {ms:.3f} ms  {2.0 * M * N * K / ms / 1e9:.1f} TFLOP/s fp32-equivalent")
