"""In-kernel phase timing of conv_gemm2quad_kernel<Gemm2QuadS> (a -DKN_QUAD_PROF build, see tools/quad_prof.sh): per block
start / prologue done / main loop done / epilogue done (stores acknowledged), 10 ns ticks of the constant 100 MHz counter.
  KNNSVC_LIB=knn_svc_amd/libknnsvc_prof.so KNNSVC_QUADP=0 python tools/quad_prof.py M N K   (+ ACT=gelu OSPLIT=1 RESID=1)"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from knn_svc_amd import ops, _lib
M, N, K = (int(v) for v in sys.argv[1:4])
x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") / K ** 0.5; b = torch.randn(N, device="cuda")
ops.attach_split(w)
xin = ops.split_pack(x)
kw = dict(act=ops.ACT_GELU if os.environ.get("ACT") == "gelu" else ops.ACT_NONE,
          resid=torch.randn(M, N, device="cuda") if os.environ.get("RESID") == "1" else None,
          out_split=os.environ.get("OSPLIT") == "1")
out = torch.empty(M, N, device="cuda")
for _ in range(40):
    ops.linear(xin, w, b, out=out, x_split=True, **kw)
torch.cuda.synchronize()
assert ops.last_conv_kernel() == "Q256S", ops.last_conv_kernel()
nb = min(8192, (-(-M // 256) + 7) // 8 * 8 * -(-N // 256))       # (the row-padded order: encoder shapes)
buf = (ctypes.c_longlong * (nb * 4))()
lib = _lib.load()
assert lib.knnsvc_debug_quad_prof(buf, nb) == 0
t = np.frombuffer(buf, dtype=np.int64).reshape(nb, 4).astype(np.float64) * 0.01       # us
live = t[:, 3] > t[:, 0]
t = t[live]
t0 = t[:, 0].min()
print(f"M={M} N={N} K={K} {dict((k, bool(v) if not isinstance(v, torch.Tensor) else True) for k, v in kw.items())}: {len(t)} tiles, kernel span {t[:, 3].max() - t0:.1f} us")
d = np.diff(t, axis=1)
for name, col in (("prologue", 0), ("main loop", 1), ("epilogue", 2)):
    print(f"  {name:10s} mean {d[:, col].mean():7.2f} us   p10 {np.percentile(d[:, col], 10):7.2f}   p90 {np.percentile(d[:, col], 90):7.2f}")
order = np.argsort(t[:, 0])
starts = t[order, 0] - t0
print("  block start times (us), every 64th in start order:", np.round(starts[::64], 1).tolist())
ids = np.nonzero(live)[0]
ends = t[:, 3] - t0
print("  last tile end per XCD (block id mod 8), us:", [round(float(ends[ids % 8 == x].max()), 1) for x in range(8)],
      " tiles per XCD:", [int((ids % 8 == x).sum()) for x in range(8)])
busy = (t[:, 3] - t[:, 0])
print(f"  sum of tile times / (256 CUs x span) = {busy.sum() / (256 * (t[:, 3].max() - t0)):.3f}")
