"""In-kernel phase timing of the 256x256 kernel on the conv stack's first layer as the bench's encoder launches it (k = 3, stride 2,
512 -> 512 channels, 21 chunks of 96 015 input frames; split input, bias epilogue) and, beside it, the same launch timed with HIP events;
needs the -DKN_QUAD_PROF build (tools/quad_prof.sh):   KNNSVC_LIB=knn_svc_amd/libknnsvc_prof.so python tools/quad_prof_conv.py [B]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from knn_svc_amd import ops, _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 21
T, C, k, st = int(os.environ.get("T_IN", "96015")), 512, 3, 2
t_out = (T - k) // st + 1
x = torch.randn(B * T, C, device="cuda") * 0.5
w = ops.attach_split(ops.pack_conv_weight(torch.randn(C, C, k) / (C * k) ** 0.5).cuda())
b = torch.randn(C, device="cuda")
xs = ops.split_pack(x)
y = torch.empty(B * t_out, C, device="cuda")
def run():
    ops.conv_gemm(xs, w, y, m=t_out, n=C, cin=C, taps=k, stride=st, t_in=T, batches=B, x_bstride=T * C, o_bstride=t_out * C, x_split=True, bias=b)
for _ in range(3): run()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record()
for _ in range(5): run()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
print(f"{ops.last_conv_kernel()} m={t_out} n={C} K={C * k} z={B}: {ms:.3f} ms  {2.0 * B * t_out * C * C * k / ms / 1e9:.1f} TFLOP/s fp32-equivalent")
lib = _lib.load()
if hasattr(lib, "knnsvc_debug_quad_prof"):
    nb = 8192
    buf = (ctypes.c_longlong * (nb * 4))()
    if lib.knnsvc_debug_quad_prof(buf, nb) == 0:
        t = np.frombuffer(buf, dtype=np.int64).reshape(nb, 4).astype(np.float64) * 0.01
        t = t[t[:, 3] > t[:, 0]]
        d = np.diff(t, axis=1)
        span = t[:, 3].max() - t[:, 0].min()
        print(f"  span {span:.1f} us, sum of tile times / (256 CUs x span) = {(t[:, 3] - t[:, 0]).sum() / (256 * span):.3f}; main loop by start-time quartile: "
              + ", ".join(f"{d[np.argsort(t[:, 0])][q * len(t) // 4:(q + 1) * len(t) // 4, 1].mean():.1f}" for q in range(4)))
        print(f"  first {len(t)} block ids of the last batch slices recorded (blockIdx.x < 8192): prologue {d[:, 0].mean():.2f}  main loop {d[:, 1].mean():.2f} "
              f"(p10 {np.percentile(d[:, 1], 10):.2f}, p90 {np.percentile(d[:, 1], 90):.2f})  epilogue {d[:, 2].mean():.2f} us")
