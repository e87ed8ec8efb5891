"""The STFT's DFT product as the bench runs it (21 signals of 30 s: [1501, 416] x [416, 400] per signal, rows = frames at hop 320 of the
padded signal): us per launch, alone on the GPU."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from knn_svc_amd import features as F_, ops
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 21
dev = "cuda"
n_fft, hop = 400, 320
basis = F_._dft_basis(n_fft, dev)
kp = basis.shape[1]
L = 480000
Tmax = 1 + L // hop
stride = ((Tmax - 1) * hop + kp + 3) // 4 * 4
xp = torch.randn(nb * stride, device=dev) * 0.1
reim = torch.empty(nb * Tmax, n_fft, device=dev)
def run():
    ops.conv_gemm(xp, basis, reim, m=Tmax, n=n_fft, cin=kp, taps=1, stride=1, pad=0, t_in=Tmax, ldx=hop, a_scale=F_.X_SCALE,
                  batches=nb, x_bstride=stride, o_bstride=Tmax * n_fft)
for _ in range(3): run()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record()
for _ in range(20): run()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 20 * 1e3
print(f"nb={nb}: {us:.1f} us per launch, kernel {ops.last_conv_kernel() if hasattr(ops, 'last_conv_kernel') else '?'} / {ops.last_conv_epilogue()}, "
      f"{2 * nb * Tmax * n_fft * kp / us / 1e6:.1f} TFLOP/s fp32-equivalent")
