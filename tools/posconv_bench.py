"""WavLM's positional convolution as the bench's encoder launches it (k = 128, 16 groups of 64 channels, 21 chunks of 1500 frames, GELU +
residual epilogue): us per launch alone on the GPU and a checksum."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from knn_svc_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 21
T, E, G, K = 1500, 1024, 16, 128
cg = E // G
torch.manual_seed(0)
x = torch.randn(B * T, E, device="cuda") * 0.3
w = torch.randn(E, cg, K) / (cg * K) ** 0.5
pw = ops.attach_split(ops.pack_grouped_conv_weight(w, G).cuda())
b = torch.randn(E, device="cuda") * 0.1
out = torch.empty_like(x)
def run():
    ops.conv_gemm(x, pw, out, m=T, n=cg, cin=cg, taps=K, pad=K // 2, t_in=T, ldx=E, ldo=E, bias=b, act=ops.ACT_GELU, resid=x, ldr=E,
                  batches=B, groups=G, x_bstride=T * E, o_bstride=T * E, r_bstride=T * E, x_gstride=cg, o_gstride=cg, r_gstride=cg,
                  bias_gstride=cg, w_gstride=cg * cg * K)
for _ in range(3): run()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record()
for _ in range(10): run()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 10 * 1e3
print(f"B={B}: {us:.1f} us per launch ({ops.last_conv_kernel()}), {2.0 * B * T * E * cg * K / us / 1e6:.1f} TFLOP/s fp32-equivalent, checksum {float(out.double().sum()):.9e}")
