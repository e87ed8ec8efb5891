"""Attention micro-benchmark (WavLM-Large layer shape: 21 chunks x 1500 frames, 16 heads of 64): fp32 K/V vs K/V
pre-split to the f16x2 layout by the QKV projection (kv_split)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from knn_svc_amd import ops
B, T, H = (int(v) for v in (sys.argv[1:4] if len(sys.argv) >= 4 else (21, 1500, 16)))
E = H * 64
torch.manual_seed(0)
qkv = torch.randn(B * T, 3 * E, device="cuda") * 0.5
gate = torch.rand(B * T, H, device="cuda")
table = torch.randn(H, 2 * T - 1, device="cuda") * 0.1
qkv2 = qkv.clone()
qkv2[:, E:] = ops.split_pack(qkv[:, E:].contiguous())
flop = 4.0 * T * T * 64 * H * B
for name, x, kv in (("fp32 K/V", qkv, False), ("pre-split K/V", qkv2, True)):
    for _ in range(2): y = ops.wavlm_attention(x, gate, table, B, T, H, kv_split=kv)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(10): y = ops.wavlm_attention(x, gate, table, B, T, H, kv_split=kv)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"{name:14s} B={B} T={T} H={H}: {ms:.3f} ms  {flop / ms / 1e9:.1f} TFLOP/s fp32-equivalent")
    print(f"   checksum {float(y.double().sum()):.9e} {float(y.double().abs().sum()):.9e}")
    if kv: print("max |diff| between the two:", float((y - y0).abs().max()))
    y0 = y
