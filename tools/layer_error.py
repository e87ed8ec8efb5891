"""Where does the encoder's distance from exact arithmetic come from?  Per exit layer: relative rms of the GPU output and of
the CPU fp32 oracle against the oracle evaluated in fp64 (WavLM-Large, seeded weights, one chunk of `secs` seconds)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from knn_svc_amd import config as C, synthetic as S
from knn_svc_amd.wavlm import WavLMEncoder
from oracle import wavlm_ref
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 6.0
nl = int(sys.argv[2]) if len(sys.argv) > 2 else 6
cfg = C.WAVLM_LARGE
sd = S.seeded_state(S.wavlm_param_spec(cfg, nl), seed=1)
if os.environ.get("STRESS") == "1":      # the outlier-weight state of tests/test_gpu_range.py (needs nl >= 3)
    for k, f in {"encoder.layers.1.final_layer_norm.weight": 60.0, "encoder.layers.1.fc1.weight": 40.0, "encoder.layers.1.fc2.weight": 1.0 / 2400.0,
                 "encoder.layers.0.self_attn.k_proj.weight": 200.0, "encoder.layers.0.self_attn.q_proj.weight": 1.0 / 200.0,
                 "encoder.layers.0.self_attn.q_proj.bias": 1.0 / 200.0,
                 "encoder.layers.2.self_attn_layer_norm.weight": 1500.0, "encoder.layers.2.self_attn.v_proj.weight": 1.0 / 1500.0,
                 "encoder.layers.2.self_attn.q_proj.weight": 1.0 / 1500.0, "encoder.layers.2.self_attn.k_proj.weight": 1.0 / 1500.0}.items():
        sd[k] = sd[k] * f
w, _ = S.synth_clip(int(secs * 16000), 31)
x = torch.from_numpy(np.pad(w, (0, 320)))[None]
r32 = wavlm_ref.extract_layer(sd, cfg, x, nl, all_layers=True)
r64 = wavlm_ref.extract_layer({k: v.double() for k, v in sd.items()}, cfg, x.double(), nl, all_layers=True)
rel = lambda a, b: float((a.double() - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt())
for l in range(nl + 1):
    out = WavLMEncoder(sd, cfg, "cuda", n_layers=l).encode_batch(x.cuda()).cpu()
    print(f"exit layer {l}: GPU vs fp64 {rel(out, r64[l]):.3e}   CPU fp32 vs fp64 {rel(r32[l], r64[l]):.3e}   (rms {float(r64[l].pow(2).mean().sqrt()):.3f})")

# front end in pieces: conv stack output (before the LayerNorm), post_extract_proj output, after the positional conv
import torch.nn.functional as F
enc = WavLMEncoder(sd, cfg, "cuda", n_layers=0)
enc._tap = {}
enc.use_graphs = False
out0 = enc.encode_batch(x.cuda()).cpu()
def front(sdx, xx):
    f = wavlm_ref.feature_extractor(sdx, cfg, xx)
    t = f.transpose(1, 2)
    t2 = F.layer_norm(t, (t.shape[-1],), sdx["layer_norm.weight"], sdx["layer_norm.bias"], 1e-5)
    return t[0], F.linear(t2, sdx["post_extract_proj.weight"], sdx["post_extract_proj.bias"])[0]
sd64 = {k: v.double() for k, v in sd.items()}
c32, p32 = front(sd, x); c64, p64 = front(sd64, x.double())
print(f"conv stack out: GPU {rel(enc._tap['conv'].cpu(), c64):.3e}  CPU fp32 {rel(c32, c64):.3e}")
print(f"proj out:       GPU {rel(enc._tap['proj'].cpu(), p64):.3e}  CPU fp32 {rel(p32, p64):.3e}")
# per conv layer, each fed with the fp64 reference input of that layer (isolates one layer's own error)
from knn_svc_amd import ops
xin = x.double()[:, None, :]
for i, (dim, k, st) in enumerate(wavlm_ref.conv_layers_of(cfg)):
    pz = f"feature_extractor.conv_layers.{i}."
    y64 = F.conv1d(xin, sd64[pz + "0.weight"], None, stride=st)
    y32 = F.conv1d(xin.float(), sd[pz + "0.weight"], None, stride=st)
    if i > 0:
        xg = xin.float()[0].T.contiguous().cuda()                      # [T_in, C] channel-last
        t_out = y64.shape[-1]
        yg = torch.empty(t_out, dim, device="cuda")
        c = enc.conv[i]
        ops.conv_gemm(xg, c["w"], yg, m=t_out, n=dim, cin=xg.shape[1], taps=k, stride=st, t_in=xg.shape[0])
        print(f"conv layer {i} alone (K = {k * xg.shape[1]}): GPU {rel(yg.cpu().T[None], y64):.3e}  CPU fp32 {rel(y32, y64):.3e}")
    z = F.layer_norm(y64.transpose(1, 2), (dim,), sd64[pz + "2.1.weight"], sd64[pz + "2.1.bias"], 1e-5)
    xin = F.gelu(z.transpose(1, 2))
