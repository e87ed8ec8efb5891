"""Generator at 1500 frames: ms per hipGraph replay (HIP events), the figure the pipelined benches are bound by."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from knn_svc_amd import config as C, synthetic as S
from knn_svc_amd.vocoder import Vocoder
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
voc = Vocoder(S.seeded_state(S.generator_param_spec(C.HIFIGAN_V1, "mix"), 2), C.HIFIGAN_V1, "mix", "cuda")
g = torch.Generator().manual_seed(0)
c = torch.randn(N, 1024, generator=g).cuda(); harm = (torch.rand(N, 49, generator=g) * 0.02).cuda()
_, f0 = S.synth_clip(N * 320, 5); f0 = torch.from_numpy(f0[:N].copy()).cuda()
for _ in range(4): voc.forward(c, f0, harm)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record()
for _ in range(20): voc.forward(c, f0, harm)
e1.record(); torch.cuda.synchronize()
import zlib
y = voc.forward(c, f0, harm); torch.cuda.synchronize()
crc = zlib.crc32(y.detach().float().cpu().numpy().tobytes())
print(f"N={N}: crc {crc:08x} ", end="")
print(f" {e0.elapsed_time(e1) / 20:.3f} ms per replay (graphs: {list(voc._graphs)})")
