"""BASELINE cfg 5 share through the PRODUCT entry (serving.BatchConverter), standalone: what bench.py's other_configs times."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from knn_svc_amd import config as C, ops, serving, synthetic as S
from knn_svc_amd.matcher import KNeighborsVC
from knn_svc_amd.vocoder import Vocoder
from knn_svc_amd.wavlm import WavLMEncoder
dev = torch.device("cuda", 0)
enc = WavLMEncoder(S.seeded_state(S.wavlm_param_spec(C.WAVLM_LARGE, 6), seed=1), C.WAVLM_LARGE, dev, 6)
voc = Vocoder(S.seeded_state(S.generator_param_spec(C.HIFIGAN_V1, "mix"), seed=2), C.HIFIGAN_V1, "mix", dev)
vc = KNeighborsVC(enc, voc, C.HIFIGAN_V1, dev)
n = 30 * C.SAMPLE_RATE
S_ = int(sys.argv[1]) if len(sys.argv) > 1 else 32
with torch.inference_mode():
    tv = serving.TargetVoice.from_clips(vc, [S.synth_clip(n, seed=5000 + i) for i in range(120)])
    reqs = [(torch.from_numpy(w).to(dev), torch.from_numpy((f * 1.3).astype(np.float32)).to(dev)) for w, f in (S.synth_clip(n, seed=7000 + i) for i in range(S_))]
    conv = serving.BatchConverter(vc, tv, "mix", "post_opt_0.2")
    conv.convert(reqs); conv.convert(reqs)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        conv.convert(reqs)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
print(json.dumps({"xRT": round(S_ * 30 / dt, 1), "ms_per_source": round(dt / S_ * 1e3, 2), "routes": ops.KNN_ROUTE_COUNTS}))
