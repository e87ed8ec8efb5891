"""Is the PIPELINED batch path (serving.BatchConverter) run-to-run deterministic, and equal to the per-source path?"""
import os, sys
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
NP, NS = int(os.environ.get("NP", "120")), int(os.environ.get("NS", "32"))
with torch.inference_mode():
    tv = serving.TargetVoice.from_clips(vc, [S.synth_clip(n, seed=5000 + i) for i in range(NP)])
    reqs = [(torch.from_numpy(w).to(dev), torch.from_numpy((f * 1.3).astype(np.float32)).to(dev)) for w, f in (S.synth_clip(n, seed=7000 + i) for i in range(NS))]
    conv = serving.BatchConverter(vc, tv, "mix", "post_opt_0.2")
    runs = [[y.clone() for y in conv.convert(reqs)] for _ in range(4)]
    torch.cuda.synchronize()
    for r in range(1, 4):
        bad = [i for i in range(NS) if not torch.equal(runs[0][i], runs[r][i])]
        print(f"run {r} vs run 0: {len(bad)} of {NS} sources differ", [(i, float((runs[0][i] - runs[r][i]).abs().max())) for i in bad][:8])
    alone = [conv.convert([reqs[i]])[0] for i in range(NS)]
    bad = [i for i in range(NS) if not torch.equal(runs[3][i], alone[i])]
    print(f"batch (run 3) vs alone: {len(bad)} differ", [(i, float((runs[3][i] - alone[i]).abs().max())) for i in bad][:8])
    alone2 = [conv.convert([reqs[i]])[0] for i in range(NS)]
    bad = [i for i in range(NS) if not torch.equal(alone2[i], alone[i])]
    print(f"alone vs alone: {len(bad)} differ")
