"""Run-to-run determinism of the conversion stages at cfg-5-share size: the same batch twice, stage by stage."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from knn_svc_amd import config as C, matching as M, ops, serving, synthetic as S
from knn_svc_amd.matcher import KNeighborsVC
from knn_svc_amd.vocoder import Vocoder, serial_resblocks
from knn_svc_amd.wavlm import WavLMEncoder
dev = torch.device("cuda", 0)
enc = WavLMEncoder(S.seeded_state(S.wavlm_param_spec(C.WAVLM_LARGE, 6), seed=1), C.WAVLM_LARGE, dev, 6)
voc = Vocoder(S.seeded_state(S.generator_param_spec(C.HIFIGAN_V1, "mix"), seed=2), C.HIFIGAN_V1, "mix", dev)
vc = KNeighborsVC(enc, voc, C.HIFIGAN_V1, dev)
n = 30 * C.SAMPLE_RATE
NP, NS = int(os.environ.get("NP", "40")), int(os.environ.get("NS", "8"))
with torch.inference_mode():
    tv = serving.TargetVoice.from_clips(vc, [S.synth_clip(n, seed=5000 + i) for i in range(NP)])
    srcs = [S.synth_clip(n, seed=7000 + i) for i in range(NS)]
    wavs = [torch.from_numpy(w).to(dev) for w, _ in srcs]
    f0s = [torch.from_numpy((f * 1.3).astype(np.float32)).to(dev)[:1500] for _, f in srcs]

    def stages(batch):
        out = {}
        feats = enc.encode_many(wavs if batch else wavs[:1], max_batch=32, pow2_batches=True)
        out["feats"] = feats[0].clone()
        of, hw, s0, dbg = M.match_features(feats[0], f0s[0], tv.feats, tv.f0, tv.harm, "mix", "post_opt_0.2", return_debug=True, pool_prep=tv.prep)
        for k in ("nn32", "idx_wavlm", "w_wavlm", "idx_harm", "w_harm"):
            out[k] = dbg[k].clone()
        out["of"], out["hw"], out["s0"] = of.clone(), hw.clone(), s0.clone()
        for mode in ("par", "ser"):
            for rep in range(3):
                if mode == "ser":
                    with serial_resblocks():
                        y = voc.forward(of, s0, hw)
                else:
                    y = voc.forward(of, s0, hw)
                out[f"y_{mode}{rep}"] = y.clone()
        torch.cuda.synchronize()
        return out
    a = stages(True); b = stages(True); c = stages(False)
    for k in a:
        ab = torch.equal(a[k], b[k]); ac = torch.equal(a[k], c[k])
        d = float((a[k].double() - c[k].double()).abs().max())
        print(f"{k:10s} run-to-run equal: {ab}   batch-vs-alone equal: {ac} (max diff {d:.2e})")
    print("vocoder modes/reps equal to y_par0:", {k: torch.equal(a["y_par0"], a[k]) for k in a if k.startswith("y_")})
