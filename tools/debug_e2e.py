"""Stage-by-stage comparison of the HIP path with the oracle on the tiny e2e case (debug aid)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from knn_svc_amd import config as C, synthetic as S
from knn_svc_amd.matching import match_features, side_features
from knn_svc_amd.wavlm import WavLMEncoder
from oracle import pipeline_ref, knn_ref
dev = "cuda"
cfg = C.WAVLM_TINY
sdw = S.seeded_state(S.wavlm_param_spec(cfg), seed=11)
src, sf0 = S.synth_clip(3 * 16000 + 77, 81); sf0 = sf0 * 1.25
pool = [S.synth_clip(4 * 16000 + 5 * i, 82 + i) for i in range(2)]
enc = WavLMEncoder(sdw, cfg, dev, n_layers=2)
def feats(w, f0):
    wg = torch.from_numpy(w).to(dev); ft = enc.full_features(wg); f0g, harm, spec = side_features(wg, f0, ft.shape[0]); return ft, f0g, harm, spec
q = feats(src, sf0); parts = [feats(w, f) for w, f in pool]
P = torch.cat([p[0] for p in parts]).contiguous(); Pf0 = torch.cat([p[1] for p in parts]).contiguous(); Ph = torch.cat([p[2] for p in parts]).contiguous()
oq = pipeline_ref.utterance_features(sdw, cfg, torch.from_numpy(src), torch.from_numpy(sf0), 2)
op = pipeline_ref.build_pool(sdw, cfg, [torch.from_numpy(w) for w, _ in pool], [torch.from_numpy(f) for _, f in pool], None, 2)
print("feat diff q", float((q[0].cpu() - oq["feats"]).abs().max()), "pool", float((P.cpu() - op["feats"]).abs().max()), "harm", float((Ph.cpu() - op["harm"]).abs().max()), "spec", float((parts[0][3].cpu()-op["spec"][:len(parts[0][3])]).abs().max()))
for po in ("no_post_opt", "post_opt_0.2"):
    of, hw, s0, dbg = match_features(q[0], q[1], P, Pf0, Ph, "mix", po, return_debug=True)
    rof, rhw, rs0, rdbg = pipeline_ref.match(oq, op, "mix", po, return_debug=True)
    # same-features comparison: run the oracle on the GPU features too
    gq = dict(feats=q[0].cpu(), f0=q[1].cpu()); gp = dict(feats=P.cpu(), f0=Pf0.cpu(), harm=Ph.cpu())
    sof, shw, ss0, sdbg = pipeline_ref.match(gq, gp, "mix", po, return_debug=True)
    for name, other in (("oracle(cpu feats)", rdbg), ("oracle(gpu feats)", sdbg)):
        print(po, name)
        for k in ("nn32", "idx_wavlm", "idx_harm"):
            a, b = dbg[k].cpu(), other[k]
            print(f"   {k}: rows equal {float((a == b).all(1).float().mean()):.4f}, top4 {float((a[:, :4] == b[:, :4]).all(1).float().mean()):.4f}")
        if dbg["w_wavlm"] is not None:
            print("   w_wavlm maxdiff", float((dbg["w_wavlm"].cpu() - other["w_wavlm"]).abs().max()), "w_harm", float((dbg["w_harm"].cpu() - other["w_harm"]).abs().max()))
    print("   out feats diff vs cpu-feat oracle", float((of.cpu() - rof).abs().max()), "vs gpu-feat oracle", float((of.cpu() - sof).abs().max()), "harm", float((hw.cpu() - rhw).abs().max()), float((hw.cpu() - shw).abs().max()))
    d64 = knn_ref.cosine_dist_f64(q[0].cpu(), P.cpu())
    print("   kNN agreement (same feats):", knn_ref.topk_agreement(sdbg["nn32"], dbg["nn32"].cpu(), d64))
    bad = (dbg["nn32"].cpu()[:, :4] != sdbg["nn32"][:, :4]).any(1).nonzero().reshape(-1)[:5]
    for r in bad.tolist():
        print("   row", r, "gpu", dbg["nn32"][r, :6].tolist(), "ref", sdbg["nn32"][r, :6].tolist(), "d64", d64[r, dbg["nn32"][r, :6].cpu()].tolist())

# ---- vocoder stage on the matched features ------------------------------------------------------
from knn_svc_amd.vocoder import Vocoder
from knn_svc_amd import ops
from oracle import vocoder_ref, synth_ref
h = C.HIFIGAN_TINY
sdg = S.seeded_state(S.generator_param_spec(h, "mix"), 63)
voc = Vocoder(sdg, h, "mix", dev)
of, hw, s0 = match_features(q[0], q[1], P, Pf0, Ph, "mix", "no_post_opt")
y = voc.forward(of, s0, hw).cpu()
ref = vocoder_ref.synthesizer(sdg, h, "mix", of.cpu()[None], s0.cpu()[None, :, None], hw.cpu()[None])[0, 0]
print("vocoder on identical inputs: rms", float((y - ref).pow(2).mean().sqrt()), "max", float((y - ref).abs().max()))
N = of.shape[0]
cond = torch.empty(N * 320, 8, device=dev)
exc = ops.additive_synth(s0, hw, voc.prenet_w, voc.prenet_b, cond[:, 4:], 8, want_exc=True).cpu()
rexc = synth_ref.additive_synth(s0.cpu()[None, :, None], hw.cpu()[None])[0, :, 0]
d = (exc - rexc).abs()
print("excitation: max diff", float(d.max()), "at", int(d.argmax()), "rms", float(d.pow(2).mean().sqrt()), "| harm max", float(hw.max()), "f0 range", float(s0.min()), float(s0.max()))
worst = int(d.argmax()); fr = worst // 320
print("  frame", fr, "f0 around", s0[max(0, fr - 2):fr + 3].tolist())
