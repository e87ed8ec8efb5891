import sys, os, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pathlib import Path
import numpy as np, torch
from knn_svc_amd import audio_io, config as C, synthetic as S
from knn_svc_amd.matching import match_at_inference_time, get_complete_spk_pool
from knn_svc_amd.wavlm import WavLMEncoder
from knn_svc_amd.vocoder import Vocoder
from oracle import pipeline_ref, vocoder_ref
dev = "cuda"
cfg, h = C.WAVLM_TINY, C.HIFIGAN_TINY
sdw = S.seeded_state(S.wavlm_param_spec(cfg), seed=11)
enc = WavLMEncoder(sdw, cfg, dev, n_layers=2)
tmp = Path(tempfile.mkdtemp()); (tmp / "a").mkdir(); (tmp / "b").mkdir()
src_wav, src_f0 = S.synth_clip(3 * 16000 + 77, seed=81)
audio_io.write_wav_pcm16(str(tmp / "a" / "src.wav"), src_wav, 16000); np.save(tmp / "a" / "src_f0.npy", src_f0 * 1.25)
for i in range(3):
    w, f = S.synth_clip(4 * 16000 + 5 * i, seed=82 + i)
    audio_io.write_wav_pcm16(str(tmp / "b" / f"u{i}.wav"), w, 16000); np.save(tmp / "b" / f"u{i}_f0.npy", f)
srcp = str(tmp / "a" / "src.wav")
wt = torch.zeros(4); wt[2] = 1
of, hw, _a, sf0 = match_at_inference_time(srcp, tmp / "b", enc, wt, wt, prioritize_f0=True, ckpt_type="mix", post_opt="no_post_opt", tgt_dataset_path=tmp, duration_limit=7)
rd = lambda p: torch.from_numpy(audio_io.read_wav(str(p))[0][0])
sw = rd(tmp / "a" / "src.wav"); pw = [rd(tmp / "b" / f"u{i}.wav") for i in range(3)]
pf = [torch.from_numpy(np.load(tmp / "b" / f"u{i}_f0.npy")) for i in range(3)]
sf = torch.from_numpy(np.load(tmp / "a" / "src_f0.npy"))
oq = pipeline_ref.utterance_features(sdw, cfg, sw, sf, 2)
op = pipeline_ref.build_pool(sdw, cfg, pw, pf, 7, 2)
rof, rhw, rs0, rdbg = pipeline_ref.match(oq, op, "mix", "no_post_opt", return_debug=True)
print("pool frames", op["feats"].shape)
print("out feats", float((of[srcp].cpu() - rof).abs().max()), "harm", float((hw[srcp].cpu() - rhw).abs().max()), "f0", float((sf0[srcp].cpu() - rs0).abs().max()))
mp, _, _, sp, fp, hp = get_complete_spk_pool(tmp / "b", enc, duration_limit=7)
P = torch.cat([mp[k] for k in mp]); H = torch.cat([hp[k] for k in hp]); F0 = torch.cat([fp[k] for k in fp])
print("pool feats diff", float((P.cpu() - op["feats"]).abs().max()), "harm pool diff", float((H.cpu() - op["harm"]).abs().max()), "f0", float((F0.cpu() - op["f0"]).abs().max()))
hd = (H.cpu() - op["harm"]).abs(); r = int(hd.max(1).values.argmax()); print(" worst harm row", r, "col", int(hd[r].argmax()), "f0", float(op["f0"][r]), "gpu", H[r, :3].tolist(), "ref", op["harm"][r, :3].tolist())
SP = torch.cat([sp[k] for k in sp]); print("spec diff", float((SP.cpu() - op["spec"]).abs().max()))
sdg = S.seeded_state(S.generator_param_spec(h, "mix"), 63)
y = Vocoder(sdg, h, "mix", dev).forward(of[srcp], sf0[srcp], hw[srcp]).cpu()
ref = vocoder_ref.synthesizer(sdg, h, "mix", rof[None], rs0[None, :, None], rhw[None])[0, 0]
print("wave rms", float((y - ref).pow(2).mean().sqrt()))
