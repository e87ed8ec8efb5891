"""Shared by the CPU and GPU prematch tests: rebuild the toy dataset of tests/gen_golden.py gen_prematch
(same seeds, PCM_16 files + f0 caches) and read the g12 fixture back per utterance."""
import numpy as np

from knn_svc_amd import audio_io, synthetic as S

KEYS = ("nearest_nbrs", "nearest_nbrs_f0_priority", "amp_ratio", "harmonics_best_weight_para")


def layout(g):
    """{'spkA': [(n_samples, seed), ...], ...} as recorded in the fixture."""
    out = {}
    for ent in g["layout"]:
        name, rest = str(ent).split(":")
        out[name] = [tuple(int(v) for v in p.split("/")) for p in rest.split(",")]
    return out


def write_dataset(root, g):
    for name, utts in layout(g).items():
        (root / name).mkdir(parents=True)
        for i, (n, seed) in enumerate(utts):
            w, f = S.synth_clip(n, seed=seed)
            audio_io.write_wav_pcm16(str(root / name / f"u{i}.wav"), w, 16000)
            np.save(root / name / f"u{i}_f0.npy", f)
    return layout(g)
