"""Model loaders with the reference's signatures (ddsp_hubconf.py:17-128).

``knn_vc(...)`` returns a :class:`KNeighborsVC` whose WavLM encoder and generator
are packed for libknnsvc_hip.so.  Checkpoint formats are the reference's:
generator ``{'generator': state_dict}`` found by glob ``*{ckpt_type}*`` in
``local_ckpt_dir`` (hifigan/utils.py:55-60), WavLM ``{'cfg': dict, 'model':
state_dict}`` from the torch.hub cache / URL (ddsp_hubconf.py:113-119).

There is no network on the build or benchmark boxes.  ``weights='seeded'`` (or the
environment variable KNNSVC_SEEDED_WEIGHTS=1) swaps both checkpoints for seeded
random weights of the exact architecture so that every code path still runs; the
default ``weights='checkpoint'`` fails loudly when a file is missing.
"""
from __future__ import annotations

import glob
import logging
import os

import torch

from . import config as C, synthetic as S
from .matcher import KNeighborsVC
from .vocoder import Vocoder
from .wavlm import WavLMEncoder

dependencies = ["torch", "numpy"]
WAVLM_URL = "https://github.com/bshall/knn-vc/releases/download/v0.1/WavLM-Large.pt"
DEFAULT_CKPT_DIR = "/home/ken/Downloads/knn_vc_data/ckpt_saved"      # the reference's default (ddsp_hubconf.py:17)


def _seeded(weights) -> bool:
    return weights == "seeded" or (weights == "auto" and os.environ.get("KNNSVC_SEEDED_WEIGHTS") == "1")


def generator_kind(ckpt_type: str) -> str:
    """Class routing by substring (ddsp_hubconf.py:45-60)."""
    if "wavlm_only" in ckpt_type or "no_harm_no_amp" in ckpt_type:
        if "wavlm_only_original" in ckpt_type:
            raise NotImplementedError("wavlm_only_original needs hifigan/models.py, which the reference snapshot lacks")
        return "f0"
    return "mix"


def scan_checkpoint(cp_dir: str, prefix: str):
    found = glob.glob(os.path.join(cp_dir, "*" + prefix + "*"))
    return sorted(found)[-1] if found else None


def hifigan_wavlm(pretrained=True, progress=True, prematched=True, ckpt_type="mix", device="cuda",
                  local_ckpt_dir=DEFAULT_CKPT_DIR, weights="auto", h=None):
    h = dict(h or C.HIFIGAN_V1)
    h["hubert_dim"] = h.get("hubert_dim", 1024)
    kind = generator_kind(ckpt_type)
    if _seeded(weights):
        sd = S.seeded_state(S.generator_param_spec(h, kind), seed=2)
    else:
        if not os.path.isdir(local_ckpt_dir):
            raise SystemExit(f"Bad ckpt {local_ckpt_dir} location")
        cp = scan_checkpoint(local_ckpt_dir, ckpt_type)
        if cp is None:
            raise SystemExit(f"no checkpoint matching *{ckpt_type}* in {local_ckpt_dir}")
        sd = torch.load(cp, map_location="cpu")["generator"]
        print("Loaded ckpt from local", cp)
    gen = Vocoder(sd, h, kind, device)
    print(f"[HiFiGAN] Generator loaded with {sum(v.numel() for v in sd.values()):,d} parameters.")
    return gen, h


def wavlm_large(pretrained=True, progress=True, device="cuda", weights="auto", cfg=None, n_layers=C.MATCH_LAYER):
    if not torch.cuda.is_available():
        raise RuntimeError("knn_svc_amd needs a ROCm GPU: there is no CPU path (the reference's --device cpu "
                           "route is reproduced by the oracle for tests only)")
    if _seeded(weights):
        cfg = dict(cfg or C.WAVLM_LARGE)
        sd = S.seeded_state(S.wavlm_param_spec(cfg, n_layers), seed=1)
    else:
        local = os.environ.get("KNNSVC_WAVLM_PT")          # an explicit file instead of the torch.hub cache / URL
        ck = (torch.load(local, map_location="cpu") if local else
              torch.hub.load_state_dict_from_url(WAVLM_URL, map_location="cpu", progress=progress))
        cfg = dict(C.WAVLM_LARGE, **ck["cfg"])
        sd = ck["model"]
        print("Pretrained WavLM loaded")
    enc = WavLMEncoder(sd, cfg, device, n_layers)
    print(f"WavLM loaded: {n_layers} of {cfg['encoder_layers']} layers packed.")
    return enc


def knn_vc(pretrained=True, progress=True, prematched=True, ckpt_type="mix", device="cuda",
           local_ckpt_dir=DEFAULT_CKPT_DIR, weights="auto") -> KNeighborsVC:
    """Load kNN-SVC (WavLM encoder + conditioned HiFi-GAN) — ddsp_hubconf.py:17-25."""
    if str(device).startswith("cpu"):
        # ddsp_inference.py:39 offers --device cpu; this build has NO CPU path (the oracle under oracle/ is test infrastructure and
        # must never serve a conversion).  Fail once and clearly instead of remapping to 'cuda' and dying later elsewhere.
        raise RuntimeError("device='cpu' requested: knn_svc_amd only runs on a ROCm GPU (MI355X, gfx950) — there is no CPU path; "
                           "pass --device cuda")
    hifigan, h = hifigan_wavlm(pretrained, progress, prematched, ckpt_type, device, local_ckpt_dir, weights)
    wavlm = wavlm_large(pretrained, progress, device, weights)
    return KNeighborsVC(wavlm, hifigan, h, device)
