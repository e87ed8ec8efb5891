"""Command line front end with the flags of the reference's ``ddsp_inference.py`` (:27-46).

  python ddsp_inference.py SRC TGT [--ckpt_dir D] [--ckpt_type mix] [--post_opt post_opt_0.2] ...

SRC/TGT are both audio files (single conversion, output next to SRC as
``<src>_to_<tgt>_knn_<ckpt_type>_<post_opt>.wav``) or both dataset roots (folders of
speaker folders; outputs under ``<tgt parent>/<src>_to_<tgt>_<ckpt_type>_post_opt_<post_opt>/``,
prefixed ``duration_limit_<n>_`` when --dur_limit is given).  --topk and
--tgt_loudness_db are accepted and ignored, as upstream; --dur_limit is compared in
seconds, as upstream (ddsp_prematch_dataset.py:408-411).
"""
from __future__ import annotations

import argparse
import os

FLAGS = [
    ("--ckpt_dir", dict(type=str, default="/home/ken/Downloads/knn_vc_data/ckpt_saved")),
    ("--ckpt_type", dict(type=str, default="mix")),
    ("--post_opt", dict(type=str, default="no_post_opt")),
    ("--required_subset_file", dict(type=str, default=None)),
    ("--topk", dict(type=int, default=4)),
    ("--device", dict(type=str, default="cuda")),
    ("--tgt_loudness_db", dict(type=float, default=-16)),
    ("--dur_limit", dict(type=int, default=None)),
]


def _bool(v: str) -> bool:
    v = v.lower()
    if v in ("yes", "true", "t", "1", "y"):
        return True
    if v in ("no", "false", "f", "0", "n"):
        return False
    raise argparse.ArgumentTypeError("boolean value expected")


def build_parser() -> argparse.ArgumentParser:
    ap = argparse.ArgumentParser(description="kNN-SVC inference (MI355X build): file or folder mode")
    ap.add_argument("src")
    ap.add_argument("tgt")
    for flag, kw in FLAGS:
        ap.add_argument(flag, **kw)
    ap.add_argument("--prioritize_f0", type=_bool, default=True)
    ap.add_argument("--weights", default="auto", choices=["auto", "checkpoint", "seeded"],
                    help="(extension) 'seeded' runs with random weights when the released checkpoints are unavailable")
    return ap


def output_dir_for(src: str, tgt: str, ckpt_type: str, post_opt: str, dur_limit) -> str:
    parent = f"{os.path.dirname(os.path.abspath(tgt))}/"
    out = f"{parent}{os.path.basename(src)}_to_{os.path.basename(tgt)}_{ckpt_type}_post_opt_{post_opt}/"
    if dur_limit is not None:
        out = out.replace(parent, parent + f"duration_limit_{dur_limit}_")
    return out


def main(argv=None) -> int:
    a = build_parser().parse_args(argv)
    from .hubconf import knn_vc
    knn = knn_vc(pretrained=True, progress=True, prematched=True, device=a.device, ckpt_type=a.ckpt_type,
                 local_ckpt_dir=a.ckpt_dir, weights=a.weights)
    common = dict(topk=a.topk, device=a.device, prioritize_f0=a.prioritize_f0, ckpt_type=a.ckpt_type,
                  tgt_loudness_db=a.tgt_loudness_db, post_opt=a.post_opt)
    if os.path.isfile(a.src) and os.path.isfile(a.tgt):
        knn.special_match(src_wav_file=a.src, ref_wav_file=a.tgt, **common)
        return 0
    if os.path.isdir(a.src) and os.path.isdir(a.tgt):
        knn.bulk_match(src_dataset_path=a.src, tgt_dataset_path=a.tgt,
                       converted_audio_dir=output_dir_for(a.src, a.tgt, a.ckpt_type, a.post_opt, a.dur_limit),
                       required_subset_file=a.required_subset_file, duration_limit=a.dur_limit, **common)
        return 0
    raise SystemExit("Both inputs must be files or both must be folders.")


if __name__ == "__main__":
    raise SystemExit(main())
