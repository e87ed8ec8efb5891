"""``KNeighborsVC`` — the orchestrator object the reference's entry points hand out.

Mirror of ddsp_matcher.py:303-1155 restricted to its live methods: ``vocode``
(:375-406), ``special_match`` (:937-1023) and ``bulk_match`` (:1027-1155), with the
same argument names, output file naming and ignored arguments (``topk``,
``tgt_loudness_db``).  Differences that are deliberate: ``special_match`` returns
the waveform instead of calling ``sys.exit()`` after saving (the CLI exits 0, the
observable behaviour), and ``bulk_match`` does not ``rm -rf`` a hard-coded cache
directory (ddsp_matcher.py:1066-1068).
"""
from __future__ import annotations

import csv
import os
from pathlib import Path

import torch

from . import audio_io, config as C, dist as kdist, ops
from .matching import match_at_inference_time
from .vocoder import Vocoder
from .wavlm import WavLMEncoder

# one-hot on layer 6 of 25 (ddsp_matcher.py:88-89, knnvc_utils.py:3-6)
SPEAKER_INFORMATION_WEIGHTS = [1.0 if i == C.MATCH_LAYER else 0.0 for i in range(25)]


class KNeighborsVC:
    def __init__(self, wavlm: WavLMEncoder, hifigan: Vocoder, hifigan_cfg: dict, device="cuda") -> None:
        self.wavlm = wavlm
        self.hifigan = hifigan
        self.h = hifigan_cfg
        self.device = torch.device(device)
        self.weighting = torch.zeros(wavlm.cfg["encoder_layers"] + 1)
        self.weighting[wavlm.n_layers] = 1.0
        self.sr = hifigan_cfg["sampling_rate"]
        self.hop_length = 320

    @torch.inference_mode()
    def vocode(self, c, f0=None, harmonics_out_feats_weighted=None):
        """c (bs, seq_len, c_dim), f0 (bs, seq_len, 1), harmonics (bs, seq_len, 49) -> (bs, seq_len*320)."""
        if f0 is None:
            raise NotImplementedError("the f0-free 'wavlm_only_original' generator (hifigan/models.py) is missing "
                                      "from the reference snapshot and unsupported")
        outs = []
        for b in range(c.shape[0]):
            harm = harmonics_out_feats_weighted[b] if harmonics_out_feats_weighted is not None else None
            outs.append(self.hifigan.forward(c[b].to(self.device).float(), f0[b].reshape(-1).to(self.device).float(),
                                             None if harm is None else harm.to(self.device).float()))
        wav = torch.stack(outs, 0)
        self._check_finite(wav)
        return wav

    @staticmethod
    def _check_finite(wav):
        # operand scales follow the data (range slots), so a non-finite sample means a non-finite input or weight
        if not bool(torch.isfinite(wav).all()):
            raise ops.KnnSvcError("vocode: non-finite waveform (non-finite features, f0, harmonics or weights)")

    def _vocode_async(self, c, f0, harm=None):
        """One utterance, enqueue only (no host sync): the tail stage of the dataset-mode pipeline."""
        return self.hifigan.forward(c.float(), f0.reshape(-1).float(), None if harm is None else harm.float())

    @torch.inference_mode()
    def special_match(self, src_wav_file, ref_wav_file, topk: int = 4, device=None, prioritize_f0=True,
                      ckpt_type="wavlm_only", tgt_loudness_db=-16, post_opt="no_post_opt", save=True):
        f0only = "wavlm_only" in ckpt_type or "no_harm_no_amp" in ckpt_type
        if "wavlm_only_original" in ckpt_type:
            raise NotImplementedError("wavlm_only_original needs hifigan/models.py, absent upstream")
        key = str(src_wav_file)
        if not f0only:
            of, hw, _a, sf0 = match_at_inference_time(Path(src_wav_file), Path(ref_wav_file), self.wavlm,
                                                      self.weighting, self.weighting, topk=topk, device=self.device,
                                                      prioritize_f0=prioritize_f0, ckpt_type=ckpt_type, post_opt=post_opt)
            pred = self.vocode(of[key][None], sf0[key][None, :, None], hw[key][None]).squeeze()
        else:
            # the reference does not forward post_opt on this branch (ddsp_matcher.py:970)
            of, _a, sf0 = match_at_inference_time(Path(src_wav_file), Path(ref_wav_file), self.wavlm,
                                                  self.weighting, self.weighting, topk=topk, device=self.device,
                                                  prioritize_f0=prioritize_f0, ckpt_type=ckpt_type)
            pred = self.vocode(of[key][None], sf0[key][None, :, None]).squeeze()
        src_id = os.path.basename(src_wav_file).split(".")[0]
        ref_id = os.path.basename(ref_wav_file).split(".")[0]
        out_file = str(Path(src_wav_file).parent) + "/" + src_id + "_to_" + ref_id + f"_knn_{ckpt_type}_{post_opt}.wav"
        if save:
            print("->", out_file)
            audio_io.save_audio(out_file, pred.detach().cpu().numpy(), sample_rate=16000)
        return pred

    @torch.inference_mode()
    def many_to_one(self, src_files, ref_wav_file, converted_audio_dir=None, ckpt_type="mix", post_opt="post_opt_0.2",
                    duration_limit=None, target=None):
        """BASELINE cfg 5 (no counterpart upstream: the reference would call ``special_match`` once per source and rebuild the
        target pool each time, ddsp_matcher.py:937-1023): MANY sources against ONE target pool that is built once and stays
        resident, all sources through the stream pipeline as one batch (knn_svc_amd/serving.py).  Under a process group the
        sources are dealt over the ranks (no collective).  ``target``: a serving.TargetVoice to reuse.  -> written paths (all
        ranks' on every rank), named like ``special_match``'s outputs."""
        from . import serving
        if target is None:
            target = serving.TargetVoice(self, ref_wav_file, duration_limit)
        mine = kdist.my_share([str(p) for p in src_files])
        written = serving.BatchConverter(self, target, ckpt_type, post_opt).convert_files(mine, converted_audio_dir)
        return kdist.gather_paths(written)

    @torch.inference_mode()
    def bulk_match(self, src_dataset_path, tgt_dataset_path, converted_audio_dir, topk: int = 4, device=None,
                   prioritize_f0=True, ckpt_type="mix", tgt_loudness_db=-16, required_subset_file=None,
                   post_opt="no_post_opt", duration_limit=None):
        assert os.path.isdir(src_dataset_path) and os.path.isdir(tgt_dataset_path)
        Path(converted_audio_dir).mkdir(parents=True, exist_ok=True)
        spk = lambda root: sorted([p for p in Path(root).iterdir() if p.is_dir() and "f0_cache" not in os.path.basename(p)])
        src_spk, tgt_spk = spk(src_dataset_path), spk(tgt_dataset_path)
        if src_dataset_path != tgt_dataset_path:
            assert len(set(src_spk).intersection(set(tgt_spk))) == 0
        assert len(src_spk) > 0, [f"Are you sure {src_dataset_path} is a FOLDER containing speaker folders, i.e. dataset root"]
        assert len(tgt_spk) > 0, [f"Are you sure {tgt_dataset_path} is a FOLDER containing speaker folders, i.e. dataset root"]
        required = None
        if required_subset_file:
            with open(required_subset_file, "r") as fp:
                rows = list(csv.reader(fp, delimiter=",", quotechar='"'))
            required = [r[2] for i, r in enumerate(rows) if i != 0 and r[-1] == "0"]
        f0only = "wavlm_only" in ckpt_type or "no_harm_no_amp" in ckpt_type
        # Files are encoded and written by a small thread pool while the next speaker pair is on the GPU (a pair's 80 files cost
        # the host ~20 ms during which the device idled: 7 % of a cfg-3 run, tools/timeline_cmd.sh); the paths are collected in
        # order at the end, where a failed write raises.
        from concurrent.futures import ThreadPoolExecutor
        writer = ThreadPoolExecutor(max_workers=2, thread_name_prefix="knnsvc-writer")
        pending = []
        pairs = [(s, t) for i, s in enumerate(src_spk) for j, t in enumerate(tgt_spk)
                 if not (src_dataset_path == tgt_dataset_path and i == j)]
        # One process per GPU, two ways to share a dataset run, never both at once:
        #   * default: speaker pairs are independent (clip-DP) and are dealt round-robin over the ranks, every rank
        #     handles ITS pairs alone (pool_sharded=False: no collective may run, the other ranks are on other pairs);
        #   * KNNSVC_POOL_SHARD=1 (BASELINE cfg 4, pools too large / too slow for one GPU): EVERY rank walks EVERY pair
        #     in the same order, each pair's target pool and source files are encoded in shares over the ranks, the kNN
        #     is searched per shard, and the utterances of the pair are dealt over the ranks for matching + vocoding.
        rank, ws = kdist.world()
        shard = os.environ.get("KNNSVC_POOL_SHARD") == "1" and ws > 1
        my_pairs = pairs if shard else kdist.my_share(pairs)
        try:
            for s, t in my_pairs:
                print(f"{s} -> {t}")
                common = dict(topk=topk, device=self.device, prioritize_f0=prioritize_f0, ckpt_type=ckpt_type,
                              src_dataset_path=src_dataset_path, tgt_dataset_path=tgt_dataset_path,
                              required_subset=required, duration_limit=duration_limit,
                              pool_sharded=shard, share_items=shard)
                # the generator of every utterance is the tail stage of the match pipeline (same kernels and inputs as
                # `vocode` after the fact, ddsp_matcher.py:1114-1128, but enqueued under the next utterances' matching);
                # one finiteness check per speaker pair instead of one host sync per utterance
                preds = {}
                if not f0only:
                    match_at_inference_time(Path(s), Path(t), self.wavlm, self.weighting, self.weighting, post_opt=post_opt,
                                            vocode_fn=self._vocode_async, waves_out=preds, **common)
                else:
                    match_at_inference_time(Path(s), Path(t), self.wavlm, self.weighting, self.weighting,
                                            vocode_fn=lambda c, f0, _h: self._vocode_async(c, f0), waves_out=preds, **common)
                if preds:      # max |x| of a waveform is NaN / inf iff the waveform holds one
                    self._check_finite(torch.stack([p.abs().max() for p in preds.values()]))
                for k, pred in preds.items():
                    out = os.path.join(converted_audio_dir, os.path.basename(s), os.path.basename(k).split(".")[0],
                                       os.path.basename(t) + "." + os.path.basename(k).split(".")[-1])
                    Path(out).parent.mkdir(parents=True, exist_ok=True)
                    assert pred.dim() == 1
                    pending.append(writer.submit(audio_io.save_audio, out, pred[None, :].cpu().numpy(), self.sr))
                print(f"{os.path.basename(s)}, {os.path.basename(t)} -> {converted_audio_dir}")
            written = [f.result() for f in pending]
        finally:
            writer.shutdown(wait=True)
        return kdist.gather_paths(written)
