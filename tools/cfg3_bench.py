"""BASELINE cfg 3 on one GPU: dataset mode (KNeighborsVC.bulk_match, reference loop ddsp_matcher.py:1073-1133) over a synthetic
corpus — S speakers x U utterances of 5-10 s with f0 caches, every speaker converted to every other one, target pool limited to
--dur-limit seconds (the reference compares it in seconds, ddsp_prematch_dataset.py:408-411), ckpt_type=mix, post_opt_0.2,
seeded weights of the real architectures.  Prints one JSON line: xRT = converted source seconds / wall seconds of bulk_match
(file reads, f0 loads, pool encoding — once per file thanks to the pool store —, matching, vocoding, WAV writes all inside),
the graph-cache population (buckets, not utterance lengths) and the per-stage host clock.

    python tools/cfg3_bench.py [--speakers 4] [--utts 80] [--dur-limit 600] [--out DIR]
"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from knn_svc_amd import audio_io, config as C, hubconf, synthetic as S          # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--speakers", type=int, default=4)
    ap.add_argument("--utts", type=int, default=80)
    ap.add_argument("--dur-limit", type=float, default=600.0)
    ap.add_argument("--out", default=None)
    ap.add_argument("--passes", type=int, default=2, help="2: the second pass runs with every graph bucket captured (steady state)")
    a = ap.parse_args()
    root = a.out or tempfile.mkdtemp(prefix="cfg3_")
    data = os.path.join(root, "data")
    rng = np.random.default_rng(3)
    secs = 0.0
    t0 = time.perf_counter()
    for s in range(a.speakers):
        d = os.path.join(data, f"spk{s:02d}")
        os.makedirs(d, exist_ok=True)
        for u in range(a.utts):
            n = int(rng.uniform(5.0, 10.0) * C.SAMPLE_RATE)
            w, f0 = S.synth_clip(n, 100000 + 1000 * s + u)
            audio_io.write_wav_pcm16(os.path.join(d, f"u{u:03d}.wav"), w, C.SAMPLE_RATE)
            np.save(os.path.join(d, f"u{u:03d}_f0.npy"), (f0 * (1.0 + 0.15 * s)).astype(np.float32))
            secs += n / C.SAMPLE_RATE
    gen_s = time.perf_counter() - t0
    os.environ["KNNSVC_SEEDED_WEIGHTS"] = "1"
    vc = hubconf.knn_vc(pretrained=True, ckpt_type="mix", device="cuda")
    res = []
    for p in range(a.passes):
        from knn_svc_amd import matching
        matching._POOL_CACHE = None                       # every pass encodes every file once (cold pool store)
        out = os.path.join(root, f"converted{p}")
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        written = vc.bulk_match(data, data, out, ckpt_type="mix", post_opt="post_opt_0.2", duration_limit=a.dur_limit)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        conv_secs = secs * (a.speakers - 1)               # every utterance is converted to every other speaker
        res.append(dict(wall_s=round(dt, 3), files=len(written), xrt=round(conv_secs / dt, 1)))
    print(json.dumps({"workload": f"cfg 3: dataset mode, {a.speakers} speakers x {a.utts} utterances (5-10 s, {secs:.0f} s of audio), "
                                  f"dur_limit {a.dur_limit:.0f} s, mix, post_opt_0.2, 1 GPU",
                      "converted_audio_s": round(secs * (a.speakers - 1), 1), "passes": res,
                      "value": res[-1]["xrt"], "unit": "x real-time",
                      "encoder_graphs": [list(k) for k in vc.wavlm._graphs], "vocoder_graph_buckets": list(vc.hifigan._graphs),
                      "corpus_generation_s": round(gen_s, 1), "data": "synthetic", "weights": "seeded"}))


if __name__ == "__main__":
    main()
