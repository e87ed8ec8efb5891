"""BASELINE cfg 5 on one GPU's share: S concurrent 30 s sources against one P-minute target pool, ckpt_type=mix,
post_opt_0.2, through the dataset-mode pipeline (matching.match_features on LanePipeline lanes, the generator as the
tail stage).  Prints one JSON line: warm xRT (pool features resident, sources encoded inside the timed region),
ms per source, and the kNN rate.  Seeded random weights, synthetic clips.

    python tools/cfg5_bench.py --sources 32 --pool-minutes 60 [--lanes 3] [--sequential]
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from knn_svc_amd import config as C, ops, synthetic as S          # noqa: E402
from knn_svc_amd.matching import grouped_knn, match_features, prepare_pool, side_features, wait_for_neighbours   # noqa: E402
from knn_svc_amd.pipeline import LanePipeline                      # noqa: E402
from knn_svc_amd.vocoder import Vocoder                            # noqa: E402
from knn_svc_amd.wavlm import WavLMEncoder                         # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sources", type=int, default=32)
    ap.add_argument("--pool-minutes", type=int, default=60)
    ap.add_argument("--lanes", type=int, default=3)
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--sequential", action="store_true", help="one item after the other on the current stream")
    ap.add_argument("--per-item-knn", action="store_true", help="one kNN call per source (1500 queries each) instead of one for all")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    enc = WavLMEncoder(S.seeded_state(S.wavlm_param_spec(C.WAVLM_LARGE, 6), seed=1), C.WAVLM_LARGE, dev, 6)
    voc = Vocoder(S.seeded_state(S.generator_param_spec(C.HIFIGAN_V1, "mix"), seed=2), C.HIFIGAN_V1, "mix", dev)
    g = lambda x: torch.from_numpy(x).to(dev)
    n = 30 * C.SAMPLE_RATE
    with torch.inference_mode():
        # ---- target pool (resident afterwards: "warm") --------------------------------------------------------
        clips = [S.synth_clip(n, seed=5000 + i) for i in range(2 * a.pool_minutes)]
        t0 = time.perf_counter()
        feats, f0s, harms = [], [], []
        for b in range(0, len(clips), 20):
            ws = [g(w) for w, _ in clips[b:b + 20]]
            fs = enc.encode_many(ws, max_batch=32)
            for w, (_, f), ft in zip(ws, clips[b:b + 20], fs):
                f0, harm, _ = side_features(w, f, ft.shape[0])
                feats.append(ft); f0s.append(f0); harms.append(harm)
        P, Pf0, Ph = torch.cat(feats).contiguous(), torch.cat(f0s).contiguous(), torch.cat(harms).contiguous()
        del feats
        prep = prepare_pool(P)
        torch.cuda.synchronize()
        pool_s = time.perf_counter() - t0
        srcs = [S.synth_clip(n, seed=7000 + i) for i in range(a.sources)]
        src_w = [g(w) for w, _ in srcs]
        src_f = [f * 1.3 for _, f in srcs]
        pipe = LanePipeline(dev, a.lanes)

        def run():
            flags = []
            q = enc.encode_many(src_w, max_batch=32)
            qf0 = [side_features(w, f, ft.shape[0])[0] for w, f, ft in zip(src_w, src_f, q)]     # as get_complete_spk_pool does
            # as the dataset-mode path does (matching.grouped_knn): frames of several items per search (fused screen + refine route),
            # group after group on a stream of its own, each item waiting only for its group
            if a.per_item_knn:
                nn = [None] * a.sources
            else:
                nn, ready = grouped_knn(list(range(a.sources)), q, P, prep, flags)

            def head(i):
                if not a.per_item_knn:
                    wait_for_neighbours(nn[i], ready.get(i), dev)
                return match_features(q[i], qf0[i], P, Pf0, Ph, "mix", "post_opt_0.2", nan_flags=flags, pool_prep=prep, nn32=nn[i])
            tail = lambda i, r: voc.forward(r[0], r[2], r[1])
            if a.sequential:
                ys = [tail(i, head(i)) for i in range(a.sources)]
            else:
                from knn_svc_amd.vocoder import serial_resblocks
                with serial_resblocks():
                    ys = pipe.run(range(a.sources), head, tail)
            peak = torch.stack([y.abs().max() for y in ys])
            for f in flags:
                ops.raise_if_nan(f)
            assert bool(torch.isfinite(peak).all())
            return ys
        run(); run()                                # first sight runs eagerly, the second captures the graphs
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.reps):
            run()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / a.reps
    nq, npool = a.sources * 1500, P.shape[0]
    print(json.dumps({"workload": f"cfg5 share: {a.sources} x 30 s sources vs {a.pool_minutes}-min pool ({npool} frames), mix, post_opt_0.2, warm pool",
                      "xRT": round(a.sources * 30 / dt, 1), "ms_per_source": round(dt / a.sources * 1e3, 2), "seconds": round(dt, 3),
                      "lanes": 0 if a.sequential else a.lanes, "pool_encode_seconds_once": round(pool_s, 2),
                      "knn_pairs_per_s": round(nq * npool / dt, 0)}))


if __name__ == "__main__":
    main()
