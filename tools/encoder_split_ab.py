"""What-if: the encoder's 21 independent 30 s chunks as ONE batch on one stream (the bench's form) against 2 / 3 sub-batches
on streams measured to run side by side (pipeline.new_stream) — does one sub-batch's memory-bound / tail phases hide under the
other's GEMMs?  Prints ms per 21 chunks for each form (eager one-call encodes; events on the main stream around fork .. join).
Usage: python tools/encoder_split_ab.py [reps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from knn_svc_amd import config as C, pipeline, synthetic as S          # noqa: E402
from knn_svc_amd.wavlm import WavLMEncoder                                 # noqa: E402


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    dev = torch.device("cuda:0")
    enc = WavLMEncoder(S.seeded_state(S.wavlm_param_spec(C.WAVLM_LARGE, 6), seed=1), C.WAVLM_LARGE, dev, 6)
    enc.use_graphs = False
    L = 30 * C.SAMPLE_RATE
    wav = torch.stack([torch.from_numpy(S.synth_clip(L, seed=2000 + i)[0]) for i in range(21)]).to(dev)
    main_s = pipeline.new_stream(dev, priority=0, kind="split_main", owner=None)       # (the bench's front half runs on a lane stream too)
    torch.cuda.set_stream(main_s)
    sides = []
    for i in range(2):
        sides.append(pipeline.new_stream(dev, priority=0, kind=f"split{i}", owner=None, must=[main_s] + sides))

    def run(parts):
        """parts: list of row ranges; part 0 on the main stream, part i on side stream i - 1"""
        outs = []
        for s in sides[:len(parts) - 1]:
            s.wait_stream(main_s)
        for i, (lo, hi) in enumerate(parts):
            st = main_s if i == 0 else sides[i - 1]
            with torch.cuda.stream(st):
                outs.append(enc._encode_batch(wav[lo:hi]))
        for s in sides[:len(parts) - 1]:
            main_s.wait_stream(s)
        return outs

    forms = {
        "1 x 21 (one stream)": [(0, 21)],
        "11 + 10 (two streams)": [(0, 11), (11, 21)],
        "7 + 7 + 7 (three streams)": [(0, 7), (7, 14), (14, 21)],
        "11 + 10 in series (one stream)": None,
    }
    ref = torch.cat([o.reshape(-1, o.shape[-1]) for o in run(forms["1 x 21 (one stream)"])])
    for name, parts in forms.items():
        ts = []
        for r in range(reps + 2):
            torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            if parts is None:
                outs = [enc._encode_batch(wav[0:11]), enc._encode_batch(wav[11:21])]
            else:
                outs = run(parts)
            e1.record()
            torch.cuda.synchronize()
            if r >= 2:
                ts.append(e0.elapsed_time(e1))
        got = torch.cat([o.reshape(-1, o.shape[-1]) for o in outs])
        err = float((got - ref).abs().max())
        print(f"{name:34s} {min(ts):7.3f} ms min  {sum(ts) / len(ts):7.3f} ms mean   max |diff| vs one batch {err:.2e}", flush=True)


if __name__ == "__main__":
    t0 = time.time()
    main()
    print(f"({time.time() - t0:.0f} s)")
