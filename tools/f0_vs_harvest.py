import sys, numpy as np, torch
sys.path.insert(0, ".")
from knn_svc_amd import audio_io, ops
for n in ("src", "tgt"):
    x, sr = audio_io.read_wav(f"tests/golden/sample_content/{n}.wav")
    ref = np.load(f"tests/golden/sample_content/{n}_f0.npy")
    est = ops.f0_yin(torch.from_numpy(x[0]).cuda()).cpu().numpy()
    m = min(len(ref), len(est)); ref, est = ref[:m], est[:m]
    both = (ref > 0) & (est > 0)
    rel = np.abs(est[both] - ref[both]) / ref[both]
    print(n, "frames", m, "harvest voiced", int((ref > 0).sum()), "yin voiced", int((est > 0).sum()), "both", int(both.sum()),
          "voicing agreement %.3f" % np.mean((ref > 0) == (est > 0)), "median rel dev %.4f" % np.median(rel), "90th pct %.4f" % np.percentile(rel, 90),
          "octave errors", int((rel > 0.3).sum()))
