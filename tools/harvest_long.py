"""GPU Harvest on long inputs (the 60 s sample clip tiled to 5 and 10 minutes): time, peak memory, frame count."""
import sys, time, numpy as np, torch
sys.path.insert(0, ".")
from knn_svc_amd import audio_io, ops
x = audio_io.read_wav("tests/golden/sample_content_full/Tiken_lead_07_resampled_16000_cut.wav")[0][0]
ref = np.load("tests/golden/sample_content_full/Tiken_lead_07_resampled_16000_cut_f0.npy")
for reps in (5, 10):
    xl = torch.from_numpy(np.tile(x, reps)).cuda()
    torch.cuda.synchronize(); t = time.time()
    f = ops.f0_harvest(xl); torch.cuda.synchronize(); dt = time.time() - t
    f = f.cpu().numpy()
    print(f"{reps * 60} s of audio: {dt * 1e3:.1f} ms, {torch.cuda.max_memory_allocated() / 1e9:.1f} GB peak, {len(f)} frames, "
          f"{float((f > 0).mean()):.3f} voiced", flush=True)
