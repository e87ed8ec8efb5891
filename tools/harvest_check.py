"""GPU Harvest vs the reference's shipped tracks (tests/golden/sample_content_full) and, on 12 s heads, vs oracle/f0_ref.py."""
import sys, time, numpy as np, torch
sys.path.insert(0, ".")
from knn_svc_amd import audio_io, ops
fx = "tests/golden/sample_content_full/"
def cmp(tag, f, r):
    n = min(len(f), len(r)); f = f[:n].astype(np.float64); r = r[:n].astype(np.float64)
    vf, vr = f > 0, r > 0; both = vf & vr; d = np.abs(f[both] - r[both])
    print(f"{tag}: n {n} voicing agree {(vf == vr).mean():.5f} (ref voiced {vr.mean():.3f}) |d|<1e-3 on {(d < 1e-3).mean():.5f} max {d.max():.4g} median {np.median(d):.3g}", flush=True)
for name in ["Danakil-voice_resampled_16000_cut", "Tiken_lead_07_resampled_16000_cut"]:
    x = audio_io.read_wav(fx + name + ".wav")[0][0]
    ref = np.load(fx + name + "_f0.npy")
    xg = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).cuda()
    f = ops.f0_harvest(xg); torch.cuda.synchronize()
    t = time.time()
    for _ in range(3): f = ops.f0_harvest(xg, check_status=False)
    torch.cuda.synchronize(); dt = (time.time() - t) / 3
    print(f"{name}: {len(x) / 16000:.1f} s audio in {dt * 1e3:.1f} ms", flush=True)
    cmp("  gpu vs reference track", f.cpu().numpy(), ref)
    if "--oracle" in sys.argv:
        from oracle import f0_ref
        xs = x[:12 * 16000]
        fo = f0_ref.harvest(xs.astype(np.float64))
        fg = ops.f0_harvest(torch.from_numpy(np.ascontiguousarray(xs)).cuda()).cpu().numpy()
        cmp("  gpu vs oracle (12 s head)", fg, fo)
