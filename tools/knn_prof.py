"""In-kernel phase timing of knn_screen_kernel (a -DKN_KNN_PROF build: tools/knn_prof.sh): per tile start / main loop done / cold
bound done / coarse pass (with its drains) done / last drain done, 10 ns ticks; candidates queued by thread 0 and drains.
  KNNSVC_LIB=knn_svc_amd/libknnsvc_prof.so python tools/knn_prof.py NQ NP"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from knn_svc_amd import ops, _lib, synthetic as S
if sys.argv[1] == "bench":          # the features bench.py searches: WavLM-Large (6 layers, seeded weights) on its synthetic clips
    from knn_svc_amd import config as C
    from knn_svc_amd.wavlm import WavLMEncoder, cat_rows
    enc = WavLMEncoder(S.seeded_state(S.wavlm_param_spec(C.WAVLM_LARGE, 6), seed=1), C.WAVLM_LARGE, "cuda", 6)
    n = 30 * C.SAMPLE_RATE
    with torch.inference_mode():
        feats = enc.encode_many([torch.from_numpy(S.synth_clip(n, seed=2000 + i)[0]).cuda() for i in range(20)] +
                                [torch.from_numpy(S.synth_clip(n, seed=1000)[0]).cuda()], max_batch=32)
    q, p = feats[-1].contiguous(), cat_rows(feats[:-1]).contiguous()
    nq, npool = q.shape[0], p.shape[0]
else:
    nq, npool = int(sys.argv[1]), int(sys.argv[2])
    q = S.clustered_features(nq, 1024, 1).cuda(); p = S.clustered_features(npool, 1024, 2).cuda()
qs, ps = ops.row_norms(q), ops.row_norms(p)
lib = _lib.load()
real_screen = lib.knnsvc_knn_screen
calls = []


def screen(*a):
    rc = real_screen(*a)
    torch.cuda.synchronize()
    buf = (ctypes.c_longlong * (4096 * 8))()
    assert lib.knnsvc_debug_knn_prof(buf, 4096) == 0
    calls.append(np.frombuffer(buf, dtype=np.int64).reshape(4096, 8).copy())
    return rc


for _ in range(3):
    ops.knn_topk(q, p, 32, q_stats=qs, p_stats=ps, check_nan=False)
lib.knnsvc_knn_screen = screen
ops.knn_topk(q, p, 32, q_stats=qs, p_stats=ps, check_nan=False)
lib.knnsvc_knn_screen = real_screen
ops.KNN_DEBUG_COUNTS = []
ops.knn_topk(q, p, 32, q_stats=qs, p_stats=ps, check_nan=False)
for e, c in ops.KNN_DEBUG_COUNTS:
    c = c.float().cpu().numpy()
    print(f" survivors per row, epoch {e}: mean {c.mean():.0f}, median {np.median(c):.0f}, p90 {np.percentile(c, 90):.0f}, p99 {np.percentile(c, 99):.0f}, max {c.max():.0f}; "
          f"rows above 1000: {int((c > 1000).sum())}")
ops.KNN_DEBUG_COUNTS = None
fl = ops.knn_topk(q, p, 32, q_stats=qs, p_stats=ps, check_nan=False, return_flag=True)[2]
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    ops.knn_topk(q, p, 32, q_stats=qs, p_stats=ps, check_nan=False)
e1.record(); torch.cuda.synchronize()
print(f"Nq={nq} Np={npool} epochs={ops.knn_epochs(nq, npool)} flag={int(fl.item())} search {e0.elapsed_time(e1) / 5:.3f} ms (prof build)")
for e, t in enumerate(calls):
    t = t.astype(np.float64)
    live = t[:, 4] > t[:, 0]
    t = t[live]
    us = t[:, :5] * 0.01
    d = np.diff(us, axis=1)
    if e == 0:
        c = t[t[:, 7] > t[:, 1]]
        if len(c):
            xch = bool((c[:, 5] > c[:, 7]).all())          # slots 5 / 6 are only stamped by tiles that exchange bounds
            mid = (f"publish + wait {np.mean(c[:, 5] - c[:, 7]) * 0.01:.1f}, read + select {np.mean(c[:, 6] - c[:, 5]) * 0.01:.1f}, "
                   f"set_row + atomics + barrier {np.mean(c[:, 2] - c[:, 6]) * 0.01:.1f}, " if xch else
                   f"(no exchange: fewer than 20 column tiles) set_row + barrier {np.mean(c[:, 2] - c[:, 7]) * 0.01:.1f}, ")
            print(f" cold tiles ({len(c)}): local bound {np.mean(c[:, 7] - c[:, 1]) * 0.01:.1f} us, " + mid +
                  f"coarse + drains {np.mean(c[:, 3] - c[:, 2]) * 0.01:.1f} (p90 {np.percentile(c[:, 3] - c[:, 2], 90) * 0.01:.1f}), last drain "
                  f"{np.mean(c[:, 4] - c[:, 3]) * 0.01:.1f}, main loop {np.mean(c[:, 1] - c[:, 0]) * 0.01:.1f}, tile {np.mean(c[:, 4] - c[:, 0]) * 0.01:.1f} "
                  f"(max {np.max(c[:, 4] - c[:, 0]) * 0.01:.1f}), kernel span {(c[:, 4].max() - c[:, 0].min()) * 0.01:.1f} us")
        continue
    print(f" epoch {e}: {len(t)} tiles, span {us[:, 4].max() - us[:, 0].min():.1f} us; per tile (mean / p90 us): main loop {d[:, 0].mean():.1f} / {np.percentile(d[:, 0], 90):.1f}, "
          f"bound {d[:, 1].mean():.1f} / {np.percentile(d[:, 1], 90):.1f}, coarse + drains {d[:, 2].mean():.1f} / {np.percentile(d[:, 2], 90):.1f}, "
          f"last drain {d[:, 3].mean():.1f} / {np.percentile(d[:, 3], 90):.1f}; queued by thread 0: {t[:, 5].mean():.1f}, drains {t[:, 6].mean():.1f}")
