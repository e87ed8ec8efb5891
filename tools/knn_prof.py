"""In-kernel phase timing of knn_screen_kernel (a -DKN_KNN_PROF build: tools/knn_prof.sh): per tile start / main loop done / cold
bound done / coarse pass (with its drains) done / last drain done, 10 ns ticks; candidates queued by thread 0 and drains.
  KNNSVC_LIB=knn_svc_amd/libknnsvc_prof.so python tools/knn_prof.py NQ NP"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from knn_svc_amd import ops, _lib, synthetic as S
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
print(f"Nq={nq} Np={npool} epochs={ops.knn_epochs(nq, npool)}")
for e, t in enumerate(calls):
    t = t.astype(np.float64)
    live = t[:, 4] > t[:, 0]
    t = t[live]
    us = t[:, :5] * 0.01
    d = np.diff(us, axis=1)
    if e == 0:
        c = t[t[:, 7] > t[:, 1]]
        if len(c):
            print(f" cold tiles ({len(c)}): local bound {np.mean(c[:, 7] - c[:, 1]) * 0.01:.1f} us, publish + wait {np.mean(c[:, 5] - c[:, 7]) * 0.01:.1f}, "
                  f"read + select {np.mean(c[:, 6] - c[:, 5]) * 0.01:.1f}, set_row + atomics + barrier {np.mean(c[:, 2] - c[:, 6]) * 0.01:.1f}")
        continue
    print(f" epoch {e}: {len(t)} tiles, span {us[:, 4].max() - us[:, 0].min():.1f} us; per tile (mean / p90 us): main loop {d[:, 0].mean():.1f} / {np.percentile(d[:, 0], 90):.1f}, "
          f"bound {d[:, 1].mean():.1f} / {np.percentile(d[:, 1], 90):.1f}, coarse + drains {d[:, 2].mean():.1f} / {np.percentile(d[:, 2], 90):.1f}, "
          f"last drain {d[:, 3].mean():.1f} / {np.percentile(d[:, 3], 90):.1f}; queued by thread 0: {t[:, 5].mean():.1f}, drains {t[:, 6].mean():.1f}")
