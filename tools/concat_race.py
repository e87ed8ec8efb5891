"""Minimal reproducer for the run-to-run nondeterminism of round 3 (csrc/Makefile): concat_reselect_pipe_kernel —
one workgroup, frame-sequential, its distance sums are plain fp32 reductions — launched RUNS times on the same inputs while a
second stream keeps MFMA-issuing workgroups (the generator's C = 256 stage: 64-row windowed convolutions, one per CU and more) on
the chip, so that they share the re-selection's CU.  Counts the launches whose output differs from the quiet reference.
    python tools/concat_race.py [RUNS]            (KNNSVC_LIB selects the library: product or the SLP probe)
Prints one JSON line."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from knn_svc_amd import _lib, ops, synthetic as S


def run(runs=40, frames=1500, pool_rows=30000):
    dev = torch.device("cuda", 0)
    sm = lambda x: (x + torch.roll(x, 1, 0) + torch.roll(x, 2, 0)) / 3            # temporal continuity: concat costs matter
    q = sm(S.clustered_features(frames, 1024, 1, n_centres=80)).to(dev)
    pool = sm(S.clustered_features(pool_rows, 1024, 2, n_centres=80)).to(dev)
    qn, _ = ops.row_norms(q); pn, _ = ops.row_norms(pool)
    nn, _ = ops.knn_topk(q, pool, 32)
    idx = nn[:, :4].contiguous()
    ref = ops.concat_reselect(idx, q, qn, pool, pn, concat_weight=0.2)
    again = ops.concat_reselect(idx, q, qn, pool, pn, concat_weight=0.2)
    torch.cuda.synchronize()
    quiet_ok = bool(torch.equal(ref, again))
    side = torch.cuda.Stream()
    x256 = torch.randn(15000, 256, device=dev); o256 = torch.empty_like(x256)
    w256 = ops.attach_split(ops.pack_conv_weight(torch.randn(256, 256, 3) / 28).to(dev))
    x128 = torch.randn(120000, 128, device=dev); o128 = torch.empty_like(x128)
    w128 = ops.attach_split(ops.pack_conv_weight(torch.randn(128, 128, 3) / 20).to(dev))
    kernels = set()
    differ, first_bad = 0, None
    for rep in range(runs):
        with torch.cuda.stream(side):
            for _ in range(60):          # ~7 ms of convolutions: the re-selection (6.6 ms) runs underneath them
                ops.conv_gemm(x256, w256, o256, m=15000, n=256, cin=256, taps=3, pad=1, t_in=15000)
                kernels.add(ops.last_conv_kernel())
                ops.conv_gemm(x128, w128, o128, m=120000, n=128, cin=128, taps=3, pad=1, t_in=120000)
                kernels.add(ops.last_conv_kernel())
        out = ops.concat_reselect(idx, q, qn, pool, pn, concat_weight=0.2)
        torch.cuda.synchronize()
        if not torch.equal(out, ref):
            differ += 1
            if first_bad is None:
                first_bad = int((out != ref).any(1).nonzero()[0])
    return dict(lib=os.path.basename(_lib.LIB_PATH), runs=runs, differ=differ, quiet_repeat_equal=quiet_ok, first_differing_frame=first_bad,
                co_runners=sorted(kernels))


if __name__ == "__main__":
    print(json.dumps(run(int(sys.argv[1]) if len(sys.argv) > 1 else 40)))
