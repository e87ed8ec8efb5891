"""Which stage of the pipelined path differs run to run?  Keeps every item's match outputs and waveform of two pipelined runs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from knn_svc_amd import config as C, matching as M, ops, pipeline, serving, synthetic as S
from knn_svc_amd.matcher import KNeighborsVC
from knn_svc_amd.vocoder import Vocoder, serial_resblocks
from knn_svc_amd.wavlm import WavLMEncoder
dev = torch.device("cuda", 0)
enc = WavLMEncoder(S.seeded_state(S.wavlm_param_spec(C.WAVLM_LARGE, 6), seed=1), C.WAVLM_LARGE, dev, 6)
voc = Vocoder(S.seeded_state(S.generator_param_spec(C.HIFIGAN_V1, "mix"), seed=2), C.HIFIGAN_V1, "mix", dev)
vc = KNeighborsVC(enc, voc, C.HIFIGAN_V1, dev)
n = 30 * C.SAMPLE_RATE
NP, NS = 60, 12
MODE = os.environ.get("MODE", "pipe")
with torch.inference_mode():
    tv = serving.TargetVoice.from_clips(vc, [S.synth_clip(n, seed=5000 + i) for i in range(NP)])
    srcs = [S.synth_clip(n, seed=7000 + i) for i in range(NS)]
    wavs = [torch.from_numpy(w).to(dev) for w, _ in srcs]
    f0s = [torch.from_numpy((f * 1.3).astype(np.float32)).to(dev)[:1500] for _, f in srcs]
    feats = enc.encode_many(wavs, max_batch=32, pow2_batches=True)
    feats = [f.clone() for f in feats]
    torch.cuda.synchronize()

    XG = torch.randn(8192, 4096, device=dev); WG = torch.randn(4096, 4096, device=dev)
    FIX = [t.clone() for t in M.match_features(feats[0], f0s[0], tv.feats, tv.f0, tv.harm, "mix", "post_opt_0.2", pool_prep=tv.prep)]
    torch.cuda.synchronize()
    X256 = torch.randn(15000, 256, device=dev); O256 = torch.empty_like(X256)
    W256 = ops.attach_split(ops.pack_conv_weight(torch.randn(256, 256, 3) / 28).to(dev))
    X128 = torch.randn(120000, 128, device=dev); O128 = torch.empty_like(X128)
    W128 = ops.attach_split(ops.pack_conv_weight(torch.randn(128, 128, 3) / 20).to(dev))
    COND = torch.empty(480000, 32, device=dev)
    CAP = {}
    _orig = ops.concat_reselect

    def spy(idx4, q, q_norm, pool, p_norm, shifted_f0=None, pool_f0=None, concat_weight=0.2):
        out = _orig(idx4, q, q_norm, pool, p_norm, shifted_f0, pool_f0, concat_weight)
        if shifted_f0 is None:
            CAP.setdefault("calls", []).append(dict(idx_in=idx4.clone(), qn=q_norm.clone(), q=q, out=out.clone(),
                                                    pn_sum=p_norm.double().sum(), q_sum=q.double().sum()))
        return out
    ops.concat_reselect = spy

    def run():
        flags = []
        CAP["calls"] = []
        nn, ready = M.grouped_knn(list(range(NS)), dict(enumerate(feats)), tv.feats, tv.prep, flags)
        if os.environ.get("PREKNN") == "1":
            torch.cuda.synchronize()
        keep = {}

        def body(i):
            M.wait_for_neighbours(nn.get(i), ready.get(i), dev)
            r = M.match_features(feats[i], f0s[i], tv.feats, tv.f0, tv.harm, "mix", "post_opt_0.2", nan_flags=flags, pool_prep=tv.prep,
                                 nn32=nn.get(i), return_debug=True)
            keep[i] = r
            return r[:3]
        TAIL = os.environ.get("TAIL", "voc")
        if os.environ.get("NOTAIL") == "1":
            tail = lambda i, r: r[0]
        elif TAIL == "gemm":
            def tail(i, r):
                for _ in range(40):
                    y = XG @ WG
                return y
        elif TAIL == "eager":
            def tail(i, r):
                voc.use_graphs = False
                try:
                    return voc.forward(r[0], r[2], r[1])
                finally:
                    voc.use_graphs = True
        elif TAIL == "w128s":
            def tail(i, r):
                for _ in range(60):
                    ops.conv_gemm(X256, W256, O256, m=15000, n=256, cin=256, taps=3, pad=1, t_in=15000)
                return O256
        elif TAIL == "w160":
            def tail(i, r):
                for _ in range(40):
                    ops.conv_gemm(X128, W128, O128, m=120000, n=128, cin=128, taps=3, pad=1, t_in=120000)
                return O128
        elif TAIL == "synth":
            def tail(i, r):
                for _ in range(8):
                    ops.additive_synth(FIX[2].contiguous(), FIX[1].contiguous(), voc.prenet_w, voc.prenet_b, COND, 32, hop=320, sr=16000, mode=0)
                return COND
        elif TAIL == "fixedin":            # the generator on FIXED inputs (not this item's match outputs)
            tail = lambda i, r: voc.forward(FIX[0], FIX[2], FIX[1])
        else:
            tail = lambda i, r: voc.forward(r[0], r[2], r[1])
        if MODE == "seq":
            ys = [tail(i, body(i)) for i in range(NS)]
        else:
            with serial_resblocks():
                ys = pipeline.LanePipeline(dev, int(os.environ.get("LANES", "3"))).run(list(range(NS)), body, tail)
        torch.cuda.synchronize()
        return nn, keep, [y.clone() for y in ys], list(CAP["calls"])
    run()
    a = run(); b = run()
    for name, get in (("nn32", lambda r, i: r[0][i]), ("of", lambda r, i: r[1][i][0]), ("hw", lambda r, i: r[1][i][1]), ("s0", lambda r, i: r[1][i][2]),
                      ("idx_wavlm", lambda r, i: r[1][i][3]["idx_wavlm"]), ("w_wavlm", lambda r, i: r[1][i][3]["w_wavlm"]),
                      ("idx_harm", lambda r, i: r[1][i][3]["idx_harm"]), ("w_harm", lambda r, i: r[1][i][3]["w_harm"]), ("y", lambda r, i: r[2][i])):
        bad = [i for i in range(NS) if not torch.equal(get(a, i), get(b, i))]
        print(f"{MODE}: {name:10s} differs run to run for items {bad}")

    for k in ("idx_in", "qn", "out", "pn_sum", "q_sum"):
        bad = [j for j in range(len(a[3])) if not torch.equal(a[3][j][k], b[3][j][k])]
        print(f"{MODE}: concat(no f0) call field {k:7s} differs run to run at call positions {bad}")
    # recompute the kernel on the captured inputs, quietly
    ops.concat_reselect = _orig
    torch.cuda.synchronize()
    for j, c in enumerate(b[3]):
        o = _orig(c["idx_in"], c["q"], c["qn"], tv.feats, tv.prep["stats"][0], concat_weight=0.2)
        if not torch.equal(o, c["out"]):
            print(f"  call {j}: recomputed quietly != pipelined output ({int((o != c['out']).any(dim=1).sum())} rows)")


