"""One eager generator forward (1500 frames, branches in series) with a log of every conv / fused-pair / synth launch and its
ALGORITHMIC bytes, for tools/pmc_generator.sh (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes + a plain kernel trace; the
report joins them by launch order).  Writes gpurun_out/<OUT>/launch_log.json.
    python tools/pmc_generator.py OUT [N]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from knn_svc_amd import config as C, ops, synthetic as S
from knn_svc_amd import vocoder as V

out_dir = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", sys.argv[1]); os.makedirs(out_dir, exist_ok=True)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
voc = V.Vocoder(S.seeded_state(S.generator_param_spec(C.HIFIGAN_V1, "mix"), 2), C.HIFIGAN_V1, "mix", "cuda")
voc.use_graphs = False
g = torch.Generator().manual_seed(0)
c = torch.randn(N, 1024, generator=g).cuda(); harm = (torch.rand(N, 49, generator=g) * 0.02).cuda()
_, f0 = S.synth_clip(N * 320, 5); f0 = torch.from_numpy(f0[:N].copy()).cuda()
LOG = []
_conv, _pair, _synth = ops.conv_gemm, ops.resblock_pair, ops.additive_synth


def conv(x, w, out, *, m, n, cin, taps=1, stride=1, t_in=None, resid=None, accumulate=False, convt_u=0, convt_cout=0, **kw):
    r = _conv(x, w, out, m=m, n=n, cin=cin, taps=taps, stride=stride, t_in=t_in, resid=resid, accumulate=accumulate, convt_u=convt_u,
              convt_cout=convt_cout, **kw)
    rows_in = t_in if t_in is not None else m
    cout_total = n                                           # transposed conv: n = u * cout columns, scattered to u * m rows of cout
    b = 4 * (rows_in * cin + m * cout_total + n * cin * taps) + (4 * m * cout_total if resid is not None else 0) + (4 * m * cout_total if accumulate else 0)
    LOG.append(dict(family="conv_pair" if False else ("conv_gemm2win" if ops.last_conv_kernel().startswith("W") else "conv_gemm_other"),
                    kernel=ops.last_conv_kernel(), m=int(m), n=int(n), cin=int(cin), taps=int(taps), resid=resid is not None,
                    accumulate=bool(accumulate), alg_bytes=int(b)))
    return r


def pair(x, w1, b1, w2, b2, out, *, t, channels, taps, **kw):
    r = _pair(x, w1, b1, w2, b2, out, t=t, channels=channels, taps=taps, **kw)
    LOG.append(dict(family="conv_pair", kernel="pair", m=int(t), n=int(channels), cin=int(channels), taps=int(taps), resid=True, accumulate=False,
                    alg_bytes=int(8 * t * channels + 2 * 4 * channels * channels * taps)))       # x in, out out, two weight sets (t1 never leaves the chip)
    return r


def synth(f0_, amp, pw, pb, cond, ld_cond, *, hop=320, **kw):
    r = _synth(f0_, amp, pw, pb, cond, ld_cond, hop=hop, **kw)
    n = f0_.numel()
    LOG.append(dict(family="additive_synth", kernel="synth", m=int(n * hop), n=int(pb.numel()), cin=int(amp.shape[1] if amp is not None else 0), taps=3,
                    resid=False, accumulate=False, alg_bytes=int(4 * n * (1 + (amp.shape[1] if amp is not None else 0)) + 4 * n * hop * pb.numel())))
    return r


with torch.inference_mode(), V.serial_resblocks():
    voc.forward(c, f0, harm)                                 # warm-up (weight splits, allocator)
    torch.cuda.synchronize()
    ops.conv_gemm, ops.resblock_pair, ops.additive_synth = conv, pair, synth
    voc.forward(c, f0, harm)
    torch.cuda.synchronize()
json.dump(dict(frames=N, launches=LOG), open(os.path.join(out_dir, "launch_log.json"), "w"))
print(f"{len(LOG)} launches logged:", {f: sum(1 for l in LOG if l['family'] == f) for f in sorted({l['family'] for l in LOG})})
