"""concat_reselect beside SMALL kernels that can share its CU (the generator's co-residents)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from knn_svc_amd import config as C, ops, serving, synthetic as S
from knn_svc_amd.matcher import KNeighborsVC
from knn_svc_amd.vocoder import Vocoder
from knn_svc_amd.wavlm import WavLMEncoder
dev = torch.device("cuda", 0)
enc = WavLMEncoder(S.seeded_state(S.wavlm_param_spec(C.WAVLM_LARGE, 6), seed=1), C.WAVLM_LARGE, dev, 6)
n = 30 * C.SAMPLE_RATE
with torch.inference_mode():
    pool = torch.cat(enc.encode_many([torch.from_numpy(S.synth_clip(n, seed=5000 + i)[0]).to(dev) for i in range(20)], max_batch=32)).contiguous()
    q = enc.encode_many([torch.from_numpy(S.synth_clip(n, seed=7003)[0]).to(dev)])[0].contiguous()
    qn, _ = ops.row_norms(q); pn, _ = ops.row_norms(pool)
    nn, _ = ops.knn_topk(q, pool, 32)
    idx = nn[:, :4].contiguous()
    ref = ops.concat_reselect(idx, q, qn, pool, pn, concat_weight=0.2)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    a = torch.randn(1 << 20, device=dev); b = torch.randn(1 << 20, device=dev)
    big = torch.randn(480000, 32, device=dev); slot = ops.new_slot(dev)
    x64 = torch.randn(240000, 64, device=dev)
    w64 = ops.attach_split(ops.pack_conv_weight(torch.randn(64, 64, 3) / 14).to(dev)); o64 = torch.empty_like(x64)
    modes = {"quiet": lambda: None,
             "elementwise": lambda: [torch.add(a, b, out=a) for _ in range(400)],
             "absmax": lambda: [ops.absmax(big, slot) for _ in range(200)],
             "small-tile conv": lambda: [ops.conv_gemm(x64[:3000], w64, o64[:3000], m=3000, n=64, cin=64, taps=3, pad=1, t_in=3000) for _ in range(300)],
             "win conv": lambda: [ops.conv_gemm(x64, w64, o64, m=240000, n=64, cin=64, taps=3, pad=1, t_in=240000) for _ in range(60)]}
    for name, fn in modes.items():
        bad = 0
        for rep in range(8):
            with torch.cuda.stream(side):
                fn()
            o = ops.concat_reselect(idx, q, qn, pool, pn, concat_weight=0.2)
            torch.cuda.synchronize()
            bad += int(not torch.equal(o, ref))
        print(f"{name:16s}: {bad} of 8 runs differ from the quiet reference")
