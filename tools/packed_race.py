"""the kernels that hand-emit packed fp32 (v_pk_fma/mul/add_f32: kn_gelu2 in csrc/common.h, the channel
pairs of conv0_ln_gelu_kernel, the GELU epilogues of the f16x2 GEMMs) bit-compared beside MFMA-issuing co-runners — the
situation tools/concat_race.py showed to matter for round 3's SLP-vectorised re-selection walk.  Each kernel is launched RUNS
times on fixed inputs while a second stream keeps the generator's C = 256 / C = 128 windowed convolutions on the chip; every
output is compared with a quiet launch.
    python tools/packed_race.py [RUNS]
Prints one JSON line: {kernel: launches that differ}."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from knn_svc_amd import _lib, ops


def run(runs=30):
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(7)
    rn = lambda *s: torch.randn(*s, generator=g)
    # (1) conv0 + LayerNorm + GELU (csrc/elementwise.hip: channel pairs on packed fp32), one 30 s chunk: 96 063 rows x 512
    wav = (rn(1, 480320) * 0.1).to(dev)
    w0, g0, b0 = (rn(512, 10) / 3).to(dev), (1 + 0.1 * rn(512)).to(dev), (0.1 * rn(512)).to(dev)
    # (2) FFN1-shaped GEMM with the GELU epilogue on the 256 x 256 kernel (1500 x 1024 -> 4096), (3) a GELU epilogue on the
    # 128 x 128 kernel (K = 512: below the quad kernel's K >= 1024 rule; three workgroups per CU: they DO share CUs with co-runners)
    x1 = ops.split_pack(rn(1500, 1024)).to(dev); w1 = ops.attach_split((rn(4096, 1024) / 32).to(dev)); b1 = (0.1 * rn(4096)).to(dev)
    x2 = rn(6000, 512).to(dev); w2 = ops.attach_split((rn(512, 512) / 22).to(dev)); b2 = (0.1 * rn(512)).to(dev)
    cases = {
        "conv0_ln_gelu": lambda: ops.wavlm_conv0(wav, w0, g0, b0, 10, 5),
        "gemm_gelu_quad": lambda: ops.linear(x1, w1, b1, act=ops.ACT_GELU, x_split=True),      # pre-split A -> conv_gemm2quad_kernel, EPI = 2
        "gemm_gelu_128": lambda: ops.linear(x2, w2, b2, act=ops.ACT_GELU),
    }
    names = {}
    quiet = {}
    for k, fn in cases.items():
        quiet[k] = fn().clone()
        names[k] = ops.last_conv_kernel() if k != "conv0_ln_gelu" else "conv0_ln_gelu_kernel"
        torch.cuda.synchronize()
        assert torch.equal(fn(), quiet[k]), f"{k}: quiet repeats differ"
    side = torch.cuda.Stream()
    x256 = torch.randn(15000, 256, device=dev); o256 = torch.empty_like(x256)
    w256 = ops.attach_split(ops.pack_conv_weight(torch.randn(256, 256, 3) / 28).to(dev))
    x128 = torch.randn(120000, 128, device=dev); o128 = torch.empty_like(x128)
    w128 = ops.attach_split(ops.pack_conv_weight(torch.randn(128, 128, 3) / 20).to(dev))
    co = set()
    differ = {k: 0 for k in cases}
    for rep in range(runs):
        with torch.cuda.stream(side):
            for _ in range(12):
                ops.conv_gemm(x256, w256, o256, m=15000, n=256, cin=256, taps=3, pad=1, t_in=15000); co.add(ops.last_conv_kernel())
                ops.conv_gemm(x128, w128, o128, m=120000, n=128, cin=128, taps=3, pad=1, t_in=120000); co.add(ops.last_conv_kernel())
        outs = {k: fn() for k, fn in cases.items()}
        torch.cuda.synchronize()
        for k, o in outs.items():
            differ[k] += 0 if torch.equal(o, quiet[k]) else 1
    return dict(lib=os.path.basename(_lib.LIB_PATH), runs=runs, differ=differ, kernels=names, co_runners=sorted(co))


if __name__ == "__main__":
    print(json.dumps(run(int(sys.argv[1]) if len(sys.argv) > 1 else 30)))
