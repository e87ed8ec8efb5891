"""GPU: each HIP kernel against the CPU oracle / a plain torch fp32 reference of the same op,
called through the C ABI (knn_svc_amd.ops -> libknnsvc_hip.so)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from knn_svc_amd import config as C, synthetic as S

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _ops():
    from knn_svc_amd import ops
    return ops


def _knob(monkeypatch, name, value):
    """An A/B switch of the C++ dispatcher: the library reads those once, so set the variable AND have it re-read them."""
    monkeypatch.setenv(name, value)
    _ops().reload_knobs()


@pytest.fixture(autouse=True)
def _knobs_back_to_default(monkeypatch):
    yield
    monkeypatch.undo()
    _ops().reload_knobs()


def _t(a):
    return torch.from_numpy(np.asarray(a))


def _err(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).abs().max()), float((a - b).abs().max() / (b.abs().max() + 1e-30))


# ------------------------------------------------------------------ implicit GEMM conv
@pytest.mark.parametrize("M,K,N", [(65, 64, 40), (300, 1024, 1024), (1500, 1024, 3072), (129, 36, 130), (7, 4, 3)])
def test_linear(M, K, N):
    ops = _ops()
    g = torch.Generator().manual_seed(M + K + N)
    x = torch.randn(M, K, generator=g); w = torch.randn(N, K, generator=g) / K ** 0.5; b = torch.randn(N, generator=g)
    r = torch.randn(M, N, generator=g)
    ref = F.gelu(F.linear(x, w, b)) + r
    out = ops.linear(x.to(DEV), w.to(DEV), b.to(DEV), act=ops.ACT_GELU, resid=r.to(DEV))
    assert _err(out, ref)[0] < 2e-5


@pytest.mark.parametrize("mode,attr", [("f16x2", "_w2"), ("bf16x3", "_w3")])
@pytest.mark.parametrize("M,K,N", [(300, 1024, 1024), (1500, 4096, 1024), (129, 64, 130), (500, 96, 40), (77, 32, 20)])
def test_linear_emulated_fp32(M, K, N, mode, attr, monkeypatch):
    """fp32 emulated on the fp16 (3 MFMAs / product) or bf16 (6) matrix cores: must be as accurate as the
    fp32-MFMA path (vs fp64)."""
    ops = _ops()
    monkeypatch.setenv("KNNSVC_GEMM", mode)
    g = torch.Generator().manual_seed(M + K + N)
    x = torch.randn(M, K, generator=g) * 3; w = torch.randn(N, K, generator=g) / K ** 0.5; b = torch.randn(N, generator=g)
    ref = x.double() @ w.double().T + b.double()
    wd = w.to(DEV)
    o32 = ops.linear(x.to(DEV), wd, b.to(DEV))
    ops.attach_split(wd)
    assert hasattr(wd, attr)
    o3 = ops.linear(x.to(DEV), wd, b.to(DEV))
    e32, e3 = float((o32.cpu().double() - ref).abs().max()), float((o3.cpu().double() - ref).abs().max())
    print(f"fp32 MFMA err {e32:.2e}, {mode} err {e3:.2e}")
    assert e3 <= 2.0 * e32 + 1e-6 and e3 < 1e-5 * float(ref.abs().max())


def test_linear_split_layout_in_and_out(monkeypatch):
    """A2: a GEMM fed with pre-split activations gives bit-identical results to splitting them in the kernel, and a
    GEMM writing the split layout round-trips to its fp32 output within the split error (2^-22 relative)."""
    ops = _ops()
    g = torch.Generator().manual_seed(21)
    M, K, N = 700, 1024, 384
    x = (torch.randn(M, K, generator=g) * 2).to(DEV); w = (torch.randn(N, K, generator=g) / K ** 0.5).to(DEV)
    b = torch.randn(N, generator=g).to(DEV)
    wd = ops.attach_split(w)
    if not hasattr(wd, "_w2"):
        pytest.skip("KNNSVC_GEMM is not f16x2")
    ref = ops.linear(x, wd, b, act=ops.ACT_GELU)
    # (by default a pre-split operand with K >= 1024 takes the 256x256 kernel, whatever M: another K grouping — fp32 noise apart)
    got_q = ops.linear(ops.split_pack(x), wd, b, act=ops.ACT_GELU, x_split=True)
    assert ops.last_conv_kernel() == "Q256S" and float((got_q - ref).abs().max()) < 2e-5
    _knob(monkeypatch, "KNNSVC_QUAD", "0")          # the same 128x128 kernel for both: staging is the only difference
    got = ops.linear(ops.split_pack(x), wd, b, act=ops.ACT_GELU, x_split=True)
    assert torch.equal(ref, got)
    packed = ops.linear(x, wd, b, act=ops.ACT_GELU, out_split=True)
    back = ops.split_unpack(packed)
    assert float((back - ref).abs().max()) <= 4e-7 * float(ref.abs().max())
    # same round-to-nearest split as the packer (compared as values: the sign of a zero half may differ)
    assert torch.equal(packed.view(torch.float16).float(), ops.split_pack(ref).view(torch.float16).float())
    from knn_svc_amd._lib import KnnSvcError
    with pytest.raises(KnnSvcError):
        ops.linear(x, wd, b, resid=ref, out_split=True)


@pytest.mark.parametrize("M,K,N", [(2048 + 77, 1024, 1024), (700, 2048, 520), (256, 32, 256), (31, 4096, 1028), (300, 96, 260)])
def test_linear_quad_kernel(M, K, N, monkeypatch):
    """conv_gemm2quad_kernel<Gemm2QuadS> (256x256 block, 128x128 wave tiles of v_mfma_f32_16x16x32_f16, hand-pipelined register
    staging, LDS-transposed 16-byte epilogue): ragged M and N tails, 1 / 32 / 64 / 128 slabs (odd and even counts), every epilogue
    operand — bias, GELU, residual, accumulate, divide, range slot, split-layout output — against fp64 and against the 128x128
    kernel."""
    ops = _ops()
    q16 = "1"
    qname = "Q256S"
    g = torch.Generator().manual_seed(17)
    x = torch.randn(M, K, generator=g); w = torch.randn(N, K, generator=g) / K ** 0.5; b = torch.randn(N, generator=g)
    r = torch.randn(M, N, generator=g)
    wd = ops.attach_split(w.to(DEV))
    if not hasattr(wd, "_w2"):
        pytest.skip("KNNSVC_GEMM is not f16x2")
    xs = ops.split_pack(x.to(DEV))
    ref = x.double() @ w.double().T + b.double()
    outs = {}
    for mode in ("2", "0"):
        _knob(monkeypatch, "KNNSVC_QUAD", mode)
        o = ops.linear(xs, wd, b.to(DEV), x_split=True)
        assert ops.last_conv_kernel() == (qname if mode == "2" else ("F128a2" if M * N >= 256 * 128 * 128 else ops.last_conv_kernel()))
        outs[mode] = o.cpu()
        e = float((o.cpu().double() - ref).abs().max())
        assert e < 3e-5 * max(1.0, (K / 1024) ** 0.5), (mode, e)
        # GELU + range slot
        slot = ops.new_slot(DEV)
        og = ops.linear(xs, wd, b.to(DEV), act=ops.ACT_GELU, x_split=True, out_absmax=slot)
        rg = torch.nn.functional.gelu(ref)
        assert float((og.cpu().double() - rg).abs().max()) < 3e-5 * max(1.0, (K / 1024) ** 0.5)
        assert float(og.abs().max()) <= float(slot.max()) <= float(og.abs().max()) * 1.0001 + float(torch.nn.functional.gelu(b).abs().max())
        # residual, then accumulate + divide on top of an existing output
        orr = ops.linear(xs, wd, b.to(DEV), resid=r.to(DEV), x_split=True)
        assert float((orr.cpu().double() - (ref + r.double())).abs().max()) < 4e-5 * max(1.0, (K / 1024) ** 0.5)
        acc = r.to(DEV).clone()
        ops.conv_gemm(xs, wd, acc, m=M, n=N, cin=K, bias=b.to(DEV), accumulate=True, div=3.0, x_split=True)
        assert float((acc.cpu().double() - (ref + r.double()) / 3.0).abs().max()) < 2e-5 * max(1.0, (K / 1024) ** 0.5)
        if N % 32 == 0:     # split-layout output (whole, and from column 32 on)
            osp = ops.linear(xs, wd, b.to(DEV), x_split=True, out_split=True)
            assert float((ops.split_unpack(osp).cpu().double() - ref).abs().max()) < 3e-5 * max(1.0, (K / 1024) ** 0.5)
            oh = ops.linear(xs, wd, b.to(DEV), x_split=True, out_split=32).cpu()
            assert float((oh[:, :32].double() - ref[:, :32]).abs().max()) < 3e-5
            assert float((ops.split_unpack(oh[:, 32:].contiguous().to(DEV)).cpu().double() - ref[:, 32:]).abs().max()) < 3e-5 * max(1.0, (K / 1024) ** 0.5)
    # same products, same order along K inside a tile row: the two kernels agree to fp32 accumulation noise
    assert float((outs["2"] - outs["0"]).abs().max()) < 2e-5 * max(1.0, (K / 1024) ** 0.5)
    # the specialised epilogues (conv_epilogue_wide_fast: bias, GELU, residual, split columns) against the generic one: same
    # arithmetic in the same order -> the same bits
    if q16 == "1":
        _knob(monkeypatch, "KNNSVC_QUAD", "2")
        res = {}
        for epi in ("1", "0"):
            _knob(monkeypatch, "KNNSVC_QUAD_EPI", epi)
            r_ = [ops.linear(xs, wd, b.to(DEV), x_split=True), ops.linear(xs, wd, None, x_split=True),
                  ops.linear(xs, wd, b.to(DEV), act=ops.ACT_GELU, x_split=True),
                  ops.linear(xs, wd, b.to(DEV), resid=r.to(DEV), x_split=True)]
            if N % 128 == 0:
                r_ += [ops.linear(xs, wd, b.to(DEV), act=ops.ACT_GELU, x_split=True, out_split=True),
                       ops.linear(xs, wd, b.to(DEV), x_split=True, out_split=128 if N > 128 else True)]
            assert ops.last_conv_kernel() == "Q256S"
            res[epi] = [t.cpu() for t in r_]
        assert float((res["1"][2].double() - torch.nn.functional.gelu(ref)).abs().max()) < 3e-5 * max(1.0, (K / 1024) ** 0.5)
        for k_, (t1, t0) in enumerate(zip(res["1"], res["0"])):
            bad = (t1.view(torch.int32) != t0.view(torch.int32))
            assert not bool(bad.any()), (k_, int(bad.sum()), bad.nonzero()[:4].tolist(), float((t1 - t0).abs().max()))


def test_conv_quad_kernel_taps_stride_batches(monkeypatch):
    """The quad kernel on a strided 3-tap convolution over a batch of sequences (WavLM's conv stack shape: A2 input,
    stride 2, per-batch strides) and with conv padding rows, against the 128x128 kernel."""
    ops = _ops()
    g = torch.Generator().manual_seed(19)
    B, T, Cin, Cout, k, st = 3, 2001, 256, 512, 3, 2
    x = torch.randn(B * T, Cin, generator=g)
    w = ops.attach_split(ops.pack_conv_weight(torch.randn(Cout, Cin, k, generator=g) / (Cin * k) ** 0.5).to(DEV))
    xs = ops.split_pack(x.to(DEV))
    t_out = (T - k) // st + 1
    res = {}
    for mode in ("2", "0"):
        _knob(monkeypatch, "KNNSVC_QUAD", mode)
        y = torch.empty(B * t_out, Cout, device=DEV)
        ops.conv_gemm(xs, w, y, m=t_out, n=Cout, cin=Cin, taps=k, stride=st, t_in=T, batches=B, x_bstride=T * Cin, o_bstride=t_out * Cout, x_split=True)
        yp = torch.empty(B * T, Cout, device=DEV)
        ops.conv_gemm(xs, w, yp, m=T, n=Cout, cin=Cin, taps=k, stride=1, pad=1, t_in=T, batches=B, x_bstride=T * Cin, o_bstride=T * Cout, x_split=True)
        res[mode] = (y.cpu(), yp.cpu(), ops.last_conv_kernel())
    assert res["2"][2] == "Q256S"
    ref = torch.nn.functional.conv1d(x.view(B, T, Cin).transpose(1, 2).double(), w.cpu().view(Cout, k, Cin).permute(0, 2, 1).double(), stride=st)
    assert float((res["2"][0].view(B, t_out, Cout).transpose(1, 2).double() - ref).abs().max()) < 2e-5
    assert float((res["2"][0] - res["0"][0]).abs().max()) < 2e-5 and float((res["2"][1] - res["0"][1]).abs().max()) < 2e-5


@pytest.mark.parametrize("xs,ws,a_scale", [(1e-3, 1e-4, 4096.0), (30.0, 5.0, 0.0), (0.05, 40.0, 0.0)])
def test_linear_f16x2_scales(xs, ws, a_scale, monkeypatch):
    """The power-of-two operand scaling keeps small activations / odd weight magnitudes at fp32 accuracy,
    and an activation beyond the fp16 range turns the output NaN instead of silently wrong."""
    ops = _ops()
    monkeypatch.setenv("KNNSVC_GEMM", "f16x2")
    g = torch.Generator().manual_seed(7)
    M, K, N = 200, 512, 96
    x = torch.randn(M, K, generator=g) * xs; w = torch.randn(N, K, generator=g) * ws
    ref = x.double() @ w.double().T
    wd = ops.attach_split(w.to(DEV))
    assert 2 ** 13 <= float(w.abs().max()) * wd._w2_scale < 2 ** 14
    o = torch.empty(M, N, device=DEV)
    ops.conv_gemm(x.to(DEV), wd, o, m=M, n=N, cin=K, a_scale=a_scale)
    err = float((o.cpu().double() - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt())
    print(f"x scale {xs}, w scale {ws}: rms rel err {err:.2e}")
    assert err < 4e-7
    if a_scale:          # the default activation scale (16) is past its accuracy floor at rms 1e-3: degraded, not wrong
        ops.conv_gemm(x.to(DEV), wd, o, m=M, n=N, cin=K)
        err = float((o.cpu().double() - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt())
        assert err < 3e-6
    x[3, 5] = 70000.0 / (a_scale or 16.0)
    ops.conv_gemm(x.to(DEV), wd, o, m=M, n=N, cin=K, a_scale=a_scale)
    assert not bool(torch.isfinite(o[3]).all()) and bool(torch.isfinite(o[4]).all())


@pytest.mark.parametrize("cin,cout,k,s,d,T", [(1, 64, 10, 5, 1, 1000), (64, 64, 3, 2, 1, 199), (32, 32, 11, 1, 5, 700),
                                             (512, 512, 3, 2, 1, 301), (8, 16, 4, 2, 1, 640), (34, 34, 16, 8, 1, 800),
                                             (32, 1, 7, 1, 1, 500)])
def test_conv1d(cin, cout, k, s, d, T):
    ops = _ops()
    g = torch.Generator().manual_seed(cin * 7 + k)
    B = 2
    x = torch.randn(B, cin, T, generator=g)
    w = torch.randn(cout, cin, k, generator=g) / (cin * k) ** 0.5
    b = torch.randn(cout, generator=g)
    pad = (k * d - d) // 2 if s == 1 else k // 2
    ref = F.conv1d(F.leaky_relu(x, 0.1), w, b, stride=s, padding=pad, dilation=d)
    To = ref.shape[-1]
    xcl = x.transpose(1, 2).contiguous().to(DEV)
    out = torch.empty(B, To, cout, device=DEV)
    ops.conv_gemm(xcl, ops.pack_conv_weight(w).to(DEV), out, m=To, n=cout, cin=cin, taps=k, stride=s, dil=d, pad=pad,
                  t_in=T, bias=b.to(DEV), a_slope=0.1, batches=B, x_bstride=T * cin, o_bstride=To * cout)
    assert _err(out.transpose(1, 2), ref)[0] < 2e-5
    if cin % 32 == 0:                      # same conv through the bf16x3 kernel
        wp = ops.attach_split(ops.pack_conv_weight(w).to(DEV))
        out2 = torch.empty(B, To, cout, device=DEV)
        ops.conv_gemm(xcl, wp, out2, m=To, n=cout, cin=cin, taps=k, stride=s, dil=d, pad=pad, t_in=T, bias=b.to(DEV),
                      a_slope=0.1, batches=B, x_bstride=T * cin, o_bstride=To * cout)
        assert _err(out2.transpose(1, 2), ref)[0] < 2e-5


def test_conv_epilogue_chain():
    """residual + accumulate + divide, output into a wider (concat) buffer."""
    ops = _ops()
    g = torch.Generator().manual_seed(5)
    T, ch = 333, 32
    x = torch.randn(1, ch, T, generator=g); w = torch.randn(ch, ch, 7, generator=g) / 15; b = torch.randn(ch, generator=g)
    prev = torch.randn(T, ch, generator=g)
    ref = ((F.conv1d(x, w, b, padding=3)[0].T + x[0].T) + prev) / 3
    wide = torch.zeros(T, 2 * ch)
    wide[:, ch:] = prev
    wide = wide.to(DEV)
    xcl = x[0].T.contiguous().to(DEV)
    ops.conv_gemm(xcl, ops.pack_conv_weight(w).to(DEV), wide[:, ch:], m=T, n=ch, cin=ch, taps=7, pad=3, t_in=T,
                  bias=b.to(DEV), resid=xcl, ldr=ch, ldo=2 * ch, accumulate=True, div=3.0)
    assert _err(wide[:, ch:], ref)[0] < 1e-5
    assert float(wide[:, :ch].abs().max()) == 0.0


@pytest.mark.parametrize("cin,cout,k,u,T", [(64, 32, 20, 10, 57), (16, 8, 4, 2, 301), (512, 256, 16, 8, 40)])
def test_conv_transpose(cin, cout, k, u, T):
    ops = _ops()
    g = torch.Generator().manual_seed(k)
    x = torch.randn(1, cin, T, generator=g)
    w = torch.randn(cin, cout, k, generator=g) / (cin * 2) ** 0.5
    b = torch.randn(cout, generator=g)
    pad = (k - u) // 2
    ref = F.conv_transpose1d(F.leaky_relu(x, 0.1), w, b, stride=u, padding=pad)[0].T
    To = ref.shape[0]
    assert To == T * u
    xcl = x[0].T.contiguous().to(DEV)
    out = torch.zeros(To, cout, device=DEV)
    R = k // u
    ops.conv_gemm(xcl, ops.attach_split(ops.pack_convT_weight(w, u).to(DEV)), out, m=T + R - 1, n=u * cout, cin=cin, taps=R, stride=1,
                  dil=-1, pad=0, t_in=T, bias=b.to(DEV), bias_period=cout, a_slope=0.1, ldo=cout,
                  convt_u=u, convt_cout=cout, convt_pad=pad, t_out=To)
    assert _err(out, ref)[0] < 2e-5


@pytest.mark.parametrize("case", ["win_k3_resid_slot", "win_k11_dil5_wide_buffer", "win_k7_n40_ragged", "tapmajor_stride2", "transposed_u8",
                                  "transposed_u2_concat"])
def test_patch_epilogue_equals_column_per_lane(case, monkeypatch):
    """The LDS-patch epilogue with 16-byte stores (conv_epilogue_wide32: windowed and tap-major f16x2 kernels, plain and
    transposed) writes the bits of the column-per-lane / generic epilogues it replaces — output AND range slot — on ragged row
    counts, a residual, a wider (concat) output buffer, a bias period, column counts that are no multiple of 32."""
    ops = _ops()
    g = torch.Generator().manual_seed(len(case))

    def run():
        slot = ops.new_slot(DEV)
        if case.startswith("transposed"):
            cin, cout, k, u, T = (256, 128, 16, 8, 333) if case == "transposed_u8" else (64, 32, 4, 2, 1001)
            x = torch.randn(T, cin, generator=torch.Generator().manual_seed(3)).to(DEV)
            w = torch.randn(cin, cout, k, generator=torch.Generator().manual_seed(4)) / (cin * 2) ** 0.5
            b = torch.randn(cout, generator=torch.Generator().manual_seed(5)).to(DEV)
            ld = 2 * cout if case.endswith("concat") else cout
            out = torch.zeros(T * u, ld, device=DEV)
            R = k // u
            ops.conv_gemm(x, ops.attach_split(ops.pack_convT_weight(w, u).to(DEV)), out, m=T + R - 1, n=u * cout, cin=cin, taps=R, stride=1,
                          dil=-1, pad=0, t_in=T, bias=b, bias_period=cout, a_slope=0.1, ldo=ld, convt_u=u, convt_cout=cout,
                          convt_pad=(k - u) // 2, t_out=T * u, out_absmax=slot)
            return out, slot, ops.last_conv_epilogue()
        cin, cout, k, s_, d, T = {"win_k3_resid_slot": (128, 128, 3, 1, 1, 1217), "win_k11_dil5_wide_buffer": (64, 64, 11, 1, 5, 777),
                                  "win_k7_n40_ragged": (32, 40, 7, 1, 3, 501), "tapmajor_stride2": (64, 96, 3, 2, 1, 999)}[case]
        x = torch.randn(T, cin, generator=torch.Generator().manual_seed(6)).to(DEV)
        w = torch.randn(cout, cin, k, generator=torch.Generator().manual_seed(7)) / (cin * k) ** 0.5
        b = torch.randn(cout, generator=torch.Generator().manual_seed(8)).to(DEV)
        pad = (k * d - d) // 2 if s_ == 1 else k // 2
        To = (T + 2 * pad - d * (k - 1) - 1) // s_ + 1
        ld = 2 * cout if "wide_buffer" in case else cout
        out = torch.zeros(To, ld, device=DEV)
        kw = {}
        if "resid" in case:
            kw = dict(resid=x, ldr=cin)
        ops.conv_gemm(x, ops.attach_split(ops.pack_conv_weight(w).to(DEV)), out, m=To, n=cout, cin=cin, taps=k, stride=s_, dil=d, pad=pad,
                      t_in=T, bias=b, a_slope=0.1, ldo=ld, out_absmax=slot, **kw)
        return out, slot, ops.last_conv_epilogue()

    _knob(monkeypatch, "KNNSVC_WIN_WIDE", "0")
    ref, ref_slot, ref_epi = run()
    _knob(monkeypatch, "KNNSVC_WIN_WIDE", "1")
    got, got_slot, got_epi = run()
    torch.cuda.synchronize()
    assert (ref_epi, got_epi) == ("lane", "patch")            # both forms really ran
    assert torch.equal(got, ref)
    assert float(ref.abs().max()) > 0.1
    if ref_slot is not None:
        assert torch.equal(got_slot, ref_slot) and float(ref_slot.max()) > 0.0



def test_grouped_pos_conv():
    ops = _ops()
    g = torch.Generator().manual_seed(9)
    E, G, K, T, B = 128, 16, 128, 150, 2
    x = torch.randn(B, T, E, generator=g)
    w = torch.randn(E, E // G, K, generator=g) / (E // G * K) ** 0.5
    b = torch.randn(E, generator=g)
    pc = F.conv1d(x.transpose(1, 2), w, b, padding=K // 2, groups=G)[:, :, :-1]
    ref = x + F.gelu(pc).transpose(1, 2)
    xd = x.to(DEV)
    out = torch.empty_like(xd)
    cg = E // G
    ops.conv_gemm(xd, ops.attach_split(ops.pack_grouped_conv_weight(w, G).to(DEV)), out, m=T, n=cg, cin=cg, taps=K, pad=K // 2, t_in=T,
                  ldx=E, ldo=E, bias=b.to(DEV), act=ops.ACT_GELU, resid=xd, ldr=E, batches=B, groups=G,
                  x_bstride=T * E, x_gstride=cg, w_gstride=cg * cg * K, bias_gstride=cg, o_bstride=T * E, o_gstride=cg,
                  r_bstride=T * E, r_gstride=cg)
    assert _err(out, ref)[0] < 2e-5


# ------------------------------------------------------------------ WavLM pieces
def test_layernorm_gelu():
    ops = _ops()
    g = torch.Generator().manual_seed(1)
    for dim in (64, 512, 1024):
        x = torch.randn(301, dim, generator=g) * 3 + 1
        ga, be = torch.randn(dim, generator=g), torch.randn(dim, generator=g)
        ref = F.gelu(F.layer_norm(x, (dim,), ga, be, 1e-5))
        out = ops.layernorm(x.to(DEV), ga.to(DEV), be.to(DEV), gelu=True)
        assert _err(out, ref)[0] < 1e-5


def test_attention_and_gate():
    from oracle import wavlm_ref
    ops = _ops()
    cfg = dict(C.WAVLM_LARGE, encoder_layers=1)
    H, E, T, B = 16, 1024, 333, 2
    sd = S.seeded_state([s for s in S.wavlm_param_spec(cfg) if s[0].startswith("encoder.layers.0.self_attn")], 3)
    g = torch.Generator().manual_seed(2)
    xn = torch.randn(T, B, E, generator=g)
    p = "encoder.layers.0.self_attn."
    gate_ref = wavlm_ref.gate(sd, cfg, 0, xn)                                   # [B,H,T,1]
    w8, b8 = sd[p + "grep_linear.weight"], sd[p + "grep_linear.bias"]
    w2 = torch.stack([w8[:4].sum(0), w8[4:].sum(0)]).contiguous()
    b2 = torch.stack([b8[:4].sum(), b8[4:].sum()])
    xbt = xn.transpose(0, 1).reshape(B * T, E).contiguous().to(DEV)
    gate = ops.wavlm_gate(xbt, H, w2.to(DEV), b2.to(DEV), sd[p + "grep_a"].reshape(-1).to(DEV))
    assert _err(gate.reshape(B, T, H).permute(0, 2, 1), gate_ref[..., 0])[0] < 1e-5
    # attention with that gate
    pb = wavlm_ref.position_bias(sd, cfg, T)                                   # [H,T,T]
    q = F.linear(xn, sd[p + "q_proj.weight"], sd[p + "q_proj.bias"])
    k = F.linear(xn, sd[p + "k_proj.weight"], sd[p + "k_proj.bias"])
    v = F.linear(xn, sd[p + "v_proj.weight"], sd[p + "v_proj.bias"])
    sh = lambda t: t.reshape(T, B * H, 64).transpose(0, 1).reshape(B, H, T, 64)
    ref = F.scaled_dot_product_attention(sh(q), sh(k), sh(v), attn_mask=gate_ref * pb[None])
    ref = ref.permute(0, 2, 1, 3).reshape(B * T, E)
    lut = wavlm_ref.rel_bucket_table(T, 320, 800)
    table = sd[p + "relative_attention_bias.weight"][lut].T.contiguous()        # [H, 2T-1]
    qkv = torch.cat([q, k, v], -1).transpose(0, 1).reshape(B * T, 3 * E).contiguous().to(DEV)
    out = ops.wavlm_attention(qkv, gate, table.to(DEV), B, T, H)
    assert _err(out, ref)[0] < 2e-5


@pytest.mark.parametrize("B,T,split", [(2, 333, False), (3, 700, True), (1, 1500, True)])
def test_attention_64_queries_per_wave_equals_32(B, T, split, monkeypatch):
    """attention2q_kernel<2> (64 queries per wave: the large-batch dispatch) against attention2_kernel (32 per wave): the
    same operations per query in the same order — bit-identical, with ragged last blocks, key-padding lengths, pre-split K / V
    and split output."""
    ops = _ops()
    H, E = 16, 1024
    g = torch.Generator().manual_seed(B * 1000 + T)
    qkv = (torch.randn(B * T, 3 * E, generator=g) * 0.5).to(DEV)
    gate = torch.rand(B * T, H, generator=g).to(DEV)
    table = (torch.randn(H, 2 * T - 1, generator=g) * 0.1).to(DEV)
    lens = torch.tensor([T - 37 * b for b in range(B)], dtype=torch.int32, device=DEV)
    if split:
        qkv = qkv.clone(); qkv[:, E:] = ops.split_pack(qkv[:, E:].contiguous())
    res = {}
    for qb in ("1", "2"):
        monkeypatch.setenv("KNNSVC_ATT_QB", qb)
        res[qb] = [ops.wavlm_attention(qkv, gate, table, B, T, H, kv_split=split, out_split=split, kv_len=kl).clone() for kl in (None, lens)]
    for a, b in zip(res["1"], res["2"]):
        assert torch.equal(a.view(torch.int32), b.view(torch.int32))


@pytest.mark.parametrize("B,T,split", [(2, 333, False), (3, 700, True), (1, 1500, True), (2, 1061, True)])
def test_attention_eight_waves_per_workgroup_equals_four(B, T, split, monkeypatch):
    """attention2w_kernel (512 queries per workgroup, K / V tiles in three LDS buffers staged two ahead, one barrier per tile at a
    different point of the step for the two waves of a SIMD — the large-batch dispatch) against attention2q_kernel: per query the
    same operations in the same order, so bit-identical — with an odd number of 32-key steps (T = 333, 1061: the last tile has one
    step), fewer than three tiles after key padding, ragged last query blocks, pre-split K / V and split output."""
    ops = _ops()
    H, E = 16, 1024
    g = torch.Generator().manual_seed(B * 1000 + T + 1)
    qkv = (torch.randn(B * T, 3 * E, generator=g) * 0.5).to(DEV)
    gate = torch.rand(B * T, H, generator=g).to(DEV)
    table = (torch.randn(H, 2 * T - 1, generator=g) * 0.1).to(DEV)
    lens = torch.tensor([max(T - 237 * b, 70) for b in range(B)], dtype=torch.int32, device=DEV)
    short = torch.tensor([70 + 31 * b for b in range(B)], dtype=torch.int32, device=DEV)       # two or three tiles only
    one = torch.tensor([(5, 64, 33)[b % 3] for b in range(B)], dtype=torch.int32, device=DEV)    # a single tile (one or two 32-key steps)
    if split:
        qkv = qkv.clone(); qkv[:, E:] = ops.split_pack(qkv[:, E:].contiguous())
    monkeypatch.setenv("KNNSVC_ATT_QB", "2")
    res = {}
    for nw in ("4", "8"):
        monkeypatch.setenv("KNNSVC_ATT_NW", nw)
        res[nw] = [ops.wavlm_attention(qkv, gate, table, B, T, H, kv_split=split, out_split=split, kv_len=kl).clone()
                   for kl in (None, lens, short, one)]
    for a, b in zip(res["4"], res["8"]):
        assert torch.equal(a.view(torch.int32), b.view(torch.int32))


# ------------------------------------------------------------------ kNN
def test_knn_golden(golden):
    from oracle import knn_ref
    ops = _ops()
    g = golden("g3_knn")
    q = S.clustered_features(int(g["nq"]), 1024, int(g["q_seed"]))
    p = S.clustered_features(int(g["np_"]), 1024, int(g["p_seed"]))
    idx, dist = ops.knn_topk(q.to(DEV), p.to(DEV), 32)
    idx, dist = idx.cpu(), dist.cpu()
    ref_idx = _t(g["idx"]).long()
    st = knn_ref.topk_agreement(ref_idx, idx, knn_ref.cosine_dist_f64(q, p), tau=5e-7)
    print("kNN agreement vs reference fixture:", st)
    # index parity is defined up to the reference's own fp32 rounding gaps (SURVEY §7 hard part 1):
    # every disagreement must sit inside a gap <= tau of the exact distance
    assert st["unexplained"] == 0, st
    # ratcheted to the measured values (r02: top-4 100 %, ordered top-32 99.5 %, sets 100 %, largest inversion 1.7e-7):
    # the first four neighbours — the ones the path uses — and the top-32 set are exactly the reference's
    assert st["top4"] == 1.0 and st["sets"] == 1.0 and st["allk"] >= 0.99 and st["max_gap"] <= 2.5e-7, st
    assert float((dist - _t(g["dist"])).abs().max()) < 5e-6   # a few ulp of |q|^2+|p|^2 over |q||p|
    # internal consistency: sorted ascending, indices in range, no duplicates
    assert bool((dist[:, 1:] >= dist[:, :-1]).all())
    assert int(idx.min()) >= 0 and int(idx.max()) < len(p)
    assert all(len(set(r.tolist())) == 32 for r in idx)


def test_knn_golden_with_exact_ties(golden):
    """Fixture G3b (reference-generated, tests/gen_golden.py: knn_ties): duplicated pool rows, a block of 400 bit-identical
    rows, queries that are pool rows / the identical row.  torch.topk returns tied rows in no particular order (the fixture
    records what it did: only 62 % of its rows list ties by ascending index); this build's rule is lower index first on every
    route and device count.  Checked: the distances are the reference's position by position; wherever the reference's list has
    no tie the indices are the reference's; every returned index has the distance the reference formula gives it; nothing
    closer was left out; ties come out lower index first; and the f0 re-rank (a STABLE sort over the 32, so it inherits the
    tie order of its input) reproduces the reference's output on the reference's own lists."""
    from oracle import knn_ref
    ops = _ops()
    g = golden("g3b_knn_ties")
    q, p = _knn_ties_inputs()
    ref_i, ref_d = _t(g["idx"]).long(), _t(g["dist"])
    full = torch.cat([knn_ref.cosine_dist(q[s0:s0 + 20], p) for s0 in range(0, len(q), 20)])      # the reference formula, 20 rows at a time
    for fused in ("0", "1"):
        os.environ["KNNSVC_KNN_FUSED_MIN_Q"] = "1"
        os.environ["KNNSVC_KNN_FUSED"] = fused
        try:
            idx, dist = ops.knn_topk(q.to(DEV), p.to(DEV), 32)
        finally:
            os.environ.pop("KNNSVC_KNN_FUSED", None); os.environ.pop("KNNSVC_KNN_FUSED_MIN_Q", None)
        idx, dist = idx.cpu(), dist.cpu()
        assert float((dist - ref_d).abs().max()) < 5e-6
        got_ref_d = full.gather(1, idx)
        assert float((got_ref_d - dist).abs().max()) < 5e-6                 # each returned row really lies at that distance
        kth = dist[:, -1:]
        assert int(((full < kth - 5e-6).sum(1) > 32).sum()) == 0           # nothing closer than the list's end was left out
        assert all(len(set(r.tolist())) == 32 for r in idx)
        # positions whose reference distance is separated from every other pool row's by more than the rounding gap: same index
        srt = full.sort(1).values
        gap_ok = torch.ones_like(ref_d, dtype=torch.bool)
        for r in range(ref_d.shape[0]):
            d = srt[r, :40]
            lonely = torch.ones(40, dtype=torch.bool)
            lonely[1:] &= (d[1:] - d[:-1]) > 2e-6
            lonely[:-1] &= (d[1:] - d[:-1]) > 2e-6
            gap_ok[r] = lonely[:32]
        assert bool((idx[gap_ok] == ref_i[gap_ok]).all()) and float(gap_ok.float().mean()) > 0.5
        assert bool(((dist[:, 1:] > dist[:, :-1]) | (idx[:, 1:] > idx[:, :-1])).all())      # ties: lower index first
        assert bool(((idx[20:30] >= 500) & (idx[20:30] < 900)).all()) and bool((idx[20:30, 0] == 500).all())
    ranked = ops.f0_rerank(ref_i.to(DEV), _t(g["shifted"]).to(DEV), _t(g["pf0"]).to(DEV)).cpu()
    assert torch.equal(ranked, _t(g["ranked"]).long())


def _knn_ties_inputs():
    """== tests/gen_golden.py: knn_ties_inputs (the generator imports the reference and cannot travel to the GPU box)."""
    q = S.clustered_features(200, 1024, seed=23, n_centres=25)
    p = S.clustered_features(4096, 1024, seed=24, n_centres=25)
    g = torch.Generator().manual_seed(25)
    sil = 0.05 * torch.randn(1, 1024, generator=g)
    p[100:164] = p[0:64]
    p[2000:2003] = p[1999:2000]
    p[500:900] = sil
    q[:20] = p[:20]
    q[20:30] = sil
    q[30:40] = sil + 1e-3 * torch.randn(10, 1024, generator=g)
    return q, p


@pytest.mark.parametrize("nq,npool,k", [(1, 32, 32), (37, 129, 5), (130, 1000, 32), (257, 4099, 32)])
def test_knn_ragged(nq, npool, k):
    from oracle import knn_ref
    ops = _ops()
    q = S.clustered_features(nq, 64, 1, n_centres=10)
    p = S.clustered_features(npool, 64, 2, n_centres=10)
    idx, dist = ops.knn_topk(q.to(DEV), p.to(DEV), k, idx_offset=1000)
    ref_idx, ref_d = knn_ref.knn_topk(q, p, k)
    st = knn_ref.topk_agreement(ref_idx, idx.cpu() - 1000, knn_ref.cosine_dist_f64(q, p), tau=5e-7)
    assert st["unexplained"] == 0, st
    assert float((dist.cpu() - ref_d).abs().max()) < 5e-6


def test_knn_shard_merge_equals_single():
    ops = _ops()
    q = S.clustered_features(300, 1024, 5).to(DEV)
    p = S.clustered_features(5000, 1024, 6).to(DEV)
    idx, dist = ops.knn_topk(q, p, 32)
    cuts = [0, 1300, 2600, 3777, 5000]
    pd, pi = [], []
    for a, b in zip(cuts[:-1], cuts[1:]):
        i, d = ops.knn_topk(q, p[a:b].contiguous(), 32, idx_offset=a)
        pi.append(i); pd.append(d)
    midx, mdist = ops.knn_merge(torch.stack(pd), torch.stack(pi))
    assert torch.equal(midx, idx) and torch.equal(mdist, dist)


def test_knn_nan_raises():
    ops = _ops()
    from knn_svc_amd._lib import KnnSvcError
    q = torch.zeros(4, 64, device=DEV)
    p = torch.randn(64, 64, device=DEV)
    with pytest.raises(KnnSvcError):
        ops.knn_topk(q, p, 4)


# ------------------------------------------------------------------ selection
def _select_inputs(g):
    nq, npool = int(g["nq"]), int(g["npool"])
    q = S.clustered_features(nq, 1024, seed=31, n_centres=40)
    p = S.clustered_features(npool, 1024, seed=32, n_centres=40)
    q = (q + torch.roll(q, 1, 0) + torch.roll(q, 2, 0)) / 3
    p = (p + torch.roll(p, 1, 0) + torch.roll(p, 2, 0)) / 3
    return q, p


def test_f0_shift_and_rerank(golden):
    ops = _ops()
    g = golden("g4_select")
    qf0, pf0 = _t(g["qf0"]).to(DEV), _t(g["pf0"]).to(DEV)
    qm, pm = ops.log_f0_median(qf0), ops.log_f0_median(pf0)
    lq = torch.log(_t(g["qf0"])[_t(g["qf0"]) != 0]); lp = torch.log(_t(g["pf0"])[_t(g["pf0"]) != 0])
    assert abs(float(qm[0]) - float(torch.median(lq))) < 1e-6 and int(qm[1]) == len(lq)
    assert abs(float(pm[0]) - float(torch.median(lp))) < 1e-6
    sh = ops.shift_f0(qf0, qm, pm)
    assert _err(sh, _t(g["shifted"]))[1] < 5e-6      # logf/expf ulp differences
    nn32 = _t(g["nn32"]).long().to(DEV)
    rk = ops.f0_rerank(nn32, _t(g["shifted"]).to(DEV), pf0)
    match = float((rk.cpu() == _t(g["ranked"]).long()).all(dim=1).float().mean())
    print("f0 rerank exact-row match:", match)
    assert match == 1.0


def test_concat_reselect(golden):
    ops = _ops()
    g = golden("g4_select")
    q, p = _select_inputs(g)
    qd, pd = q.to(DEV), p.to(DEV)
    qn, _ = ops.row_norms(qd); pn, _ = ops.row_norms(pd)
    a = ops.concat_reselect(_t(g["nn32"]).long()[:, :4].contiguous().to(DEV), qd, qn, pd, pn, concat_weight=0.2)
    b = ops.concat_reselect(_t(g["ranked"]).long()[:, :4].contiguous().to(DEV), qd, qn, pd, pn,
                            _t(g["shifted"]).to(DEV), _t(g["pf0"]).to(DEV), concat_weight=0.2)
    ma = float((a.cpu() == _t(g["sel_plain"]).long()).all(dim=1).float().mean())
    mb = float((b.cpu() == _t(g["sel_f0"]).long()).all(dim=1).float().mean())
    print("concat reselect exact-row match: plain", ma, "f0", mb)
    assert ma == 1.0 and mb == 1.0


@pytest.mark.parametrize("nq,npool,dim", [(1, 40, 64), (2, 40, 256), (3, 9, 768), (5, 12, 1000), (64, 48, 1024), (257, 300, 512)])
def test_concat_reselect_small_and_odd_shapes_vs_oracle(nq, npool, dim):
    """The frame-sequential walk against the oracle away from the golden's shape: one / two / three frames (the prologue stages frames
    0 and 1 and the prefetch runs two frames ahead), pools so small that "previous selection + 1" hits the clamp at the pool's
    end and candidates repeat, feature widths that are not the two full column halves of 1024.  Both variants."""
    from oracle import select_ref
    ops = _ops()
    gen = torch.Generator().manual_seed(nq * 1000 + dim)
    p = S.clustered_features(npool, dim, seed=dim, n_centres=5)
    q = p[torch.randint(0, npool, (nq,), generator=gen)] + 0.05 * torch.randn(nq, dim, generator=gen)
    idx4 = torch.randint(0, npool, (nq, 4), generator=gen)
    idx4[:, 0] = torch.randint(max(npool - 3, 0), npool, (nq,), generator=gen)      # rows next to the pool's end: the clamp
    sf0 = torch.rand(nq, generator=gen) * 200 + 100
    pf0 = torch.rand(npool, generator=gen) * 200 + 100
    qd, pd = q.to(DEV), p.to(DEV)
    qn, _ = ops.row_norms(qd); pn, _ = ops.row_norms(pd)
    a = ops.concat_reselect(idx4.to(DEV), qd, qn, pd, pn, concat_weight=0.2).cpu()
    b = ops.concat_reselect(idx4.to(DEV), qd, qn, pd, pn, sf0.to(DEV), pf0.to(DEV), concat_weight=0.2).cpu()
    ra = select_ref.concat_reselect(idx4.clone(), q, p, concat_weight=0.2)
    rb = select_ref.concat_reselect(idx4.clone(), q, p, sf0, pf0, concat_weight=0.2)
    # duplicated candidates tie exactly (same row twice): the order among equal costs is torch.topk's, which the oracle inherits;
    # compare the selected SETS per frame plus the exact rows wherever a frame's costs are distinct (no duplicate candidates)
    for got, ref, name in ((a, ra, "plain"), (b, rb, "f0")):
        same_sets = all(sorted(got[i].tolist()) == sorted(ref[i].tolist()) for i in range(nq))
        assert same_sets, f"{name}: selected sets differ"
        if npool >= 100:
            assert float((got == ref).all(dim=1).float().mean()) == 1.0, f"{name}: order differs"


# ------------------------------------------------------------------ smoothness weights
def test_smooth_weights(golden):
    ops = _ops()
    g = golden("g5_smooth")
    npool = int(g["npool"])
    p = S.clustered_features(npool, 1024, seed=int(g["p_seed"]), n_centres=30)
    p = (p + torch.roll(p, 1, 0) + torch.roll(p, 2, 0)) / 3
    idx = _t(g["idx"]).long().to(DEV)
    w, it = ops.smooth_weights(idx, p.to(DEV), 0.1, return_iters=True)
    e = _err(w, _t(g["w_wavlm"]))[0]
    print("wavlm weights max|d|", e, "iters", int(it), "ref", int(g["iters_wavlm"]))
    assert e < 4e-4 and int(it) == int(g["iters_wavlm"])       # measured 1.2e-4 (IEEE divides / square roots in the loop: 3.5e-4)
    assert abs(float(w.sum(1).mean()) - 1.0) < 1e-5
    wh, ith = ops.smooth_weights(idx, _t(g["harm_pool"]).to(DEV), 1000.0, return_iters=True)
    e = _err(wh, _t(g["w_harm"]))[0]
    print("harm weights max|d|", e, "iters", int(ith), "ref", int(g["iters_harm"]))
    assert e < 4e-4 and int(ith) == int(g["iters_harm"])       # measured 2.8e-4 (IEEE form: 2.4e-4)
    # weighted gather
    out = ops.weighted_gather(idx, w, p.to(DEV))
    ref = (p[idx.cpu().reshape(-1)].reshape(-1, 4, 1024) * w.cpu()[..., None]).sum(1)
    assert _err(out, ref)[0] < 1e-5


# ------------------------------------------------------------------ synth / side features
def test_additive_synth(golden):
    ops = _ops()
    g = golden("g6_synth")
    f0, amp = _t(g["f0"]).to(DEV), _t(g["amp"]).to(DEV)
    N = f0.numel()
    pw = torch.randn(32, 1, 3); pb = torch.randn(32)
    cond = torch.empty(N * 320, 64, device=DEV)
    exc = ops.additive_synth(f0, amp, pw.reshape(32, 3).contiguous().to(DEV), pb.to(DEV), cond[:, 32:], 64, want_exc=True)
    e = _err(exc, _t(g["wave"]))[0]
    print("additive synth max|d|", e)
    assert e < 2e-5
    ref_cond = F.conv1d(_t(g["wave"])[None, None], pw, pb, padding=1)[0].T
    assert _err(cond[:, 32:], ref_cond)[0] < 5e-5


def test_stft_and_harmonics(golden):
    from knn_svc_amd import features
    ops = _ops()
    g = golden("g6_synth")
    wav, _ = S.synth_clip(320 * 120, int(g["clip_seed"]))
    spec = features.stft_mag(torch.from_numpy(wav).to(DEV))[:120]
    assert _err(spec, _t(g["spec"]))[0] < 2e-4
    harm = ops.harmonic_amps(_t(g["spec"]).to(DEV), _t(g["f0w"]).to(DEV))
    assert _err(harm, _t(g["harm"]))[0] < 1e-6


# ------------------------------------------------------------------ prematch kernels (per_spk_extract)
@pytest.mark.parametrize("mode", ["f16x2", "fp32"])
@pytest.mark.parametrize("nq,npool,lo,hi", [(50, 100, 20, 95), (130, 3000, 1000, 1130), (33, 700, 0, 33)])
def test_knn_self_mask(nq, npool, lo, hi, mode, monkeypatch):
    """dists[:, start:end] = 1 before topk (ddsp_prematch_dataset.py:1606-1607): masked rows compete at exactly 1;
    with a pool this small they must show up, in ascending index order (the build's tie rule)."""
    from oracle import knn_ref
    monkeypatch.setenv("KNNSVC_KNN", mode)
    ops = _ops()
    p = S.clustered_features(npool, 64, 5, n_centres=12)
    q = p[lo:lo + nq].clone() if hi - lo >= nq else S.clustered_features(nq, 64, 6, n_centres=12)
    idx, dist = ops.knn_topk(q.to(DEV), p.to(DEV), 32, mask=(lo, hi))
    idx, dist = idx.cpu(), dist.cpu()
    d = knn_ref.cosine_dist_all(q, p)
    d[:, lo:hi] = 1
    ref = d.topk(k=32, dim=-1, largest=False)
    assert float((dist - ref.values).abs().max()) < 5e-6
    masked = (idx >= lo) & (idx < hi)
    assert bool((dist[masked] == 1.0).all())
    exact = knn_ref.cosine_dist_f64(q, p)
    exact[:, lo:hi] = 1.0
    st = knn_ref.topk_agreement(ref.indices, idx, exact, tau=5e-7)
    assert st["unexplained"] == 0, st
    for r in range(nq):                      # ties at exactly 1: lower pool index first
        m = idx[r][masked[r]]
        assert bool((m[1:] > m[:-1]).all())
    if npool == 100:
        assert int(masked.sum()) > 0          # 5 unmasked rows only: the mask region must fill the list


def test_round_f16_and_amp_ratio():
    from oracle import prematch_ref
    ops = _ops()
    g = torch.Generator().manual_seed(3)
    x = torch.randn(1003, 37, generator=g) * torch.logspace(-9, 5, 37)[None]      # subnormal-fp16 .. overflow range
    got = ops.round_f16(x.to(DEV)).cpu()
    assert torch.equal(got, x.half().float())
    spec_q = torch.rand(211, 200, generator=g) * 3
    spec_p = torch.rand(1500, 200, generator=g) * 3
    spec_p[7] = 0                                                                 # silent frame: ratio = L1 / 1e-5
    idx = torch.randint(0, 1500, (211, 4), generator=g)
    idx[0, 0] = 7
    got = ops.amp_ratio(spec_q.to(DEV), spec_p.to(DEV), idx.to(DEV)).cpu()
    ref = prematch_ref.amp_ratio(spec_q, spec_p, idx)
    assert float(((got - ref).abs() / ref.abs()).max()) < 2e-6


def test_smooth_weights_with_amp_ratio():
    """compute_weight_with_amp (ddsp_prematch_dataset.py:684-804): the row scale enters the Gram matrices."""
    from oracle import smooth_ref
    ops = _ops()
    g = torch.Generator().manual_seed(9)
    pool = torch.rand(900, 49, generator=g) * 0.02
    pool = (pool + torch.roll(pool, 1, 0) + torch.roll(pool, 2, 0)) / 3
    idx = torch.randint(0, 900, (120, 4), generator=g)
    idx[5] = torch.tensor([0, 899, 1, 898])                                       # clamped +-1 neighbours
    ar = torch.rand(120, 4, generator=g) * 2 + 0.3
    ref, it_ref = smooth_ref.smooth_weights(idx, pool, 1000.0, return_iters=True, row_scale=ar)
    w, it = ops.smooth_weights(idx.to(DEV), pool.to(DEV), 1000.0, return_iters=True, row_scale=ar.to(DEV))
    e = _err(w, ref)[0]
    print("amp-scaled weights max|d|", e, "iters", int(it), "ref", it_ref)
    assert e < 2e-5 and abs(int(it) - it_ref) <= 2        # measured (r03): 1.1e-6, 401 iterations on both sides
    plain = ops.smooth_weights(idx.to(DEV), pool.to(DEV), 1000.0)
    assert float((plain - w).abs().max()) > 1e-3                                  # the scale is not ignored


@pytest.mark.parametrize("scale_q,scale_p", [(1.0, 1.0), (0.01, 30.0), (200.0, 0.05)])
def test_knn_screen_is_exact(scale_q, scale_p, monkeypatch):
    """The approximate-distance screen of knnsvc_knn_select only skips work: indices and distances are bit-identical to
    evaluating the reference formula on every element, also with badly unbalanced norms, near-duplicate pool rows and a
    self-mask; a NaN in a pool row is still reported."""
    ops = _ops()
    g = torch.Generator().manual_seed(21)
    p = S.clustered_features(6000, 1024, 31, n_centres=20) * scale_p
    p[100:140] = p[50:90] * (1 + 1e-6 * torch.randn(40, 1, generator=g))          # near-ties
    p[2000:2100] *= torch.logspace(-3, 3, 100)[:, None]                           # wildly different row norms
    q = S.clustered_features(300, 1024, 32, n_centres=20) * scale_q
    q[:40] = p[50:90] / scale_p * scale_q
    outs = {}
    for screen in ("1", "0"):
        monkeypatch.setenv("KNNSVC_KNN_SCREEN", screen)
        outs[screen] = [t.cpu() for t in ops.knn_topk(q.to(DEV), p.to(DEV), 32, mask=(50, 70))]
    assert torch.equal(outs["1"][0], outs["0"][0]) and torch.equal(outs["1"][1], outs["0"][1])
    monkeypatch.setenv("KNNSVC_KNN_SCREEN", "1")
    pb = p.clone(); pb[4321, 7] = float("nan")
    with pytest.raises(ops.KnnSvcError):
        ops.knn_topk(q.to(DEV), pb.to(DEV), 32)
    pz = p.clone(); pz[777] = 0                                                  # zero row: the formula gives +-inf or NaN there
    a = ops.knn_topk(q.to(DEV), pz.to(DEV), 32, check_nan=False)
    monkeypatch.setenv("KNNSVC_KNN_SCREEN", "0")
    b = ops.knn_topk(q.to(DEV), pz.to(DEV), 32, check_nan=False)
    assert torch.equal(a[0], b[0])


def test_knn_fused_route_equals_dot_matrix_route(monkeypatch):
    """Large query sets take the fused route (knnsvc_knn_screen + knnsvc_knn_refine: threshold from a strided pool sample,
    products screened in the GEMM's registers, reference formula on the survivors only — no [Nq, Np] dot matrix).  Its
    result must be IDENTICAL to the dot-matrix route: same indices, same distance bits — plain, with a self-mask that covers
    sampled rows, with an offset, and when the candidate buffer overflows (fallback)."""
    from knn_svc_amd import ops
    nq, npool = 4096 + 37, 40000 + 123
    q = S.clustered_features(nq, 1024, 51, n_centres=60)
    p = S.clustered_features(npool, 1024, 52, n_centres=60)
    p[5000:5000 + nq // 8] = q[: nq // 8]                      # exact duplicates of some queries inside the pool (distance ~0)
    qd, pd = q.to(DEV), p.to(DEV)

    def run(fused, **kw):
        monkeypatch.setenv("KNNSVC_KNN_FUSED", "1" if fused else "0")
        return ops.knn_topk(qd, pd, 32, **kw)
    for kw in (dict(), dict(mask=(4990, 5600), idx_offset=777)):
        i0, d0 = run(False, **kw)
        i1, d1 = run(True, **kw)
        assert torch.equal(i0, i1) and torch.equal(d0, d1), kw
    # the fused route really ran (no dot matrix): its kernels leave the candidate counts behind a small buffer; check via cap
    monkeypatch.setattr(ops, "KNN_FUSED_CAP", 8)               # every row overflows -> fallback to the dot-matrix route
    i2, d2 = run(True)
    i0, d0 = run(False)
    assert torch.equal(i0, i2) and torch.equal(d0, d2)
    # deferred mode (stream pipelines): nothing is read inside the search; the flag carries the overflow bit and raise_if_nan turns it
    # into KnnOverflow, which ops.retry_on_overflow answers by repeating the work on the dot-matrix route
    _i, _d, fl = run(True, check_nan=False, return_flag=True)
    assert int(fl.item()) & ops.KNN_OVERFLOW
    with pytest.raises(ops.KnnOverflow):
        ops.raise_if_nan(fl)
    calls = []

    def work():
        calls.append(ops.knn_fused_on())
        i, d, f = ops.knn_topk(qd, pd, 32, check_nan=False, return_flag=True)
        ops.raise_if_nan(f)
        return i, d
    i3, d3 = ops.retry_on_overflow(work)
    assert calls == [True, False] and torch.equal(i0, i3) and torch.equal(d0, d3)
    # NaN in a query row is reported by either route
    from knn_svc_amd._lib import KnnSvcError
    qn_ = qd.clone(); qn_[100, 7] = float("nan")
    monkeypatch.setattr(ops, "KNN_FUSED_CAP", 4096)
    monkeypatch.setenv("KNNSVC_KNN_FUSED", "1")
    with pytest.raises(KnnSvcError):
        ops.knn_topk(qn_, pd, 32)


@pytest.mark.parametrize("nq,npool,dim", [(1500, 30000, 1024), (300, 30000 + 77, 1024), (257, 8192 + 5, 256), (3000 + 11, 20000, 512)])
def test_knn_fused_epochs_equal_dot_matrix_route_at_small_sizes(nq, npool, dim, monkeypatch):
    """Round 4: the fused route starts at 256 query frames — the first epoch has no thresholds (every tile bounds its rows
    itself), later epochs take the k-th key so far.  The north-star size (1500 x 30 000) and ragged sizes around it give the
    dot-matrix route's indices and distance bits, plain, with an offset and with a mask that swallows a whole first-epoch tile."""
    from knn_svc_amd import ops
    q = S.clustered_features(nq, dim, 71, n_centres=50)
    p = S.clustered_features(npool, dim, 72, n_centres=50)
    p[700:700 + 64] = q[:64]                                   # exact duplicates of some queries
    qd, pd = q.to(DEV), p.to(DEV)

    def run(fused, **kw):
        monkeypatch.setenv("KNNSVC_KNN_FUSED", "1" if fused else "0")
        c0 = dict(ops.KNN_ROUTE_COUNTS)
        out = ops.knn_topk(qd, pd, 32, **kw)
        assert (ops.KNN_ROUTE_COUNTS["fused"] - c0["fused"] > 0) == fused
        return out
    for kw in (dict(), dict(mask=(200, 1100), idx_offset=12345), dict(mask=(npool - 300, npool))):
        i0, d0 = run(False, **kw)
        i1, d1 = run(True, **kw)
        assert torch.equal(i0, i1) and torch.equal(d0, d1), kw
    assert len(ops.knn_epochs(nq, npool)) >= 1


@pytest.mark.parametrize("k", [1, 5, 31])
def test_knn_fused_route_small_k_and_masks_that_leave_few_rows(k, monkeypatch):
    """The fused route for k < 32 (its first epoch always bounds a row by 32 of a tile's columns, whatever k), for a mask that
    swallows almost the whole pool (most candidates compete at exactly 1: a 9000-way tie cut by index) and for a mask that starts
    inside one epoch and ends inside the next — against the dot-matrix route, bit for bit."""
    from knn_svc_amd import ops
    q = S.clustered_features(300, 512, 61, n_centres=12).to(DEV)
    p = S.clustered_features(9000 + 13, 512, 62, n_centres=12).to(DEV)
    for mask in (None, (40, 9000), (2000, 6100), (0, 9013)):
        outs = []
        for fused in ("0", "1"):
            monkeypatch.setenv("KNNSVC_KNN_FUSED", fused)
            c0 = ops.KNN_ROUTE_COUNTS["fused"]
            i, d, f = ops.knn_topk(q, p, k, mask=mask, idx_offset=7, check_nan=False, return_flag=True)
            assert (ops.KNN_ROUTE_COUNTS["fused"] > c0) == (fused == "1")
            fl = int(f.item())
            if fl & ops.KNN_OVERFLOW:            # (a whole-pool tie may overflow the candidate buffer: flagged, the retry is exact)
                assert fused == "1" and mask is not None
                i, d = ops.knn_topk(q, p, k, mask=mask, idx_offset=7)
            outs.append((i.cpu(), d.cpu()))
        assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]), (k, mask)
        assert outs[0][0].shape == (300, k) and int(outs[0][0].min()) >= 7 and int(outs[0][0].max()) < 9013 + 7
        if mask == (0, 9013):                    # everything competes at 1: the k lowest indices, in order
            assert bool((outs[1][1] == 1.0).all()) and bool((outs[1][0] == torch.arange(7, 7 + k)[None, :]).all())


def test_knn_fused_route_survives_thousands_of_identical_pool_rows_and_silence(monkeypatch):
    """with many bit-identical pool rows (digital silence) every tied row shares one dot product.  The
    thresholds of the fused route are KEYS (distance bits, pool index) over rows already searched with the SAME operand split
    (no separately split sample any more), so ties are cut by index exactly as the full evaluation cuts them, a row never ends
    with fewer than k candidates, and a short list would be flagged rather than written.  Pool: 6000 identical rows of a small
    norm in the middle of louder material (the largest norm is far above twice theirs), queries that are nearest to them, and
    queries that ARE them."""
    from knn_svc_amd import ops
    nq, npool = 600, 24000
    g = torch.Generator().manual_seed(5)
    p = S.clustered_features(npool, 1024, 81, n_centres=30) * 40.0
    sil = torch.randn(1, 1024, generator=g) * 0.02
    p[9000:15000] = sil
    p[20000:20400] = sil * 1.0000001                           # near-identical, not identical
    q = S.clustered_features(nq, 1024, 82, n_centres=30) * 40.0
    q[:200] = sil + 1e-4 * torch.randn(200, 1024, generator=g)
    q[200:260] = sil
    qd, pd = q.to(DEV), p.to(DEV)
    monkeypatch.setenv("KNNSVC_KNN_FUSED", "0")
    i0, d0, f0 = ops.knn_topk(qd, pd, 32, check_nan=False, return_flag=True)
    monkeypatch.setenv("KNNSVC_KNN_FUSED", "1")
    c0 = ops.KNN_ROUTE_COUNTS["fused"]
    i1, d1, f1 = ops.knn_topk(qd, pd, 32, check_nan=False, return_flag=True)
    assert ops.KNN_ROUTE_COUNTS["fused"] > c0
    fl = int(f1.item())
    assert int(f0.item()) == 0 and (fl & 1) == 0
    if fl & ops.KNN_OVERFLOW:      # allowed outcome for pathological data: flagged, and the retry gives the dot-matrix result
        i1, d1 = ops.knn_topk(qd, pd, 32)
    assert torch.equal(i0, i1) and torch.equal(d0, d1)
    assert int(i1.max()) < npool and int(i1.min()) >= 0
    # the tied rows come out in index order (lower index first), as torch.topk's tie rule on equal distances is replayed
    ti, td = i1[200:260], d1[200:260]
    assert bool(((td[:, 1:] > td[:, :-1]) | (ti[:, 1:] > ti[:, :-1])).all())
    assert int(((ti >= 9000) & (ti < 15000)).sum()) >= 60 * 16                     # the silence block supplies most of their lists


def test_knn_nan_rows_leave_valid_indices_behind():
    """A NaN query row has no neighbours: its list stays unfilled.  The flag is only read after the later stages have been
    enqueued (stream pipelines), and those gather pool rows through the indices — so an unfilled place must be a valid row
    (index 0 + offset, NaN distance), never 0xFFFFFFFF (round 3: a NaN source in a serving batch took the process down with a
    memory fault instead of raising "containing nan")."""
    from knn_svc_amd import ops
    q = S.clustered_features(700, 1024, 91, n_centres=20)
    p = S.clustered_features(9000, 1024, 92, n_centres=20)
    q[13] = float("nan"); q[640, 100] = float("nan")
    for fused in ("0", "1"):
        os.environ["KNNSVC_KNN_FUSED"] = fused
        try:
            i, d, f = ops.knn_topk(q.to(DEV), p.to(DEV), 32, idx_offset=5, check_nan=False, return_flag=True)
        finally:
            os.environ.pop("KNNSVC_KNN_FUSED", None)
        assert int(f.item()) & 1
        assert int(i.min()) >= 5 and int(i.max()) < 9005
        assert bool(torch.isnan(d[13]).all()) and bool((i[13] == 5).all())
        ok = torch.ones(700, dtype=torch.bool); ok[13] = ok[640] = False
        assert bool(torch.isfinite(d[ok.to(DEV)]).all())


@pytest.mark.parametrize("sr,n", [(44100, 44100 + 37), (48000, 30001), (8000, 12345), (22050, 22050), (24000, 7)])
def test_resample_matches_torchaudio_restatement(sr, n):
    """features.resample (GPU, one framed-signal GEMM) against the oracle's restatement of torchaudio's sinc_interp_hann
    resampler: same length law, values within fp32 rounding of the filter sum."""
    from knn_svc_amd import features
    from oracle import audio_ref
    g = np.random.default_rng(sr)
    x = (0.5 * g.standard_normal(n)).astype(np.float32)
    ref = audio_ref.resample(x[None], sr, 16000)[0]
    got = features.resample(torch.from_numpy(x).to(DEV), sr, 16000).cpu().numpy()
    assert got.shape == ref.shape, (got.shape, ref.shape)
    assert float(np.abs(got - ref).max()) < 2e-6


# ------------------------------------------------------------------ f0 front end (Harvest itself: tests/test_gpu_f0.py)
def test_missing_f0_cache_is_generated_next_to_the_audio(tmp_path, monkeypatch):
    """Without `<stem>_f0.npy` the track is computed (Harvest on the GPU) and written next to the audio exactly as the reference
    does (ddsp_prematch_dataset.py:376-379); the second load reads the cache."""
    from knn_svc_amd import audio_io, matching
    wav, f0_true = S.synth_clip(2 * 16000, 9)
    for sub in ("h",):
        d = tmp_path / sub; d.mkdir()
        p = d / "a.wav"
        audio_io.write_wav_pcm16(str(p), wav, 16000)
        w, f0 = matching.load_utterance(p)
        assert (d / "a_f0.npy").is_file() and f0.shape == (len(w) // 320 + 1,) and f0.dtype == np.float32
        assert np.array_equal(np.load(d / "a_f0.npy"), f0)
        v = (f0_true[:len(f0)] > 0) & (f0 > 0)
        assert v.sum() > 20 and np.median(np.abs(f0[v] - f0_true[:len(f0)][v]) / f0_true[:len(f0)][v]) < 0.02
        w2, f02 = matching.load_utterance(p)
        assert np.array_equal(f0, f02)


def test_grouped_knn_results_do_not_depend_on_the_grouping(monkeypatch):
    """Dataset mode searches the frames of several items per kNN call, group after group on a stream of its own
    (matching.grouped_knn): whatever the group size — one item per search, a few, everything at once (6578 frames against 40 000
    pool rows: the fused screen + refine route, asserted through ops.KNN_ROUTE_COUNTS) — every item gets the same neighbours as from
    a search of its own.  The consumer sits on ANOTHER stream and only waits for the item's event: grouped_knn must hand one out for
    every item, also when a single group is searched on the caller's own stream (round 2's race)."""
    from knn_svc_amd import matching, ops as kops
    ops = _ops()
    P = S.clustered_features(40000, 1024, 5, n_centres=50).to(DEV)
    assert P.shape[0] >= kops.KNN_FUSED_MIN_P
    lens = [300, 1500, 77, 2600, 900, 1, 1200]
    assert sum(lens) >= kops.KNN_FUSED_MIN_Q
    qs = {i: S.clustered_features(n, 1024, 40 + i, n_centres=50).to(DEV) for i, n in enumerate(lens)}
    prep = matching.prepare_pool(P)
    want = {i: ops.knn_topk(q, P, 32, check_nan=False, return_flag=True)[0] for i, q in qs.items()}
    torch.cuda.synchronize()
    for gf in (1, 1000, 3000, 10 ** 9):
        monkeypatch.setattr(matching, "KNN_GROUP_FRAMES", gf)
        flags = []
        fused0 = kops.KNN_ROUTE_COUNTS["fused"]
        nn, ready = matching.grouped_knn(list(qs), qs, P, prep, flags)
        assert set(ready) == set(qs), "every item needs an ordering event"
        sizes, n = [], 0
        for L in lens:                                      # the grouping rule of grouped_knn
            n += L
            if n >= gf:
                sizes.append(n); n = 0
        sizes += [n] if n else []
        assert kops.KNN_ROUTE_COUNTS["fused"] - fused0 == sum(g >= kops.KNN_FUSED_MIN_Q for g in sizes), (gf, sizes)
        assert gf < 10 ** 9 or sizes == [sum(lens)]
        side = torch.cuda.Stream()
        got = {}
        with torch.cuda.stream(side):
            for i in qs:
                matching.wait_for_neighbours(nn[i], ready[i], DEV)
                got[i] = torch.equal(nn[i], want[i])
        torch.cuda.current_stream().wait_stream(side)
        assert all(got.values()), (gf, got)
        for f in flags:
            ops.raise_if_nan(f)
    with pytest.raises(RuntimeError):
        matching.wait_for_neighbours(nn[0], None, DEV)


def test_side_features_batched_equal_per_file():
    """matching.side_features_many (one reflect-pad / DFT GEMM / magnitude + harmonics pass per pool, knnsvc_reflect_pad_batch +
    knnsvc_spec_harm) against matching.side_features file by file (ddsp_prematch_dataset.py:361-404): ragged lengths, unvoiced
    frames, harmonics beyond Nyquist — the same bits."""
    from knn_svc_amd import matching
    lens = [16000 * 3 + 7, 16000 * 3 + 7, 5000, 320 * 40, 16000 * 2 - 1, 48000 + 319, 9000]
    wavs, f0s, Ts = [], [], []
    for i, n in enumerate(lens):
        w, f0 = S.synth_clip(n, 300 + i)
        T = n // 320
        f0 = f0[:T + 1].copy()
        f0[::7] = 0.0                                        # unvoiced frames
        f0[3::11] *= 9.0                                     # harmonics past the Nyquist bin
        wavs.append(torch.from_numpy(w).to(DEV)); f0s.append(f0.astype(np.float32)); Ts.append(T)
    many = matching.side_features_many(wavs, f0s, Ts)
    for w, f0, T, (f0m, hm, sm) in zip(wavs, f0s, Ts, many):
        f0o, ho, so = matching.side_features(w, f0, T)
        assert torch.equal(f0m, f0o) and sm.shape == so.shape == (T, 200) and hm.shape == ho.shape == (T, 49)
        assert torch.equal(sm, so), float((sm - so).abs().max())
        assert torch.equal(hm, ho), float((hm - ho).abs().max())


@pytest.mark.parametrize("C_,k,d", [(32, 3, 1), (32, 11, 5), (64, 7, 3), (64, 11, 5), (64, 3, 5), (32, 7, 1), (128, 3, 1), (128, 11, 5)])
def test_resblock_pair_equals_two_launches(C_, k, d):
    """knnsvc_resblock_pair (one ResBlock1 iteration, hifigan/ddsp_models.py:13-44, in one launch with the inner activation in LDS)
    against the two knnsvc_conv_gemm launches the generator otherwise issues: same main loop, same epilogue arithmetic, same
    f16x2 split of the inner activation -> the same bits, output and range slot; ragged length (tiles end mid-sequence, the
    second convolution's zero padding of t1 at both ends), and the bucketed form with a device-side length."""
    ops = _ops()
    g = torch.Generator().manual_seed(100 + C_ + k + d)
    T = 3 * 246 + 57
    x = (torch.randn(T, C_, generator=g) * 1.7).to(DEV)
    w1 = ops.attach_split(ops.pack_conv_weight(torch.randn(C_, C_, k, generator=g) / (C_ * k) ** 0.5).to(DEV))
    w2 = ops.attach_split(ops.pack_conv_weight(torch.randn(C_, C_, k, generator=g) / (C_ * k) ** 0.5).to(DEV))
    b1 = torch.randn(C_, generator=g).to(DEV); b2 = torch.randn(C_, generator=g).to(DEV)
    w1_ = w1.cpu().view(C_, k, C_).permute(0, 2, 1)
    bound = (float(w1_.abs().sum(dim=(1, 2)).max()), float(b1.abs().max()))
    slope = 0.1

    def two(xin, t, dyn=None):
        sx = ops.absmax(xin[:t] if dyn is None else xin); so = ops.new_slot(DEV)
        t1 = torch.full((xin.shape[0], C_), float("nan"), device=DEV); out = torch.zeros(xin.shape[0], C_, device=DEV)
        ops.conv_gemm(xin, w1, t1, m=xin.shape[0], n=C_, cin=C_, taps=k, dil=d, pad=(k * d - d) // 2, t_in=xin.shape[0], bias=b1,
                      a_slope=slope, act=ops.ACT_LRELU, act_slope=slope, x_absmax=sx, dyn=dyn)
        ops.conv_gemm(t1, w2, out, m=xin.shape[0], n=C_, cin=C_, taps=k, pad=(k - 1) // 2, t_in=xin.shape[0], bias=b2, resid=xin, ldr=C_,
                      x_absmax=sx, x_bound=bound, out_absmax=so, dyn=dyn)
        return out, so, sx

    def one(xin, t, sx, dyn=None):
        so = ops.new_slot(DEV)
        out = torch.zeros(xin.shape[0], C_, device=DEV)
        ops.resblock_pair(xin, w1, b1, w2, b2, out, t=xin.shape[0], channels=C_, taps=k, dil=d, slope=slope, x_absmax=sx,
                          t1_bound=bound, out_absmax=so, dyn=dyn)
        return out, so
    o2, s2, sx = two(x, T)
    o1, s1 = one(x, T, sx)
    assert torch.equal(o1, o2), float((o1 - o2).abs().max())
    assert float(s1.max()) == float(s2.max()) == float(o2.abs().max())
    # against fp64 (the pair is a real convolution pair, not merely self-consistent)
    xd = x.cpu().double().t()[None]
    lr = torch.nn.functional.leaky_relu
    t1r = lr(F.conv1d(lr(xd, slope), w1_.double(), b1.cpu().double(), dilation=d, padding=(k * d - d) // 2), slope)
    ref = F.conv1d(t1r, w2.cpu().view(C_, k, C_).permute(0, 2, 1).double(), b2.cpu().double(), padding=(k - 1) // 2) + xd
    assert float((o1.cpu().double().t()[None] - ref).abs().max()) < 2e-5
    # bucketed: laid out for Tb rows, valid length in a device int (frames x 8 rows per frame)
    Tb, n_valid = 8 * 110, 99
    xb = torch.zeros(Tb, C_, device=DEV); xb[:8 * n_valid] = x[:8 * n_valid]
    nd = torch.tensor([n_valid], device=DEV, dtype=torch.int32)
    ob2, sb2, sxb = two(xb, 8 * n_valid, dyn=(nd, 110))
    ob1, sb1 = one(xb, 8 * n_valid, sxb, dyn=(nd, 110))
    assert torch.equal(ob1[:8 * n_valid], ob2[:8 * n_valid]) and float(ob1[8 * n_valid:].abs().max()) == 0.0
    assert float(sb1.max()) == float(sb2.max())


@pytest.mark.parametrize("C_,T", [(256, 3750), (128, 15000), (256, 1500), (64, 7000), (32, 9001)])
def test_branches_in_one_grid_equal_one_launch_per_branch(C_, T):
    """knnsvc_conv_gemm_multi / knnsvc_resblock_pair_multi: the launches of one step of a generator stage's three ResBlock
    branches (kernel sizes 11 / 7 / 3, hifigan/ddsp_models.py:206-227) as ONE grid (blockIdx.y = branch) against one launch per
    branch: same descriptors, same kernel body -> the same bits, outputs and range slots, for the windowed convolutions
    (C = 128 / 256: both launches of a pair; whatever tile shape the merged grid's size selects) and for the fused pairs
    (C = 32 / 64), also with a device-side length."""
    ops = _ops()
    g = torch.Generator().manual_seed(7 + C_)
    x = (torch.randn(T, C_, generator=g) * 1.3).to(DEV)
    sx = ops.absmax(x)
    slope, d = 0.1, 3
    br = []
    for k in (11, 7, 3):
        w1 = ops.attach_split(ops.pack_conv_weight(torch.randn(C_, C_, k, generator=g) / (C_ * k) ** 0.5).to(DEV))
        w2 = ops.attach_split(ops.pack_conv_weight(torch.randn(C_, C_, k, generator=g) / (C_ * k) ** 0.5).to(DEV))
        b1 = torch.randn(C_, generator=g).to(DEV); b2 = torch.randn(C_, generator=g).to(DEV)
        bound = (float(w1.cpu().abs().sum(1).max()), float(b1.abs().max()))
        br.append(dict(k=k, w1=w1, w2=w2, b1=b1, b2=b2, bound=bound))

    def run(merged, dyn=None, rows=T):
        outs, slots, names = [], [], []
        c1, c2, pairs = [], [], []
        t1s = [torch.full((rows, C_), float("nan"), device=DEV) for _ in br]
        for b, t1 in zip(br, t1s):
            out = torch.zeros(rows, C_, device=DEV); so = ops.new_slot(DEV)
            k = b["k"]
            if ops.resblock_pair_ok(C_, k, d):
                ops.resblock_pair(x[:rows], b["w1"], b["b1"], b["w2"], b["b2"], out, t=rows, channels=C_, taps=k, dil=d, slope=slope, x_absmax=sx,
                                  t1_bound=b["bound"], out_absmax=so, dyn=dyn, defer=pairs if merged else None)
            else:
                ops.conv_gemm(x[:rows], b["w1"], t1, m=rows, n=C_, cin=C_, taps=k, dil=d, pad=(k * d - d) // 2, t_in=rows, bias=b["b1"], a_slope=slope,
                              act=ops.ACT_LRELU, act_slope=slope, x_absmax=sx, dyn=dyn, defer=c1 if merged else None)
                if not merged:
                    names.append(ops.last_conv_kernel())
                ops.conv_gemm(t1, b["w2"], out, m=rows, n=C_, cin=C_, taps=k, pad=(k - 1) // 2, t_in=rows, bias=b["b2"], resid=x[:rows], ldr=C_,
                              x_absmax=sx, x_bound=b["bound"], out_absmax=so, dyn=dyn, defer=c2 if merged else None)
            outs.append(out); slots.append(so)
        if merged:
            ops.resblock_pair_multi(pairs)
            ops.conv_gemm_multi(c1)
            if c1:
                names.append(ops.last_conv_kernel())
            ops.conv_gemm_multi(c2)
        torch.cuda.synchronize()
        return outs, slots, names
    a, sa, na = run(False)
    b, sb, nb = run(True)
    print(f"C = {C_}, T = {T}: per-branch kernels {na}, merged grid {nb}")
    for j in range(3):
        assert torch.equal(a[j], b[j]), (j, float((a[j] - b[j]).abs().max()))
        assert float(sa[j].max()) == float(sb[j].max()) == float(a[j].abs().max())
    if nb:
        assert nb[0].endswith("x"), nb                    # the merged grid really was one launch of the windowed kernel
    # bucketed: the same launches laid out for Tb rows with a device-side length
    Tb = (T // 8) * 8
    nd = torch.tensor([Tb // 8 - 3], device=DEV, dtype=torch.int32)
    a, _sa, _ = run(False, dyn=(nd, Tb // 8), rows=Tb)
    b, _sb, _ = run(True, dyn=(nd, Tb // 8), rows=Tb)
    n_valid = 8 * (Tb // 8 - 3)
    for j in range(3):
        assert torch.equal(a[j][:n_valid], b[j][:n_valid])
