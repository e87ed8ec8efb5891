"""Thin tensor-level wrappers over the C ABI (include/knnsvc_hip.h).

PyTorch-ROCm is used for device memory and streams only: every function takes
contiguous fp32 tensors on the GPU, passes raw device pointers plus the current
HIP stream to ``libknnsvc_hip.so`` and returns tensors.  No op has a torch
fallback — a missing library or a CPU tensor raises.
"""
from __future__ import annotations

import ctypes as C
import math

import torch

from . import _lib
from ._lib import ConvDesc, KnnSvcError, PairDesc, check

ACT_NONE, ACT_GELU, ACT_LRELU, ACT_TANH = 0, 1, 2, 3


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _need(t, dtype=torch.float32, name="tensor"):
    if not (isinstance(t, torch.Tensor) and t.is_cuda):
        raise KnnSvcError(f"{name}: expected a GPU tensor (knn_svc_amd has no CPU path)")
    if t.dtype != dtype:
        raise KnnSvcError(f"{name}: expected {dtype}, got {t.dtype}")
    return t


# ------------------------------------------------------------------ weight packing (host side, once per load)
def pack_conv_weight(w: torch.Tensor) -> torch.Tensor:
    """Conv1d weight [Cout, Cin, k] -> GEMM B operand [Cout, k*Cin] (tap major, channel minor)."""
    return w.permute(0, 2, 1).reshape(w.shape[0], -1).contiguous()


def pack_grouped_conv_weight(w: torch.Tensor, groups: int) -> torch.Tensor:
    """Grouped Conv1d weight [Cout, Cin/g, k] -> [g, Cout/g, k*Cin/g]."""
    co, cig, k = w.shape
    return w.reshape(groups, co // groups, cig, k).permute(0, 1, 3, 2).reshape(groups, co // groups, k * cig).contiguous()


def pack_convT_weight(w: torch.Tensor, u: int) -> torch.Tensor:
    """ConvTranspose1d weight [Cin, Cout, k] (stride u, k % u == 0) -> [u*Cout, (k/u)*Cin] with
    row p*Cout+co, column r*Cin+c holding w[c, co, p + r*u]: output phase p of input step q sums
    x[q-r] . w[:, co, p + r*u] over the k/u taps r."""
    cin, cout, k = w.shape
    assert k % u == 0, "transposed conv kernel must be a multiple of its stride"
    r = k // u
    return w.reshape(cin, cout, r, u).permute(3, 1, 2, 0).reshape(u * cout, r * cin).contiguous()


def gemm_mode() -> str:
    """KNNSVC_GEMM = f16x2 (default) | bf16x3 | fp32: how conv_gemm evaluates fp32 products for weights that
    went through attach_split (all three keep fp32-GEMM accuracy; see gemm2_core.h / gemm3_core.h)."""
    import os
    mode = os.environ.get("KNNSVC_GEMM", "f16x2")
    if mode not in ("f16x2", "bf16x3", "fp32"):
        raise KnnSvcError(f"KNNSVC_GEMM={mode!r}: expected f16x2, bf16x3 or fp32")
    return mode


def attach_split(w: torch.Tensor) -> torch.Tensor:
    """Pre-split a packed weight matrix [..., K] (K % 32 == 0) for the emulated-fp32 matrix-core kernels and
    hang the result on the tensor: ``w._w2`` (+ ``w._w2_scale``) = two fp16 planes of scale*w with the
    per-tensor power-of-two scale that puts max|w| in [2^13, 2^14) (default), or ``w._w3`` = three bf16
    planes (KNNSVC_GEMM=bf16x3).  No-op when the shape does not qualify or KNNSVC_GEMM=fp32 is set."""
    mode = gemm_mode()
    if mode == "fp32" or not w.is_cuda or w.dtype != torch.float32:
        return w
    K = w.shape[-1]
    if K % 32 != 0 or not w.is_contiguous():
        return w
    rows = w.numel() // K
    if mode == "bf16x3":
        out = torch.empty(rows * (K // 32) * 96, device=w.device, dtype=torch.int16)
        check(_lib.load().knnsvc_split_weight_bf16x3(_p(w), rows, K, _p(out), _stream()), "split_weight")
        w._w3 = out
        return w
    wmax = float(w.abs().max())
    if not math.isfinite(wmax):
        raise KnnSvcError("attach_split: non-finite weight")
    scale = 2.0 ** (13 - math.frexp(wmax)[1] + 1) if wmax > 0 else 1.0       # max|w| * scale in [2^13, 2^14)
    scale = min(max(scale, 2.0 ** -60), 2.0 ** 60)
    out = torch.empty(rows * (K // 32) * 64, device=w.device, dtype=torch.int16)
    check(_lib.load().knnsvc_split_weight_f16x2(_p(w), rows, K, scale, _p(out), _stream()), "split_weight")
    w._w2, w._w2_scale = out, scale
    return w


_GRAPH_READY = set()


def prepare_graph_capture(device) -> None:
    """torch registers per-generator graph-state tensors at the first ``capture_begin`` on a device.  If that first
    capture happens under ``torch.inference_mode()`` (KNeighborsVC's methods are decorated with it) they become
    inference tensors, and every later capture outside inference mode dies with "Inplace update to inference tensor".
    Doing one trivial capture in normal mode first makes the capture sites order-independent."""
    idx = torch.device(device).index
    idx = torch.cuda.current_device() if idx is None else idx
    if idx in _GRAPH_READY:
        return
    with torch.inference_mode(False):
        t = torch.zeros(1, device=torch.device("cuda", idx))
        torch.cuda.synchronize(idx)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            t.add_(1.0)
        del g
    _GRAPH_READY.add(idx)


_CAPTURE_STREAMS = {}


def capture_graph(fn, device, pool=None):
    """Capture ``fn()`` (which may only enqueue kernels on the current stream and allocate torch tensors) into a hipGraph
    WITHOUT synchronising the device: ``torch.cuda.graph`` calls ``torch.cuda.synchronize()`` on entry, which would stall a
    stream pipeline that happens to meet a new shape.  Capture runs on a private side stream ordered behind the caller's
    stream.  -> (graph, fn's return value: the graph's static outputs)."""
    prepare_graph_capture(device)
    idx = torch.device(device).index
    idx = torch.cuda.current_device() if idx is None else idx
    side = _CAPTURE_STREAMS.get(idx)
    if side is None:
        from .pipeline import new_stream
        side = _CAPTURE_STREAMS[idx] = new_stream(idx, kind="capture")
    cur = torch.cuda.current_stream(idx)
    side.wait_stream(cur)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        if pool is not None:
            g.capture_begin(pool=pool, capture_error_mode="thread_local")
        else:
            g.capture_begin(capture_error_mode="thread_local")
        try:
            out = fn()
        finally:
            g.capture_end()
    cur.wait_stream(side)
    return g, out


# ------------------------------------------------------------------ implicit-GEMM convolution
def last_conv_kernel() -> str:
    """Tag of the kernel the last conv_gemm of this thread dispatched to (measurement hook)."""
    return _lib.load().knnsvc_conv_gemm_last_kernel().decode()


def last_conv_epilogue() -> str:
    """"patch" / "lane" / "": the epilogue the last conv_gemm launch of this thread took (test hook)."""
    return _lib.load().knnsvc_conv_gemm_last_epilogue().decode()


def conv_gemm(x, w, out, *, m, n, cin, taps=1, stride=1, dil=1, pad=0, t_in=None, ldx=None, ldo=None,
              bias=None, bias_period=0, act=ACT_NONE, act_slope=0.0, a_slope=1.0, resid=None, ldr=None,
              accumulate=False, div=1.0, batches=1, groups=1, x_bstride=0, x_gstride=0, w_gstride=0,
              bias_gstride=0, o_bstride=0, o_gstride=0, r_bstride=0, r_gstride=0,
              convt_u=0, convt_cout=0, convt_pad=0, t_out=0, a_scale=0.0, x_split=False, out_split=False,
              w2=None, w2_scale=0.0, x_absmax=None, w_absmax=None, out_absmax=None, out_split_scale=0.0, dyn=None, x_bound=None,
              fixed_tile=False, defer=None):
    """See knnsvc_conv_gemm.  x/out/resid may be views into wider buffers (pass ldx/ldo/ldr).
    ``x_absmax`` / ``w_absmax`` / ``out_absmax``: one-element device tensors (range slots of the f16x2 path, see the
    header): bound of |x| / |w| the kernel derives its operand scales from, and where this launch folds max|out|.
    ``defer``: a list — the descriptor is appended to it instead of being launched (``conv_gemm_multi`` launches the list)."""
    lib = _lib.load()
    d = ConvDesc()
    d.x = x.data_ptr(); d.x_bstride = x_bstride; d.x_gstride = x_gstride
    d.ldx = ldx if ldx is not None else cin; d.t_in = t_in if t_in is not None else m
    d.cin = cin; d.taps = taps; d.stride = stride; d.dil = dil; d.pad = pad; d.a_slope = a_slope
    d.w = w.data_ptr(); d.w_gstride = w_gstride; d.n = n
    d.bias = bias.data_ptr() if bias is not None else None
    d.bias_gstride = bias_gstride; d.bias_period = bias_period
    d.out = out.data_ptr(); d.o_bstride = o_bstride; d.o_gstride = o_gstride
    d.ldo = ldo if ldo is not None else n; d.m = m
    d.act = act; d.act_slope = act_slope
    d.resid = resid.data_ptr() if resid is not None else None
    d.r_bstride = r_bstride; d.r_gstride = r_gstride; d.ldr = ldr if ldr is not None else (d.ldo if resid is not None else 0)
    d.accumulate = 1 if accumulate else 0; d.div = div
    d.batches = batches; d.groups = groups
    d.convt_u = convt_u; d.convt_cout = convt_cout; d.convt_pad = convt_pad; d.t_out = t_out
    w3 = getattr(w, "_w3", None)
    d.w_bf16x3 = w3.data_ptr() if w3 is not None else None
    if w2 is None:
        w2, w2_scale = getattr(w, "_w2", None), getattr(w, "_w2_scale", 0.0)
    d.w_f16x2 = w2.data_ptr() if w2 is not None else None
    d.w_f16x2_scale = w2_scale if w2 is not None else 0.0
    d.a_f16x2_scale = a_scale
    d.x_f16x2 = 1 if x_split else 0
    d.out_f16x2 = int(out_split) if (out_split is not True and out_split is not False) else (1 if out_split else 0)   # True / first split column
    if not x_split and (x_absmax is not None or out_absmax is not None) and not range_slots_on():
        x_absmax = out_absmax = None            # A/B aid: fixed activation scale 16 (a pre-split operand keeps its slot)
    if _SLOT_DBG and not x_split:          # timing aid: KNNSVC_SLOT_DBG=x keeps only the consumer side, =o only the producer side
        if _SLOT_DBG == "x": out_absmax = None
        if _SLOT_DBG == "o": x_absmax = None
    d.x_absmax = x_absmax.data_ptr() if x_absmax is not None else None
    d.w_absmax = w_absmax.data_ptr() if w_absmax is not None else None
    d.out_absmax = out_absmax.data_ptr() if out_absmax is not None else None
    d.out_f16x2_scale = out_split_scale
    d.fixed_tile = int(fixed_tile)          # False/0: by size, 1/True: the 128x128 kernel, 2: the 256x256 (Gemm2QuadS) kernel
    if x_bound is not None:            # (mul, add): |x| <= mul * max(x_absmax slot) + add — an input that was not measured itself
        d.x_bound_mul, d.x_bound_add = float(x_bound[0]), float(x_bound[1])
    if dyn is not None:
        # dyn = (device int32 count n, the bucket's count Nb): t_in / m / t_out of THIS call are what they are at n = Nb; each is
        # affine in the count with an offset in [0, Nb) (the generator's lengths: Nb * factor, + 1 or + taps - 1), recovered here
        n_t, nb = dyn
        d.n_dyn = n_t.data_ptr()
        d.dyn_t_in_mul, d.dyn_t_in_add = divmod(int(d.t_in), nb)
        d.dyn_m_mul, d.dyn_m_add = divmod(int(m), nb)
        d.dyn_t_out_mul = int(t_out) // nb if convt_u else 0
        if convt_u and int(t_out) % nb:
            raise KnnSvcError("conv_gemm: dynamic t_out must be a multiple of the bucket count")
    if defer is not None:
        defer.append(d)
        return out
    check(lib.knnsvc_conv_gemm(C.byref(d), _stream()), "conv_gemm")
    return out


def conv_gemm_multi(descs) -> None:
    """Launch the descriptors collected with ``conv_gemm(..., defer=descs)`` — convolutions of one output shape (the generator's
    three ResBlock branches of a step) — as ONE grid (knnsvc_conv_gemm_multi); same bits as separate launches."""
    if not descs:
        return
    for i in range(0, len(descs), 4):
        part = descs[i:i + 4]
        arr = (ConvDesc * len(part))(*part)
        check(_lib.load().knnsvc_conv_gemm_multi(arr, len(part), _stream()), "conv_gemm_multi")


def resblock_pair_multi(descs) -> None:
    """The same for ``resblock_pair(..., defer=descs)`` (knnsvc_resblock_pair_multi)."""
    if not descs:
        return
    for i in range(0, len(descs), 4):
        part = descs[i:i + 4]
        arr = (PairDesc * len(part))(*part)
        check(_lib.load().knnsvc_resblock_pair_multi(arr, len(part), _stream()), "resblock_pair_multi")


def absmax(x2d, slot=None):
    """max |x| of a [rows, cols] view (row stride allowed) folded into ``slot`` (a zeroed one-element device tensor is
    made when None): the range slot a later conv_gemm takes as x_absmax / w_absmax.  NaN in x makes the slot NaN."""
    _need(x2d, name="absmax.x")
    if not range_slots_on() and slot is not None:
        return slot
    if slot is None:
        slot = new_slot(x2d.device)
    if x2d.dim() == 1:
        x2d = x2d[None]
    if x2d.stride(1) != 1:
        raise KnnSvcError("absmax: rows must be contiguous")
    check(_lib.load().knnsvc_absmax(_p(x2d), x2d.shape[0], x2d.shape[1], x2d.stride(0) if x2d.shape[0] > 1 else x2d.shape[1],
                                    _p(slot), _stream()), "absmax")
    return slot


def resblock_pair_ok(channels: int, taps: int, dil: int) -> bool:
    """The fused ResBlock pair covers the generator's narrow stages (knnsvc_resblock_pair); KNNSVC_FUSED_PAIR=0 switches it off."""
    import os
    widths = (32, 64, 128) if os.environ.get("KNNSVC_FUSED_PAIR_128", "0") == "1" else (32, 64)
    return (os.environ.get("KNNSVC_FUSED_PAIR", "1") != "0" and gemm_mode() == "f16x2" and channels in widths and
            taps % 2 == 1 and taps <= 11 and dil * (taps - 1) <= 64)


def resblock_pair(x, w1, b1, w2, b2, out, *, t, channels, taps, dil, slope, x_absmax, t1_bound, out_absmax=None, dyn=None,
                  ldx=None, ldo=None, defer=None):
    """out = conv1d(lrelu(conv1d(lrelu(x), w1, dilation=dil) + b1), w2) + b2 + x in one launch (see the header); w1 / w2 are packed
    conv weights carrying their f16x2 split (attach_split).  Bit-identical to the two conv_gemm launches."""
    d = PairDesc()
    d.x = x.data_ptr(); d.ldx = ldx if ldx is not None else channels; d.t = t; d.channels = channels; d.taps = taps; d.dil = dil
    d.w1_f16x2 = w1._w2.data_ptr(); d.w1_scale = w1._w2_scale; d.b1 = b1.data_ptr() if b1 is not None else None
    d.w2_f16x2 = w2._w2.data_ptr(); d.w2_scale = w2._w2_scale; d.b2 = b2.data_ptr() if b2 is not None else None
    d.out = out.data_ptr(); d.ldo = ldo if ldo is not None else channels
    d.slope = slope
    if range_slots_on() and x_absmax is not None:
        d.x_absmax = x_absmax.data_ptr(); d.t1_bound_mul, d.t1_bound_add = float(t1_bound[0]), float(t1_bound[1])
        d.out_absmax = out_absmax.data_ptr() if out_absmax is not None else None
    else:                                   # A/B aid (KNNSVC_RANGE_SLOTS=0): the fixed activation scale 16
        d.x_absmax = None; d.a1_scale = d.a2_scale = 16.0; d.out_absmax = None
    if dyn is not None:
        d.n_dyn = dyn[0].data_ptr(); d.dyn_mul = t // dyn[1]
    if defer is not None:
        defer.append(d)
        return out
    check(_lib.load().knnsvc_resblock_pair(C.byref(d), _stream()), "resblock_pair")
    return out


def axpy(x, alpha, out, accumulate):
    """out = alpha * x (+ out): one term of a general WavLM layer weighting."""
    check(_lib.load().knnsvc_axpy(_p(x), x.numel(), float(alpha), 1 if accumulate else 0, _p(out), _stream()), "axpy")
    return out


def mean3(a, b, c, div, out, out_absmax=None, dyn=None, row_floats=0):
    """out = (c + (b + a)) / div over contiguous tensors; ``dyn`` = (device frame count, bucket frames): only the first
    count * (numel / bucket frames) floats are touched."""
    n = a.numel()
    nd, mul = (None, 0)
    if dyn is not None:
        nd, mul = dyn[0], n // dyn[1]
    if not range_slots_on():
        out_absmax = None
    check(_lib.load().knnsvc_mean3(_p(a), _p(b), _p(c), n, float(div), _p(out), _p(out_absmax), _p(nd), mul, _stream()), "mean3")
    return out


SLOT_W = 64 * 32     # a range slot is 64 stripes of one 128-byte cache line each (float 32 i = stripe i): include/knnsvc_hip.h


def new_slot(device) -> torch.Tensor:
    return torch.zeros(SLOT_W, device=device, dtype=torch.float32)


def pick_scale(bound: float) -> float:
    """Host mirror of the kernels' kn_pick_scale: the largest power of two s with bound * s < 2^15."""
    if not (bound > 0.0) or not math.isfinite(bound):
        return 2.0 ** 54 if bound == 0.0 else 2.0 ** -114
    return 2.0 ** (14 - max(math.frexp(bound)[1] - 1, -40))


def linear(x2d, w, bias=None, act=ACT_NONE, resid=None, out=None, x_split=False, out_split=False, **kw):
    """out[M,N] = act(x2d[M,K] @ w[N,K]^T + bias) (+ resid).  x_split / out_split: operand / result in the f16x2
    split layout (include/knnsvc_hip.h, "A2"), carried in float32 tensors of the usual shape; out_split may also be
    the first split column (a multiple of 32): columns before it stay fp32."""
    _need(x2d, name="linear.x"); _need(w, name="linear.w")
    M, K = x2d.shape
    N = w.shape[0]
    if out is None:
        out = torch.empty(M, N, device=x2d.device, dtype=torch.float32)
    return conv_gemm(x2d, w, out, m=M, n=N, cin=K, bias=bias, act=act, resid=resid, x_split=x_split, out_split=out_split, **kw)


def split_pack(x2d: torch.Tensor) -> torch.Tensor:
    """fp32 [rows, C] (C % 32 == 0) -> the f16x2 split layout in a float32 tensor of the same shape (torch ops;
    for tests and debugging — the kernels write this layout themselves)."""
    rows, C = x2d.shape
    xs = x2d.float() * 16.0
    hi = xs.half()
    lo = (xs - hi.float()).half()
    packed = torch.stack([hi.view(rows, C // 32, 32), lo.view(rows, C // 32, 32)], 2).contiguous()   # [rows, C/32, 2, 32]
    return packed.view(torch.float32).view(rows, C)


def split_unpack(t2d: torch.Tensor) -> torch.Tensor:
    """inverse of split_pack (up to the 2^-22 relative split error)."""
    rows, C = t2d.shape
    h = t2d.contiguous().view(torch.float16).view(rows, C // 32, 2, 32)
    return ((h[:, :, 0].float() + h[:, :, 1].float()) / 16.0).reshape(rows, C)


# ------------------------------------------------------------------ WavLM pieces
def layernorm(x2d, gamma, beta, gelu=False, out=None, out_split=False):
    """out_split: write the f16x2 split layout (the result then only makes sense as a GEMM's x_split operand)."""
    _need(x2d, name="layernorm.x")
    rows, dim = x2d.shape
    if out is None:
        out = torch.empty_like(x2d)
    check(_lib.load().knnsvc_layernorm(_p(x2d), rows, dim, x2d.stride(0), _p(gamma), _p(beta), (1 if gelu else 0) | (2 if out_split else 0),
                                       _p(out), out.stride(0), _stream()), "layernorm")
    return out


def wavlm_conv0(x, w, gamma, beta, k, stride, out_split=False):
    """[B, L] waveform -> [B*T, C] = GELU(LN(conv1d(x))) of the first feature-extractor layer, one kernel."""
    B, L = x.shape
    C_ = gamma.numel()
    T = (L - k) // stride + 1
    out = torch.empty(B * T, C_, device=x.device, dtype=torch.float32)
    check(_lib.load().knnsvc_wavlm_conv0(_p(x), B, L, _p(w), C_, k, stride, _p(gamma), _p(beta), _p(out),
                                         1 if out_split else 0, _stream()), "wavlm_conv0")
    return out


def wavlm_gate(xn2d, heads, w2, b2, grep_a, x_split=False):
    rows = xn2d.shape[0]
    gate = torch.empty(rows, heads, device=xn2d.device, dtype=torch.float32)
    check(_lib.load().knnsvc_wavlm_gate(_p(xn2d), rows, heads, 64, xn2d.stride(0), _p(w2), _p(b2), _p(grep_a),
                                        _p(gate), 1 if x_split else 0, _stream()), "wavlm_gate")
    return gate


def attention_mode() -> str:
    """KNNSVC_ATTENTION = f16x2 (default) | bf16x3 | fp32, as knnsvc_wavlm_attention reads it."""
    import os
    e = os.environ.get("KNNSVC_ATTENTION", "")
    return "bf16x3" if e[:1] == "b" else "fp32" if e[:2] == "fp" else "f16x2"


def mask_rows(x2d, batches, T, lens):
    """Rows t >= lens[b] of the [batches*T, dim] activation become zero in place (``lens``: int32 device tensor)."""
    _need(x2d, name="mask_rows.x"); _need(lens, torch.int32, "mask_rows.lens")
    check(_lib.load().knnsvc_mask_rows(_p(x2d), batches, T, x2d.shape[1], x2d.stride(0), _p(lens), _stream()), "mask_rows")
    return x2d


def wavlm_attention(qkv, gate, table, batches, T, heads, out_split=False, kv_split=False, wide=False, kv_len=None):
    """kv_split: the K and V column blocks of qkv hold the f16x2 split layout (QKV projection run with out_split=E).
    wide: Q / K / V may exceed the f16x2 kernel's fixed-scale range (decided at load from the weights,
    WavLMEncoder._range_plan): run the bf16x3 kernel (fp32 exponent range) whatever KNNSVC_ATTENTION says."""
    if wide and (out_split or kv_split):
        raise KnnSvcError("wavlm_attention: wide-range mode has fp32 inputs and outputs")
    out = torch.empty(batches * T, heads * 64, device=qkv.device, dtype=torch.float32)
    check(_lib.load().knnsvc_wavlm_attention(_p(qkv), _p(gate), _p(table), batches, T, heads, _p(out),
                                             (1 if out_split else 0) | (4 if wide else 0), 1 if kv_split else 0, _p(kv_len),
                                             _stream()), "wavlm_attention")
    return out


# ------------------------------------------------------------------ kNN
import os as _os
_SLOT_DBG = _os.environ.get("KNNSVC_SLOT_DBG", "")


def range_slots_on() -> bool:
    """KNNSVC_RANGE_SLOTS=0 (A/B aid): f16x2 GEMMs fall back to the fixed activation scale 16 (|x| < 4094)."""
    import os
    return os.environ.get("KNNSVC_RANGE_SLOTS", "1") != "0"


def row_norms(x2d, slot=None):
    """-> (norm [rows], sumsq [rows]).  ``norm._slot``: one-element device tensor holding the largest row norm — an
    upper bound of max|x| the kernel folds on the way (the kNN's range slot; no extra pass over the features).
    ``slot``: a ZEROED range slot to fold into (a search zeroes all its small buffers with one fill: knn_topk)."""
    _need(x2d, name="row_norms.x")
    rows, dim = x2d.shape
    norm = torch.empty(rows, device=x2d.device, dtype=torch.float32)
    sq = torch.empty(rows, device=x2d.device, dtype=torch.float32)
    slot = new_slot(x2d.device) if slot is None else slot
    check(_lib.load().knnsvc_row_norms(_p(x2d), rows, dim, x2d.stride(0), _p(norm), _p(sq), _p(slot), _stream()), "row_norms")
    norm._slot = slot
    return norm, sq


def knn_mode() -> str:
    """KNNSVC_KNN = f16x2 | fp32.  f16x2 (default while KNNSVC_GEMM is f16x2): q.p^T from the emulated-fp32 GEMM +
    knnsvc_knn_select; fp32: the fused exact-fp32-MFMA tile kernel (knnsvc_knn_topk)."""
    import os
    mode = os.environ.get("KNNSVC_KNN", "f16x2" if gemm_mode() == "f16x2" else "fp32")
    if mode not in ("f16x2", "fp32"):
        raise KnnSvcError(f"KNNSVC_KNN={mode!r}: expected f16x2 or fp32")
    return mode


def prepare_knn_pool(pool, k=32, p_stats=None):
    """Pre-split image of a pool for the two-kernel kNN route, reusable across searches against the same pool
    (dataset mode and prematch search one pool once per utterance): list of (first row, rows view, f16x2 image, range
    slot).  The fp16 split needs a power-of-two pre-scale that fits the features' range; WavLM features have no a-priori
    bound (outlier channels), so the scale is derived ON THE DEVICE from max|pool| (knnsvc_absmax -> knnsvc_split_f16x2_dyn;
    the GEMM reads the same slot as w_absmax) — no host round trip, no fixed range.
    Returns None when the route does not apply (see knn_topk)."""
    _need(pool, name="knn.pool")
    npool, dim = pool.shape
    if not (knn_mode() == "f16x2" and dim % 32 == 0 and npool >= k and pool.is_contiguous()):
        return None
    lib = _lib.load()
    p_cap = ((1 << 30) - 1) // (dim * 4) // 128 * 128
    n_chunks = -(-npool // p_cap)
    p_rows = -(-npool // n_chunks)                    # balanced chunks: no tail shorter than k
    chunks = []
    for p0 in range(0, npool, p_rows):
        pc = pool[p0:p0 + p_rows]
        npc = pc.shape[0]
        if npc < k:
            raise KnnSvcError("knn_topk: pool chunk smaller than k")
        slot = getattr(p_stats[0], "_slot", None) if p_stats is not None else None      # bound of the whole pool: fine for a chunk
        if slot is None:
            slot = absmax(pc)
        # rows padded to a multiple of 4 (zero images): the dot-matrix GEMM runs on the 256x256 kernel, whose 16-byte epilogue
        # wants n % 4 == 0 — the padded columns' dots are never looked at (knn_select gets the true row count)
        npad = -(-npc // 4) * 4
        p2 = torch.empty(npad * (dim // 32) * 64, device=pool.device, dtype=torch.int16)
        if npad != npc:
            p2[npc * (dim // 32) * 64:].zero_()
        check(lib.knnsvc_split_f16x2_dyn(_p(pc), npc, dim, _p(slot), _p(p2), _stream()), "split_pool")
        chunks.append((p0, pc, p2, slot))
    return chunks


def knn_rescore_on() -> bool:
    """KNNSVC_KNN_RESCORE=0 (A/B aid): the lists keep the order of the screening distances (round 4's behaviour) instead of
    being re-scored from exact dot products (knnsvc_knn_rescore)."""
    import os
    return os.environ.get("KNNSVC_KNN_RESCORE", "1") != "0"


KNN_WIDE = 64                     # keys per row of a wide list (include/knnsvc_hip.h, knnsvc_knn_rescore)


def _knn_rescore(wide, q, qn, qs, pc, pn, ps, k, idx_offset, mask, idx_out, dist_out):
    """wide [m, 64] int64 keys -> exact top-k (idx_out / dist_out [m, k])."""
    m, dim = q.shape
    check(_lib.load().knnsvc_knn_rescore(_p(wide), m, k, _p(q), _p(qn), _p(qs), _p(pc), _p(pn), _p(ps), pc.shape[0], dim, idx_offset,
                                         mask[0], mask[1], 1 if knn_rescore_on() else 0, _p(idx_out), _p(dist_out), _stream()), "knn_rescore")


def _knn_topk_gemm(q, pool, k, idx_offset, qn, qs, pn, ps, flag, mask=(0, 0), prepared=None, allow_fused=True, max_blocks=0, zeros=None):
    """q.p^T on the f16x2 matrix-core loop (Gemm2QuadS), then the reference's distance formula + selection.  Two routes over
    the SAME dot-product bits (both sum over K in the 16x16x32 grouping, whatever the sizes):
      * fused (default from a few hundred query rows on): knnsvc_knn_screen + knnsvc_knn_refine, no dot matrix (_knn_fused_chunk);
      * dot matrix: knnsvc_conv_gemm (fixed_tile = 2, pool rows = pre-split "weights") + knnsvc_knn_select.
    Both operands are scaled by device-chosen powers of two (range slots, see prepare_knn_pool) — exact, so the dots do not
    depend on the scale.  Pool and query are chunked so that every buffer resource stays below 1 GiB and the dot matrix below
    ~1 GiB; pool chunks are folded with knnsvc_knn_merge.  ``flag``: bit 0 = NaN distance, bit 1 = the fused route's candidate
    buffer overflowed (results of that call are then NOT valid: knn_topk / raise_if_nan turn it into a retry on the dot-matrix
    route) — read by the caller, never here: no host synchronisation inside a search."""
    lib = _lib.load()
    nq, dim = q.shape
    dev = q.device
    q_rows_cap = ((1 << 30) - 1) // (dim * 4) // 128 * 128
    parts_i, parts_d = [], []
    # the queries are split once into the A2 layout (the weight-split kernel writes exactly that image)
    q_slot = getattr(qn, "_slot", None)
    if q_slot is None:
        q_slot = absmax(q)
    q2 = torch.empty(nq, dim, device=dev, dtype=torch.float32)
    check(lib.knnsvc_split_f16x2_dyn(_p(q), nq, dim, _p(q_slot), _p(q2), _stream()), "split_queries")
    fused = allow_fused and knn_fused_on() and nq >= KNN_FUSED_MIN_Q
    for p0, pc, p2, p_slot in (prepared if prepared is not None else prepare_knn_pool(pool, k, (pn, ps))):
        npc = pc.shape[0]
        idx = torch.empty(nq, k, device=dev, dtype=torch.int64)
        dist = torch.empty(nq, k, device=dev, dtype=torch.float32)
        KNN_ROUTE_COUNTS["chunks"] += 1
        if fused and npc >= KNN_FUSED_MIN_P:
            KNN_ROUTE_COUNTS["fused"] += 1
            _knn_fused_chunk(q, q2, q_slot, qn, qs, pc, p2, p_slot, pn[p0:], ps[p0:], k, idx_offset + p0,
                             (mask[0] - p0, mask[1] - p0), idx, dist, flag, max_blocks, zeros)
        else:
            KNN_ROUTE_COUNTS["dot"] += 1
            npad = -(-npc // 4) * 4
            q_rows = max(128, min(nq, q_rows_cap, (1 << 28) // max(npad, 1) // 128 * 128))
            for q0 in range(0, nq, q_rows):
                qc = q2[q0:q0 + q_rows]
                m = qc.shape[0]
                dots = torch.empty(m, npad, device=dev, dtype=torch.float32)
                conv_gemm(qc, pc, dots, m=m, n=npad, cin=dim, w2=p2, x_split=True, x_absmax=q_slot, w_absmax=p_slot,
                          fixed_tile=2)         # a shard / a query set of any size gives the whole search's distance bits
                wide = torch.empty(m, KNN_WIDE, device=dev, dtype=torch.int64)
                check(lib.knnsvc_knn_select(_p(dots), npad, _p(qn[q0:]), _p(qs[q0:]), m, _p(pn[p0:]), _p(ps[p0:]), npc, k,
                                            mask[0] - p0, mask[1] - p0, _p(wide), _p(flag), _stream()), "knn_select")
                _knn_rescore(wide, q[q0:q0 + m], qn[q0:], qs[q0:], pc, pn[p0:], ps[p0:], k, idx_offset + p0,
                             (mask[0] - p0, mask[1] - p0), idx[q0:], dist[q0:])
        parts_i.append(idx); parts_d.append(dist)
    if len(parts_i) == 1:
        return parts_i[0], parts_d[0]
    return knn_merge(torch.stack(parts_d), torch.stack(parts_i))


KNN_ROUTE_COUNTS = {"chunks": 0, "fused": 0, "dot": 0}     # which route each (search, pool chunk) took: tests assert on it
# The fused route (epochs of knnsvc_knn_screen + knnsvc_knn_refine, no dot matrix) from 256 query frames x 8192 pool rows on.
# Round 3 (threshold from a separate sample pass: ~0.15 ms whatever the size) had 2048 / 8192; round 2 4096 / 32768.
KNN_FUSED_MIN_Q = int(_os.environ.get("KNNSVC_KNN_FUSED_MIN_Q", "256"))
KNN_FUSED_MIN_P = int(_os.environ.get("KNNSVC_KNN_FUSED_MIN_P", "8192"))
KNN_FUSED_CAP = 4096
KNN_OVERFLOW = 2                  # flag bit: the fused route's candidate buffer overflowed
KNN_DEBUG_COUNTS = None
KNN_EPOCH_GROWTH = int(_os.environ.get("KNNSVC_KNN_EPOCH_GROWTH", "4"))
KNN_COLD_TILES_MAX = 44           # column tiles of the first epoch: ~45 candidates per (row, tile) have to fit the buffer twice over


def knn_fused_on() -> bool:
    import os
    return os.environ.get("KNNSVC_KNN_FUSED", "1") != "0" and not getattr(_FUSED_OFF, "on", False)


import threading as _threading
_FUSED_OFF = _threading.local()      # per host thread (the request-queue worker's retry must not reroute the main thread's searches)


class fused_off:
    """Context: every search inside takes the dot-matrix route (the retry after a candidate-buffer overflow)."""
    def __enter__(self):
        self.prev = getattr(_FUSED_OFF, "on", False); _FUSED_OFF.on = True
    def __exit__(self, *a):
        _FUSED_OFF.on = self.prev


def knn_epochs(nq: int, npc: int, blocks: int = 256):
    """Column-tile ranges [(t0, t1), ...] (tiles of 256 pool rows) of the fused route's epochs.  The first epoch has no
    thresholds (every tile bounds its rows itself and lets ~45 of its 256 columns per row through), so it is about ONE round
    of workgroups — enough columns for a useful k-th distance (at least 4 tiles), few enough for the candidate buffer; every
    later epoch multiplies the rows seen by KNN_EPOCH_GROWTH: a row's survivors per epoch are ~k x (new rows / rows seen), and
    each epoch boundary costs one refine launch and the tail of a round.  A last epoch much smaller than its predecessor is
    merged into it."""
    gx, gy = -(-nq // 256), -(-npc // 256)
    e0 = min(gy, max(4, min(KNN_COLD_TILES_MAX, max(1, blocks) // gx)))
    out, c = [(0, e0)], e0
    while c < gy:
        nxt = min(gy, c * KNN_EPOCH_GROWTH)
        if gy - nxt < (nxt - c) // 2:
            nxt = gy
        out.append((c, nxt)); c = nxt
    return out


def _knn_fused_chunk(q, q2, q_slot, qn, qs, pc, p2, p_slot, pn, ps, k, idx_offset, mask, idx_out, dist_out, flag, max_blocks=0, zeros=None):
    """One pool chunk without a [nq, np] dot matrix: the chunk's rows in epochs (knn_epochs), each one knnsvc_knn_screen (the
    first without thresholds, the later ones against the row's k-th key so far) + one knnsvc_knn_refine (list so far + the new
    candidates -> list, next thresholds).  A row with more survivors than the candidate buffer holds (pathological data: e.g.
    thousands of bit-identical pool rows inside the first epoch) sets bit 1 of ``flag``; nothing here reads it."""
    lib = _lib.load()
    nq, dim = q.shape
    npc = pc.shape[0]
    dev = q.device
    row_u16 = (dim // 32) * 64                                     # int16 elements of one pool row in the split image
    blocks = int(max_blocks) if max_blocks else 256
    q_rows = max(256, min(nq, ((1 << 28) - 1) // dim // 256 * 256, (1 << 30) // (KNN_FUSED_CAP * 8) // 256 * 256))
    for q0 in range(0, nq, q_rows):
        m = min(q_rows, nq - q0)
        epochs = knn_epochs(m, npc, blocks)
        # one fill: candidate counts [m] | the first epoch's workspace: row bounds [m], per-half-tile bounds [2 gy0][m] (the tiles'
        # exchange, knn.hip), arrival counters [gx] one cache line each
        gy0 = epochs[0][1] - epochs[0][0]
        nz = m * (2 + 2 * gy0) + 32 * -(-m // 256)
        if zeros is not None and zeros.numel() == nz and q0 == 0:
            zeroed = zeros                              # part of the search's ONE fill (knn_topk)
        else:
            zeroed = torch.zeros(nz, device=dev, dtype=torch.int32)
        cnt, bound = zeroed[:m], zeroed[m:]
        cand = torch.empty(m * KNN_FUSED_CAP * 2, device=dev, dtype=torch.int32)
        wide = torch.empty(m, KNN_WIDE, device=dev, dtype=torch.int64)       # the rows' wide lists, handed from epoch to epoch
        thr = torch.empty(m, device=dev, dtype=torch.float32)
        thr_idx = torch.empty(m, device=dev, dtype=torch.int64)
        for e, (t0, t1) in enumerate(epochs):
            c0, c1 = t0 * 256, min(t1 * 256, npc)
            last = e == len(epochs) - 1
            # both kernels OR their bits straight into the search's flag (bit 0: NaN distance, bit 1: candidate-buffer overflow
            # or a short list): no glue kernels between the launches
            check(lib.knnsvc_knn_screen(_p(q2[q0:]), _p(q_slot), _p(qn[q0:]), _p(qs[q0:]), m, _p(p2[c0 * row_u16:]), _p(p_slot), _p(pn[c0:]),
                                        _p(ps[c0:]), c1 - c0, dim, _p(thr) if e else _p(None), _p(thr_idx) if e else _p(None), mask[0], mask[1],
                                        c0, _p(cnt), _p(cand), KNN_FUSED_CAP, _p(None) if e else _p(bound), _p(flag), int(max_blocks), _stream()), "knn_screen")
            if KNN_DEBUG_COUNTS is not None:           # tools/knn_prof.py: survivors per row and epoch (a clone: refine resets the counts)
                KNN_DEBUG_COUNTS.append((e, cnt.clone()))
            check(lib.knnsvc_knn_refine(_p(cnt), _p(cand), KNN_FUSED_CAP, m, k, _p(None) if e else _p(bound), 1 if e else 0, _p(wide),
                                        _p(None) if last else _p(thr), _p(None) if last else _p(thr_idx), 1 if last else 0, _p(flag),
                                        _stream()), "knn_refine")
        # the candidates' order came from the screening products; the order that goes out comes from exact ones
        _knn_rescore(wide, q[q0:q0 + m], qn[q0:], qs[q0:], pc, pn, ps, k, idx_offset, mask, idx_out[q0:], dist_out[q0:])


def knn_topk(q, pool, k=32, idx_offset=0, q_stats=None, p_stats=None, check_nan=True, return_flag=False, mask=None,
             prepared=None, max_blocks=0):
    """Ascending cosine-distance top-k of each q row among pool rows -> (idx int64 [nq,k], dist f32 [nq,k]).
    ``mask`` = (lo, hi): pool rows [lo, hi) compete at distance exactly 1 (self-matching of per_spk_extract,
    ddsp_prematch_dataset.py:1606-1607).
    ``check_nan=True``: the flag is read here (one sync); a NaN raises, a candidate-buffer overflow of the fused route repeats the
    search on the dot-matrix route.  ``check_nan=False, return_flag=True``: nothing is read; the caller hands the flag to
    ``raise_if_nan`` once everything is enqueued (KnnOverflow -> it repeats the work with ``fused_off()``).
    ``max_blocks``: grid cap of the persistent screening kernel (0 = the whole chip)."""
    mask = (0, 0) if mask is None else (int(mask[0]), int(mask[1]))
    _need(q, name="knn.q"); _need(pool, name="knn.pool")
    if not (q.is_contiguous() and pool.is_contiguous()):
        raise KnnSvcError("knn_topk: q and pool must be contiguous")
    lib = _lib.load()
    nq, dim = q.shape
    npool = pool.shape[0]
    f16 = knn_mode() == "f16x2" and dim % 32 == 0 and npool >= k and nq > 0 and 1 <= k <= 32
    # ONE fill for everything a search needs zeroed — its flag, the two operands' range slots and (fused route, one chunk) the
    # candidate counts + first-epoch workspace: four tiny fill launches were ~5 us each in front of a 0.4 ms search
    flag = zeros = None
    if f16:
        nz = 0
        if (prepared is None and knn_fused_on() and nq >= KNN_FUSED_MIN_Q and npool >= KNN_FUSED_MIN_P and
                npool * dim * 4 < (1 << 30) and nq * dim < (1 << 28) and nq * KNN_FUSED_CAP * 8 <= (1 << 30)):
            ep = knn_epochs(nq, npool, int(max_blocks) if max_blocks else 256)
            nz = nq * (2 + 2 * (ep[0][1] - ep[0][0])) + 32 * -(-nq // 256)
        zbuf = torch.zeros(64 + 2 * SLOT_W + nz, device=q.device, dtype=torch.int32)
        flag = zbuf[:1]
        sl_q, sl_p = zbuf[64:64 + SLOT_W].view(torch.float32), zbuf[64 + SLOT_W:64 + 2 * SLOT_W].view(torch.float32)
        zeros = zbuf[64 + 2 * SLOT_W:] if nz else None
    qn, qs = q_stats if q_stats is not None else row_norms(q, sl_q if f16 else None)
    pn, ps = p_stats if p_stats is not None else row_norms(pool, sl_p if f16 else None)
    if f16:
        idx, dist = _knn_topk_gemm(q, pool, k, idx_offset, qn, qs, pn, ps, flag, mask, prepared, max_blocks=max_blocks, zeros=zeros)
        if check_nan:
            try:
                raise_if_nan(flag)
            except KnnOverflow:
                flag.zero_()
                idx, dist = _knn_topk_gemm(q, pool, k, idx_offset, qn, qs, pn, ps, flag, mask, prepared, allow_fused=False)
                raise_if_nan(flag)
        return (idx, dist, flag) if return_flag else (idx, dist)
    ws_bytes = lib.knnsvc_knn_workspace_bytes(nq, npool, k)
    ws = torch.empty(max(ws_bytes, 8), device=q.device, dtype=torch.uint8)
    idx = torch.empty(nq, k, device=q.device, dtype=torch.int64)
    dist = torch.empty(nq, k, device=q.device, dtype=torch.float32)
    flag = torch.zeros(1, device=q.device, dtype=torch.int32)
    check(lib.knnsvc_knn_topk(_p(q), _p(qn), _p(qs), nq, _p(pool), _p(pn), _p(ps), npool, dim, k, idx_offset, mask[0], mask[1],
                              _p(idx), _p(dist), _p(ws), ws_bytes, _p(flag), 1 if knn_rescore_on() else 0, _stream()), "knn_topk")
    if check_nan:
        raise_if_nan(flag)
    return (idx, dist, flag) if return_flag else (idx, dist)


def retry_on_overflow(fn):
    """fn() with deferred flag checks inside; if one of them reports a candidate-buffer overflow of the fused kNN route, the
    whole of fn is repeated with every search on the dot-matrix route.  (fn must be repeatable: it recomputes its outputs.)"""
    try:
        return fn()
    except KnnOverflow:
        KNN_ROUTE_COUNTS["overflow_retries"] = KNN_ROUTE_COUNTS.get("overflow_retries", 0) + 1
        with fused_off():
            return fn()


class KnnOverflow(KnnSvcError):
    """The fused kNN route met a query row with more candidates than its buffer holds: the search has to be repeated on the
    dot-matrix route (``with ops.fused_off(): ...``).  Not an error of the data — many near-identical pool rows do it."""


def raise_if_nan(flag):
    """Host check of a search's device flag (one sync).  Bit 0: a NaN distance — the reference prints 'containing nan' and
    sys.exit()s inside fast_cosine_dist (lib_ongaku_test.py:166-169).  Bit 1: KnnOverflow.  Callers may defer this check to
    the end of a launch sequence so that it does not split the stream."""
    v = int(flag.item())
    if v & 1:
        raise KnnSvcError("containing nan")
    if v & KNN_OVERFLOW:
        raise KnnOverflow("fused kNN route: candidate buffer overflow (repeat with ops.fused_off())")


def reload_knobs() -> None:
    """Re-read the dispatcher's A/B switches (KNNSVC_QUAD, KNNSVC_QUAD_EPI, KNNSVC_EPILOGUE, KNNSVC_WIN*, KNNSVC_GEMM_SMALL) from
    the environment: the library reads them once, at its first launch (csrc/conv_gemm.hip: Knobs)."""
    check(_lib.load().knnsvc_reload_knobs(), "reload_knobs")


def knn_merge(part_dist, part_idx):
    """[parts, nq, k] per-shard lists (global indices) -> merged (idx, dist)."""
    parts, nq, k = part_dist.shape
    idx = torch.empty(nq, k, device=part_dist.device, dtype=torch.int64)
    dist = torch.empty(nq, k, device=part_dist.device, dtype=torch.float32)
    check(_lib.load().knnsvc_knn_merge(_p(part_dist.contiguous()), _p(part_idx.contiguous()), parts, nq, k,
                                       _p(idx), _p(dist), _stream()), "knn_merge")
    return idx, dist


# ------------------------------------------------------------------ neighbour post-processing
def log_f0_median(f0):
    """-> tensor [2] on device: (lower median of log f0 over voiced frames, voiced count)."""
    _need(f0, name="f0")
    res = torch.empty(2, device=f0.device, dtype=torch.float32)
    ws = torch.empty(f0.numel(), device=f0.device, dtype=torch.float32)
    check(_lib.load().knnsvc_log_f0_median(_p(f0), f0.numel(), _p(res), _p(ws), _stream()), "log_f0_median")
    return res


def shift_f0(f0, qmed, pmed):
    out = torch.empty_like(f0)
    check(_lib.load().knnsvc_shift_f0(_p(f0), f0.numel(), _p(qmed), _p(pmed), _p(out), _stream()), "shift_f0")
    return out


def f0_rerank(nn_idx, shifted_f0, pool_f0):
    _need(nn_idx, torch.int64, "nn_idx")
    out = torch.empty_like(nn_idx)
    nq, k = nn_idx.shape
    check(_lib.load().knnsvc_f0_rerank(_p(nn_idx), nq, k, _p(shifted_f0), _p(pool_f0), _p(out), _stream()), "f0_rerank")
    return out


def concat_reselect(idx4, q, q_norm, pool, p_norm, shifted_f0=None, pool_f0=None, concat_weight=0.2):
    _need(idx4, torch.int64, "idx4")
    idx4 = idx4.contiguous()
    out = torch.empty_like(idx4)
    use_f0 = shifted_f0 is not None
    check(_lib.load().knnsvc_concat_reselect(_p(idx4), _p(q), _p(q_norm), q.shape[0], _p(pool), _p(p_norm),
                                             pool.shape[0], q.shape[1], _p(shifted_f0), _p(pool_f0), 1 if use_f0 else 0,
                                             float(concat_weight), _p(out), _stream()), "concat_reselect")
    return out


ADAM_FORCED_ITERS = None     # measurement aid (bench.py "value_long_adam"): run exactly this many iterations per loop


def smooth_weights(idx4, pool, scale, max_iter=100000, return_iters=False, row_scale=None):
    """``row_scale`` [nq,4]: the amp_ratio of compute_weight_with_amp (ddsp_prematch_dataset.py:684-804)."""
    if ADAM_FORCED_ITERS:
        max_iter = -int(ADAM_FORCED_ITERS)
    _need(idx4, torch.int64, "idx4"); _need(pool, name="pool")
    if row_scale is not None:
        _need(row_scale, name="row_scale")
        if tuple(row_scale.shape) != tuple(idx4.shape) or not row_scale.is_contiguous():
            raise KnnSvcError("smooth_weights: row_scale must be a contiguous [nq,4] tensor")
    lib = _lib.load()
    idx4 = idx4.contiguous()
    nq = idx4.shape[0]
    npool, dim = pool.shape
    ws_bytes = lib.knnsvc_smooth_workspace_bytes(nq)
    ws = torch.empty(ws_bytes, device=pool.device, dtype=torch.uint8)
    w = torch.empty(nq, 4, device=pool.device, dtype=torch.float32)
    iters = torch.zeros(1, device=pool.device, dtype=torch.int32)
    check(lib.knnsvc_smooth_weights(_p(idx4), nq, _p(pool), npool, dim, pool.stride(0), float(scale), _p(row_scale), int(max_iter),
                                    _p(w), _p(iters), _p(ws), ws_bytes, _stream()), "smooth_weights")
    return (w, iters) if return_iters else w


def weighted_gather(idx4, w, pool):
    _need(idx4, torch.int64, "idx4")
    idx4 = idx4.contiguous()
    nq, k = idx4.shape
    dim = pool.shape[1]
    out = torch.empty(nq, dim, device=pool.device, dtype=torch.float32)
    check(_lib.load().knnsvc_weighted_gather(_p(idx4), _p(w), nq, k, _p(pool), dim, pool.stride(0), 0, _p(out),
                                             _stream()), "weighted_gather")
    return out


# ------------------------------------------------------------------ prematch helpers
def round_f16(x):
    """x.half().float() (ddsp_prematch_dataset.py:1509, 1561, 1592) as one pass."""
    _need(x, name="round_f16.x")
    x = x.contiguous()
    out = torch.empty_like(x)
    check(_lib.load().knnsvc_round_f16(_p(x), x.numel(), _p(out), _stream()), "round_f16")
    return out


def amp_ratio(spec_q, spec_pool, idx):
    """[nq,k]: L1(spec_q[t]) / (L1(spec_pool[idx[t,k]]) + 1e-5)  (ddsp_prematch_dataset.py:1657-1660)."""
    _need(spec_q, name="spec_q"); _need(spec_pool, name="spec_pool"); _need(idx, torch.int64, "idx")
    idx = idx.contiguous()
    nq, k = idx.shape
    if spec_q.shape[0] != nq or spec_q.shape[1] != spec_pool.shape[1] or spec_q.stride(1) != 1 or spec_pool.stride(1) != 1:
        raise KnnSvcError("amp_ratio: shape mismatch")
    out = torch.empty(nq, k, device=spec_q.device, dtype=torch.float32)
    check(_lib.load().knnsvc_amp_ratio(_p(spec_q), spec_q.stride(0), _p(spec_pool), spec_pool.stride(0), spec_pool.shape[0],
                                       _p(idx), nq, k, spec_q.shape[1], _p(out), _stream()), "amp_ratio")
    return out


# ------------------------------------------------------------------ f0 front end
def f0_harvest(wav_1d, sample_rate=16000, f0_floor=65.0, f0_ceil=1047.0, frame_period=20.0, zero_below=80.0, check_status=True):
    """Harvest f0 track exactly as the reference asks pyworld for it (ddsp_prematch_dataset.py:121-128):
    [L] fp32 at 16 kHz -> [int(1000 L / fs / frame_period) + 1] fp32, 0 = unvoiced, values below 80 Hz zeroed.  All stages
    on the GPU in fp64 (csrc/harvest.hip); check_status reads the overflow flag back (one host sync)."""
    import ctypes
    _need(wav_1d, name="f0_harvest.wav")
    wav_1d = wav_1d.contiguous()
    lib = _lib.load()
    n, nbytes = ctypes.c_int64(0), ctypes.c_int64(0)
    check(lib.knnsvc_f0_harvest_workspace(wav_1d.numel(), int(sample_rate), float(f0_floor), float(f0_ceil), float(frame_period),
                                          ctypes.byref(n), ctypes.byref(nbytes)), "f0_harvest_workspace")
    ws = torch.empty(nbytes.value, device=wav_1d.device, dtype=torch.uint8)
    out = torch.empty(n.value, device=wav_1d.device, dtype=torch.float32)
    status = torch.zeros(1, device=wav_1d.device, dtype=torch.int32)
    check(lib.knnsvc_f0_harvest(_p(wav_1d), wav_1d.numel(), int(sample_rate), float(f0_floor), float(f0_ceil), float(frame_period),
                                float(zero_below), _p(out), n.value, _p(ws), nbytes.value, _p(status), _stream()), "f0_harvest")
    if check_status:
        flags = int(status.item())
        if flags:
            raise RuntimeError(f"f0_harvest: a fixed-capacity list overflowed (flags {flags:#x}: 1 candidates per frame, "
                               "2 / 4 section storage); the track is incomplete")
    return out


# ------------------------------------------------------------------ side features + synth
def reflect_pad(x1d, pad, extra=0):
    """-> [n + 2 pad (+ extra zeros at the end)]"""
    out = torch.empty(x1d.numel() + 2 * pad + extra, device=x1d.device, dtype=torch.float32)
    if extra:
        out[-extra:].zero_()
    check(_lib.load().knnsvc_reflect_pad(_p(x1d), x1d.numel(), pad, _p(out), _stream()), "reflect_pad")
    return out


def reflect_pad_batch(x_flat, offs, pad, stride):
    """x_flat: signals back to back, offs: device int64 [B+1] -> [B, stride] (reflect-padded rows, zero tails)."""
    B = offs.numel() - 1
    out = torch.empty(B, stride, device=x_flat.device, dtype=torch.float32)
    check(_lib.load().knnsvc_reflect_pad_batch(_p(x_flat), _p(offs), B, pad, _p(out), stride, _stream()), "reflect_pad_batch")
    return out


def spec_harm(reim, bins, f0, n_harm=49):
    """DFT product [rows, 2 bins] + f0 [rows] -> (spec [rows, bins], harm [rows, n_harm]) in one pass."""
    rows = reim.shape[0]
    spec = torch.empty(rows, bins, device=reim.device, dtype=torch.float32)
    harm = torch.empty(rows, n_harm, device=reim.device, dtype=torch.float32)
    check(_lib.load().knnsvc_spec_harm(_p(reim), rows, bins, reim.stride(0), _p(f0), n_harm, _p(spec), _p(harm), _stream()), "spec_harm")
    return spec, harm


def complex_mag(reim, bins):
    rows = reim.shape[0]
    out = torch.empty(rows, bins, device=reim.device, dtype=torch.float32)
    check(_lib.load().knnsvc_complex_mag(_p(reim), rows, bins, reim.stride(0), _p(out), _stream()), "complex_mag")
    return out


def harmonic_amps(spec, f0, n_harm=49):
    T, bins = spec.shape
    out = torch.empty(T, n_harm, device=spec.device, dtype=torch.float32)
    check(_lib.load().knnsvc_harmonic_amps(_p(spec.contiguous()), _p(f0), T, bins, n_harm, _p(out), _stream()),
          "harmonic_amps")
    return out


def additive_synth(f0, amp, prenet_w, prenet_b, cond, ld_cond, *, hop=320, sr=16000, mode=0, want_exc=False, n_dyn=None):
    """f0 [N], amp [N,H] (None in sine mode) -> writes cond (a [N*hop, >=n_ch] view); returns exc or None."""
    N = f0.numel()
    n_ch = prenet_b.numel()
    H = amp.shape[1] if amp is not None else 0
    exc = torch.empty(N * hop, device=f0.device, dtype=torch.float32) if want_exc else None
    ph = torch.empty(N, device=f0.device, dtype=torch.float64)
    check(_lib.load().knnsvc_additive_synth(_p(f0), _p(amp), N, H, hop, sr, mode, _p(prenet_w), _p(prenet_b), n_ch,
                                            _p(cond), ld_cond, _p(exc), _p(ph), _p(n_dyn), _stream()), "additive_synth")
    return exc
