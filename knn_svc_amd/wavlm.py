"""Host side of the WavLM encoder: weight packing + the kernel schedule.

Mirrors ``WavLM.extract_features`` / ``TransformerEncoder.extract_features`` as the
kNN-SVC path uses them (wavlm/WavLM.py:323-375, 485-504, 572-612; wavlm/modules.py:
417-455, 457-564) but only for what the path consumes: the residual stream after
``n_layers`` layers (default 6 == the one-hot weighting of ddsp_matcher.py:88-89),
no final LayerNorm, no waveform normalisation, no masks.

Activations are channel-last [B*T, C] fp32 in HBM; every arithmetic step is a call
into libknnsvc_hip.so (MFMA implicit-GEMM convs/linears, fused LN(+GELU), gated
rel-pos flash attention).  The T x T bias is never materialised: the bucket
function (log + integer truncation) is evaluated once on the host, exactly as torch
does, into a (2T-1)-entry per-head table.
"""
from __future__ import annotations

import math

import os

import torch

from . import config as C
from . import ops


def _bucket_lut(T: int, num_buckets: int, max_distance: int) -> torch.Tensor:
    """rel = key - query in [-(T-1), T-1] -> bucket (wavlm/modules.py:417-442), on the host in torch
    so that the float log / truncation match the reference bit for bit."""
    rel = torch.arange(-(T - 1), T, dtype=torch.long)
    half = num_buckets // 2
    out = (rel > 0).to(torch.long) * half
    a = rel.abs()
    max_exact = half // 2
    large = max_exact + (torch.log(a.float() / max_exact) / math.log(max_distance / max_exact)
                         * (half - max_exact)).to(torch.long)
    large = torch.clamp(large, max=half - 1)
    return out + torch.where(a < max_exact, a, large)


def cat_rows(parts):
    """torch.cat(parts, 0) for row-major [rows_i, E] tensors — without the copy when the parts already lie back to back in one
    buffer (the chunks of a batch come out of encode_batch as consecutive rows of its output: a 30 s clip is ONE such view, a
    pool of 20 clips encoded in one batch is 20 adjacent ones).  The concatenation kernels were 0.34 ms of a 35 ms step."""
    if len(parts) == 1:
        return parts[0]
    first = parts[0]
    ok = first.dim() == 2 and first.is_contiguous()
    end = first.storage_offset() + first.numel() if ok else 0
    for t in parts[1:]:
        ok = (ok and t.dim() == 2 and t.is_contiguous() and t.shape[1] == first.shape[1] and t.dtype == first.dtype and
              t.untyped_storage().data_ptr() == first.untyped_storage().data_ptr() and t.storage_offset() == end)
        if not ok:
            return torch.cat(parts, 0)
        end += t.numel()
    rows = sum(int(t.shape[0]) for t in parts)
    return torch.as_strided(first, (rows, first.shape[1]), (first.shape[1], 1), first.storage_offset())


def _lib_mod():
    from . import _lib
    return _lib


def chunk_plan(n_samples: int, sr: int = C.SAMPLE_RATE, hop: int = C.HOP):
    """get_full_wavlm_features chunking (ddsp_prematch_dataset.py:275-293): 30 s windows, tails of
    <= 0.02*sr samples dropped, zero right-pad of hop - len % hop (a full hop when aligned)."""
    plan, start = [], 0
    while start < n_samples:
        ln = min(30 * sr, n_samples - start)
        if ln <= 0.02 * sr:
            break
        plan.append((start, ln, hop - (ln % hop)))
        start += 30 * sr
    return plan


class WavLMEncoder:
    """Packed WavLM weights on one GPU + forward schedule."""
    _uids = 0

    def weights_fingerprint(self) -> str:
        """Content identity of the packed weights (for the on-disk pool store): sha1 over shape, sum and |sum| (float64)
        of every packed GEMM weight and LayerNorm vector that shapes the output.  Bookkeeping, not path arithmetic."""
        if self._fingerprint is None:
            import hashlib
            ts = [c["w"] for c in self.conv] + [c["g"] for c in self.conv] + [self.ln_g, self.proj_w, self.pos_w]
            for L in self.layers:
                ts += [L["wqkv"], L["wo"], L["w1"], L["w2"], L["ln1_g"], L["ln2_g"], L["grep_a"]]
            h = hashlib.sha1()
            for t in ts:
                d = t.double()
                h.update(repr((tuple(t.shape), float(d.sum()), float(d.abs().sum()))).encode())
            self._fingerprint = h.hexdigest()
        return self._fingerprint

    def __init__(self, state: dict, cfg: dict, device="cuda", n_layers: int = C.MATCH_LAYER):
        self.cfg = cfg
        self.device = torch.device(device)
        self.n_layers = n_layers
        WavLMEncoder._uids += 1
        self.uid = WavLMEncoder._uids          # identity of this weight set in the pool-feature store
        self._fingerprint = None
        assert n_layers <= cfg["encoder_layers"]
        self.E = cfg["encoder_embed_dim"]
        self.H = cfg["encoder_attention_heads"]
        if self.E // self.H != 64:
            raise ValueError("the attention kernel is built for head_dim 64 (WavLM-Large / Base+)")
        dev = self.device
        f = lambda t: t.detach().to(torch.float32).contiguous().to(dev)
        fw = lambda t: ops.attach_split(f(t))       # GEMM weights: also keep the bf16x3 split (ops.attach_split)
        self.conv = []
        for i, (dim, k, s) in enumerate(C.conv_layers(cfg)):
            p = f"feature_extractor.conv_layers.{i}."
            self.conv.append(dict(w=fw(ops.pack_conv_weight(state[p + "0.weight"].float())), k=k, s=s, dim=dim,
                                  cin=state[p + "0.weight"].shape[1],
                                  g=f(state[p + "2.1.weight"]), b=f(state[p + "2.1.bias"])))
        self.ln_g, self.ln_b = f(state["layer_norm.weight"]), f(state["layer_norm.bias"])
        self.proj_w, self.proj_b = fw(state["post_extract_proj.weight"]), f(state["post_extract_proj.bias"])
        # positional conv: fold weight_norm(dim=2) once (wavlm/WavLM.py:526), then pack per group
        wpc = torch._weight_norm(state["encoder.pos_conv.0.weight_v"].float(), state["encoder.pos_conv.0.weight_g"].float(), 2)
        self.G = cfg["conv_pos_groups"]
        self.Kpos = cfg["conv_pos"]
        self.pos_w = fw(ops.pack_grouped_conv_weight(wpc, self.G))
        self.pos_b = f(state["encoder.pos_conv.0.bias"])
        self.layers = []
        for l in range(n_layers):
            p = f"encoder.layers.{l}."
            a = p + "self_attn."
            w8, b8 = state[a + "grep_linear.weight"].float(), state[a + "grep_linear.bias"].float()
            self.layers.append(dict(
                ln1_g=f(state[p + "self_attn_layer_norm.weight"]), ln1_b=f(state[p + "self_attn_layer_norm.bias"]),
                wqkv=fw(torch.cat([state[a + "q_proj.weight"], state[a + "k_proj.weight"], state[a + "v_proj.weight"]], 0)),
                bqkv=f(torch.cat([state[a + "q_proj.bias"], state[a + "k_proj.bias"], state[a + "v_proj.bias"]], 0)),
                wo=fw(state[a + "out_proj.weight"]), bo=f(state[a + "out_proj.bias"]),
                gate_w=f(torch.stack([w8[:4].sum(0), w8[4:].sum(0)])), gate_b=f(torch.stack([b8[:4].sum(), b8[4:].sum()])),
                grep_a=f(state[a + "grep_a"].reshape(-1)),
                ln2_g=f(state[p + "final_layer_norm.weight"]), ln2_b=f(state[p + "final_layer_norm.bias"]),
                w1=fw(state[p + "fc1.weight"]), b1=f(state[p + "fc1.bias"]),
                w2=fw(state[p + "fc2.weight"]), b2=f(state[p + "fc2.bias"]),
            ))
        self.rel_emb = state["encoder.layers.0.self_attn.relative_attention_bias.weight"].detach().float().cpu()
        self.plan = self._range_plan(state)
        self._tables = {}
        self.layer_mix = None      # general layer weighting (set_layer_mix): None = the output of layer n_layers
        self._graphs = {}          # (B, L, masked) -> (hipGraph, static output, static input, static lengths); insertion order = LRU order
        self._seen = set()
        self._graph_pool = None if not torch.cuda.is_available() else torch.cuda.graph_pool_handle()
        self.max_graphs = 32
        self.use_graphs = True
        self._h = None             # (C-side model handle, the layer mix it was made for): _handle()

    # -------------------------------------------------------------------------------------------
    A2_LIMIT = 0.9 * 65504.0 / 16.0          # |x| an activation may reach in the fixed-scale (16) split layout

    def _range_plan(self, state) -> dict:
        """Which activations may travel in the f16x2 split layout ("A2", fixed scale 16: |x| < 4094)?  Decided once, at
        load, from ANALYTIC bounds of every such tensor given the weights — never from data, so no input can overflow
        the layout: a LayerNorm output obeys |y_i| <= max|g| sqrt(C-1) + max|b| and ||y||_2 <= max|g| sqrt(C) + ||b||_2;
        a linear layer of it |y_j| <= ||x||_2 ||w_j||_2 + |b_j| (Cauchy-Schwarz); GELU and the softmax-weighted sum of
        V do not grow their inputs.  With trained or seeded WavLM weights every bound is far below the limit and the plan
        is "everything split"; a tensor whose bound is not (outlier-heavy fine-tunes) travels as fp32 instead and its
        consumer takes its scale from a device-side range slot (ops.absmax / out_absmax -> x_absmax), and a layer whose
        Q/K/V bound is not runs the bf16x3 attention kernel (fp32 exponent range)."""
        lim = self.A2_LIMIT
        f = lambda k: state[k].detach().float()
        ln_elem = lambda g, b, c: float(g.abs().max()) * math.sqrt(max(c - 1, 1)) + float(b.abs().max())
        ln_l2 = lambda g, b, c: float(g.abs().max()) * math.sqrt(c) + float(b.norm())
        lin = lambda l2, w, b: l2 * float(w.norm(dim=1).max()) + float(b.abs().max())
        plan = dict(conv=[], layers=[], bounds={})
        for i, (dim, _k, _s) in enumerate(C.conv_layers(self.cfg)):
            b = ln_elem(f(f"feature_extractor.conv_layers.{i}.2.1.weight"), f(f"feature_extractor.conv_layers.{i}.2.1.bias"), dim)
            plan["conv"].append(b < lim); plan["bounds"][f"conv{i}"] = b
        cdim = C.conv_layers(self.cfg)[-1][0]
        b = ln_elem(f("layer_norm.weight"), f("layer_norm.bias"), cdim)
        plan["feats"] = b < lim; plan["bounds"]["feats"] = b
        # the projection's output feeds the positional conv as fp32: its operand scale comes from THIS bound, fixed at load, not
        # from a range slot over the batch — a slot made a chunk's features depend (in their last bits, through the split of its
        # small elements) on which other chunks were encoded with it, and files are cached / searched across batches
        plan["bounds"]["proj"] = lin(ln_l2(f("layer_norm.weight"), f("layer_norm.bias"), cdim), f("post_extract_proj.weight"),
                                     f("post_extract_proj.bias"))
        E = self.E
        for l in range(self.n_layers):
            p = f"encoder.layers.{l}."
            g1, b1 = f(p + "self_attn_layer_norm.weight"), f(p + "self_attn_layer_norm.bias")
            g2, b2 = f(p + "final_layer_norm.weight"), f(p + "final_layer_norm.bias")
            xn_e, xn_2 = ln_elem(g1, b1, E), ln_l2(g1, b1, E)
            a = p + "self_attn."
            qb = lin(xn_2, f(a + "q_proj.weight"), f(a + "q_proj.bias"))
            kb = lin(xn_2, f(a + "k_proj.weight"), f(a + "k_proj.bias"))
            vb = lin(xn_2, f(a + "v_proj.weight"), f(a + "v_proj.bias"))
            x2_e, x2_2 = ln_elem(g2, b2, E), ln_l2(g2, b2, E)
            hb = lin(x2_2, f(p + "fc1.weight"), f(p + "fc1.bias"))
            # the f16x2 attention kernel splits K and V at scale 16 and Q at 16 log2(e)/8 (attention.hip)
            narrow = kb < lim and vb < lim and qb < lim * 5.0
            plan["layers"].append(dict(xn=xn_e < lim, xn2=x2_e < lim, h=hb < lim, attn_f16=narrow))
            plan["bounds"][f"layer{l}"] = dict(xn=xn_e, q=qb, k=kb, v=vb, xn2=x2_e, h=hb)
        return plan

    def set_layer_mix(self, weights) -> None:
        """General layer weighting (ddsp_prematch_dataset.py:349-350: ``(feats * w[:, None]).sum(0)`` over the 25 stacked layer
        results — index 0 is the encoder's input after the positional conv, index l the output of layer l).  ``weights``: a
        sequence whose entries beyond ``n_layers`` are zero, or None / a one-hot on ``n_layers`` for the plain layer output (the
        live path: layer 6).  The encoder then returns sum_l w[l] * layer_result[l], terms added in ascending l."""
        if weights is not None:
            w = [float(v) for v in torch.as_tensor(weights).reshape(-1).tolist()]
            if any(v != 0.0 for v in w[self.n_layers + 1:]):
                raise ValueError(f"layer weighting uses layers beyond the {self.n_layers} this encoder was loaded with")
            w = (w + [0.0] * (self.n_layers + 1))[:self.n_layers + 1]
            if sum(1 for v in w if v != 0.0) == 1 and w[self.n_layers] == 1.0:
                w = None
            weights = None if w is None else tuple(w)
        if weights != self.layer_mix:
            self.layer_mix = weights
            self._graphs.clear(); self._seen.clear()          # captured schedules end in the old weighting
            WavLMEncoder._uids += 1
            self.uid = WavLMEncoder._uids                     # cached pool features belong to the old weighting

    def n_frames(self, n_samples: int) -> int:
        n = n_samples
        for c in self.conv:
            n = (n - c["k"]) // c["s"] + 1
        return n

    def _table(self, T: int) -> torch.Tensor:
        if T not in self._tables:
            lut = _bucket_lut(T, self.cfg["num_buckets"], self.cfg["max_distance"])
            self._tables[T] = self.rel_emb[lut].T.contiguous().to(self.device)       # [H, 2T-1]
        return self._tables[T]

    @torch.inference_mode()
    def encode_batch(self, wav: torch.Tensor, lens: torch.Tensor | None = None) -> torch.Tensor:
        """[B, L] equal-length (already padded) chunks on the GPU -> [B, T, E].

        ``lens`` (int32 [B] on the device, optional): valid frames per row — the rows are chunks of different lengths that
        were zero-padded up to a common BUCKET length; frames >= lens[b] are masked exactly the way WavLM's own padding mask
        does it (zeroed in front of the positional conv, excluded as attention keys: wavlm/WavLM.py:311-321, 353, 574-575),
        so rows < lens[b] come out bit-identical to encoding the chunk at its own length.  Rows >= lens[b] are garbage.

        The ~70 launches of one batch are captured into a hipGraph per (B, L, masked?) and replayed: every kernel takes
        caller-owned buffers, the lengths are read on the device, so ONE graph serves every utterance of its bucket.  The
        cache is LRU (the 30 s graph of a long-running job must not be the first to go), all graphs share one memory pool
        (they are only ever replayed one after the other on the encoding stream), and capturing does not synchronise the
        device."""
        key = tuple(wav.shape) + (lens is not None,)
        if not self.use_graphs or torch.cuda.is_current_stream_capturing():
            return self._encode_batch(wav, lens)
        ent = self._graphs.get(key)
        if ent is None:
            self._table(self.n_frames(wav.shape[1]))       # bias table upload (host -> device) cannot be captured
            if key not in self._seen:                      # first sight of a shape: run it eagerly (one-time function
                self._seen.add(key)                        # attributes, allocator warm-up); capture when it comes back
                return self._encode_batch(wav, lens)
            sin = wav.clone()
            slen = lens.clone() if lens is not None else None
            ent = self._graphs[key] = ops.capture_graph(lambda: self._encode_batch(sin, slen), self.device, self._graph_pool) + (sin, slen)
            while len(self._graphs) > self.max_graphs:
                self._graphs.pop(next(iter(self._graphs)))            # least recently used
        else:
            self._graphs[key] = self._graphs.pop(key)                 # mark as most recently used
        g, out, sin, slen = ent
        sin.copy_(wav)
        if slen is not None:
            slen.copy_(lens)
        g.replay()
        return out.clone()

    # -------------------------------------------------------------------------------------------
    def _handle(self):
        """The C-side model (knnsvc_wavlm_create: include/knnsvc_hip.h, "Whole-model entry points"): a copy of the descriptor —
        pointers to THIS object's packed weights, the range plan, the layer mix — behind which ONE call enqueues the whole
        layer sequence.  Rebuilt when the layer mix changes."""
        key = self.layer_mix
        if self._h is not None and self._h[1] == key:
            return self._h[0]
        self._free_handle()
        import ctypes as Ct
        from . import _lib
        lib = _lib.load()
        plan = self.plan

        def W(t):
            w = _lib.Weight()
            w.w = t.data_ptr()
            w2 = getattr(t, "_w2", None)
            w.w_f16x2 = w2.data_ptr() if w2 is not None else None
            w.w_f16x2_scale = float(getattr(t, "_w2_scale", 0.0)) if w2 is not None else 0.0
            return w
        convs = (_lib.WavlmConv * len(self.conv))()
        for i, c in enumerate(self.conv):
            convs[i].w = W(c["w"]); convs[i].ln_g = c["g"].data_ptr(); convs[i].ln_b = c["b"].data_ptr()
            convs[i].dim, convs[i].k, convs[i].stride, convs[i].cin = c["dim"], c["k"], c["s"], c["cin"]
            convs[i].out_split = 1 if plan["conv"][i] else 0
        layers = (_lib.WavlmLayer * max(1, len(self.layers)))()
        for i, (ly, pl) in enumerate(zip(self.layers, plan["layers"])):
            L_ = layers[i]
            for f_ in ("ln1_g", "ln1_b", "ln2_g", "ln2_b", "bqkv", "bo", "b1", "b2", "gate_w", "gate_b", "grep_a"):
                setattr(L_, f_, ly[f_].data_ptr())
            L_.wqkv, L_.wo, L_.w1, L_.w2 = W(ly["wqkv"]), W(ly["wo"]), W(ly["w1"]), W(ly["w2"])
            L_.xn_split, L_.xn2_split, L_.h_split, L_.attn_f16 = (1 if pl[k_] else 0 for k_ in ("xn", "xn2", "h", "attn_f16"))
        d = _lib.WavlmDesc()
        d.n_conv, d.n_layers, d.conv, d.layers = len(self.conv), len(self.layers), convs, layers
        d.ln_g, d.ln_b, d.feats_split = self.ln_g.data_ptr(), self.ln_b.data_ptr(), 1 if plan["feats"] else 0
        d.proj, d.proj_b = W(self.proj_w), self.proj_b.data_ptr()
        d.pos, d.pos_b, d.pos_groups, d.pos_k = W(self.pos_w), self.pos_b.data_ptr(), self.G, self.Kpos
        pb = plan["bounds"]["proj"]
        d.pos_a_scale = ops.pick_scale(pb) if (math.isfinite(pb) and 0.0 < pb < 1e30) else 0.0
        d.E, d.H, d.ffn = self.E, self.H, (self.layers[0]["w1"].shape[0] if self.layers else 0)
        mix = None
        if self.layer_mix is not None:
            mix = (Ct.c_float * (self.n_layers + 1))(*self.layer_mix)
            d.layer_mix = mix
        h = Ct.c_void_p()
        ops.check(lib.knnsvc_wavlm_create(Ct.byref(d), Ct.byref(h)), "wavlm_create")
        self._h = (h, key)
        return h

    def _free_handle(self):
        if getattr(self, "_h", None) is not None:
            from . import _lib
            _lib.load().knnsvc_wavlm_free(self._h[0])
            self._h = None

    def __del__(self):
        try:
            self._free_handle()
        except Exception:
            pass

    def _handle_ok(self) -> bool:
        """The one-call path covers the PRODUCT configuration; with any of the A/B switches set (another GEMM / attention mode,
        range slots off, the fp32 layouts, a slot-scaled positional conv) or a debugging tap installed, the launch-by-launch
        host sequence below runs instead — the same kernels with the same arguments."""
        return (ops.gemm_mode() == "f16x2" and ops.attention_mode() == "f16x2" and ops.range_slots_on() and
                os.environ.get("KNNSVC_A2", "1") != "0" and os.environ.get("KNNSVC_POS_SLOT") != "1" and
                os.environ.get("KNNSVC_WAVLM_HOST_SEQ") != "1" and getattr(self, "_tap", None) is None)

    def _encode_batch(self, wav: torch.Tensor, lens: torch.Tensor | None = None) -> torch.Tensor:
        if self._handle_ok():
            # WavLM.extract_features (wavlm/WavLM.py:323-375) = ONE call into the library (knnsvc_wavlm_encode)
            lib = _lib_mod().load()
            B, L = wav.shape
            h = self._handle()
            T = int(lib.knnsvc_wavlm_frames(h, L))
            out = torch.empty(B * T, self.E, device=wav.device, dtype=torch.float32)
            nb = int(lib.knnsvc_wavlm_workspace_bytes(h, B, L))
            ws = torch.empty(nb, device=wav.device, dtype=torch.uint8)
            x = wav.contiguous()
            ops.check(lib.knnsvc_wavlm_encode(h, x.data_ptr(), B, L, lens.data_ptr() if lens is not None else None,
                                              self._table(T).data_ptr(), out.data_ptr(), ws.data_ptr(), nb, ops._stream()), "wavlm_encode")
            return out.view(B, T, self.E)
        return self._encode_batch_host(wav, lens)

    def _encode_batch_host(self, wav: torch.Tensor, lens: torch.Tensor | None = None) -> torch.Tensor:
        """The same forward, launch by launch from the host (rounds 1-4; kept for the A/B switches and as the reference the
        one-call path is tested against: tests/test_gpu_models.py)."""
        B, L = wav.shape
        dev = wav.device
        x = wav.contiguous()
        t_in, cin = L, 1
        # Activations that only feed GEMMs travel in the f16x2 split layout ("A2", include/knnsvc_hip.h): the
        # producer (LayerNorm, the fused first conv, the GELU epilogue of FFN1, attention) splits every element once
        # and the GEMM stages its A operand with plain copies instead of re-splitting it in every column tile.
        a2 = ops.gemm_mode() == "f16x2" and os.environ.get("KNNSVC_A2", "1") != "0"
        dyn = ops.gemm_mode() == "f16x2" and ops.range_slots_on()   # fp32 GEMM inputs take their scale from a device range slot
        plan = self.plan
        sp = lambda dim, ok=True: a2 and ok and dim % 32 == 0      # a [*, dim] activation can be carried split
        slot_of = lambda t: ops.absmax(t) if dyn else None          # bound of a tensor no GEMM epilogue produced
        x_sp = False                                    # is x currently in the split layout?
        for li, c in enumerate(self.conv):
            t_out = (t_in - c["k"]) // c["s"] + 1
            if li == 0 and cin == 1 and c["dim"] in (64, 128, 256, 512) and c["k"] <= 16 and c["s"] <= 8:
                x_sp = sp(c["dim"], plan["conv"][0]) and c["dim"] >= 256 and len(self.conv) > 1
                x = ops.wavlm_conv0(x, c["w"], c["g"], c["b"], c["k"], c["s"], out_split=x_sp)   # conv + LN + GELU in one pass
                t_in, cin = t_out, c["dim"]
                continue
            y = torch.empty(B * t_out, c["dim"], device=dev, dtype=torch.float32)
            ops.conv_gemm(x, c["w"], y, m=t_out, n=c["dim"], cin=cin, taps=c["k"], stride=c["s"], t_in=t_in,
                          batches=B, x_bstride=t_in * cin, o_bstride=t_out * c["dim"], x_split=x_sp,
                          x_absmax=None if (x_sp or cin % 32) else slot_of(x.view(-1, cin)))
            x_sp = sp(c["dim"], plan["conv"][li]) and li + 1 < len(self.conv)         # the last layer's output feeds a LayerNorm, not a GEMM
            ops.layernorm(y, c["g"], c["b"], gelu=True, out=y, out_split=x_sp)
            x, t_in, cin = y, t_out, c["dim"]
        T = t_in
        assert not x_sp
        tap = getattr(self, "_tap", None)               # debugging aid (tools/layer_error.py): intermediate activations
        if tap is not None:
            tap["conv"] = x.clone()
        f_sp = sp(cin, plan["feats"])
        feats = ops.layernorm(x, self.ln_g, self.ln_b, out_split=f_sp)
        pb = plan["bounds"]["proj"]
        fixed_pos = ops.gemm_mode() == "f16x2" and math.isfinite(pb) and 0.0 < pb < 1e30 and os.environ.get("KNNSVC_POS_SLOT") != "1"
        x_slot = ops.new_slot(dev) if (dyn and not fixed_pos) else None
        x = ops.linear(feats, self.proj_w, self.proj_b, x_split=f_sp, x_absmax=None if f_sp else slot_of(feats),
                       out_absmax=x_slot)             # [B*T, E]
        E, H, G, K = self.E, self.H, self.G, self.Kpos
        cg = E // G
        if tap is not None:
            tap["proj"] = x.clone()
        if lens is not None:
            ops.mask_rows(x, B, T, lens)                 # x[padding_mask] = 0 before the positional conv (WavLM.py:574-575)
        x2 = torch.empty_like(x)
        ops.conv_gemm(x, self.pos_w, x2, m=T, n=cg, cin=cg, taps=K, pad=K // 2, t_in=T, ldx=E, ldo=E,
                      bias=self.pos_b, act=ops.ACT_GELU, resid=x, ldr=E, batches=B, groups=G,
                      x_bstride=T * E, x_gstride=cg, w_gstride=cg * cg * K, bias_gstride=cg,
                      o_bstride=T * E, o_gstride=cg, r_bstride=T * E, r_gstride=cg, x_absmax=x_slot,
                      a_scale=ops.pick_scale(pb) if fixed_pos else 0.0)
        x = x2
        table = self._table(T)
        hdim = self.layers[0]["w1"].shape[0] if self.layers else 0
        mix = self.layer_mix
        acc = None
        if mix is not None:
            acc = torch.empty_like(x)
            ops.axpy(x, mix[0], acc, False)                  # layer_results[0]: the encoder's input (WavLM.py:583-585)
        for li_, (ly, pl) in enumerate(zip(self.layers, plan["layers"])):
            e_sp = sp(E, pl["xn"])
            xn = ops.layernorm(x, ly["ln1_g"], ly["ln1_b"], out_split=e_sp)
            gate = ops.wavlm_gate(xn, H, ly["gate_w"], ly["gate_b"], ly["grep_a"], x_split=e_sp)
            # K and V leave the projection pre-split (every query block of a head re-split the same keys otherwise); Q stays fp32
            narrow = pl["attn_f16"] or ops.attention_mode() != "f16x2"
            kv_sp = sp(E) and narrow and ops.attention_mode() == "f16x2"
            qkv = ops.linear(xn, ly["wqkv"], ly["bqkv"], x_split=e_sp, out_split=E if kv_sp else False,
                             x_absmax=None if e_sp else slot_of(xn))
            a_sp = sp(E) and narrow                       # attention output <= max|V|: same bound
            att = ops.wavlm_attention(qkv, gate, table, B, T, H, out_split=a_sp, kv_split=kv_sp, wide=not narrow, kv_len=lens)
            x = ops.linear(att, ly["wo"], ly["bo"], resid=x, x_split=a_sp, x_absmax=None if a_sp else slot_of(att))
            e2_sp = sp(E, pl["xn2"])
            xn = ops.layernorm(x, ly["ln2_g"], ly["ln2_b"], out_split=e2_sp)
            h_sp = sp(hdim, pl["h"])
            h_slot = ops.new_slot(dev) if (dyn and not h_sp) else None
            hmid = ops.linear(xn, ly["w1"], ly["b1"], act=ops.ACT_GELU, x_split=e2_sp, out_split=h_sp,
                              x_absmax=None if e2_sp else slot_of(xn), out_absmax=h_slot)
            x = ops.linear(hmid, ly["w2"], ly["b2"], resid=x, x_split=h_sp, x_absmax=h_slot)
            if acc is not None and mix[li_ + 1] != 0.0:
                ops.axpy(x, mix[li_ + 1], acc, True)
        if acc is not None:
            return acc.view(B, T, E)
        return x.view(B, T, E)

    def full_features(self, wav_1d: torch.Tensor, max_batch: int = 8) -> torch.Tensor:
        """One utterance [L] on the GPU -> [T_total, E] (get_full_wavlm_features + layer select)."""
        return self.encode_many([wav_1d], max_batch=max_batch)[0]

    BUCKET_FRAMES = 50          # ragged chunks are padded up to a multiple of this many frames (1 s)

    def bucket_frames(self, T: int) -> int:
        q = self.BUCKET_FRAMES
        return min(-(-T // q) * q, max(T, C.CHUNK_SAMPLES // C.HOP))

    def encode_many(self, wavs, max_batch: int = 8, pow2_batches: bool = False):
        """List of utterances -> list of [T_i, E].  Chunks are independent (ddsp_prematch_dataset.py:275-293), so the 30 s
        chunks of ALL utterances are batched together, and the ragged tails — in dataset mode: nearly every utterance — are
        grouped into length BUCKETS: each is cut / zero-padded to the bucket's sample count (320 T_b + 80: exactly the
        receptive field of T_b frames; samples past a chunk's own last frame never reach a frame < T) and encoded with its own
        frame count as the mask length (encode_batch).  A bucket is one (B, L) shape = one hipGraph, instead of one per
        distinct utterance length.  ``pow2_batches``: groups are split into batches of 2^k rows (dataset mode: the batch size
        then also comes from a small set)."""
        jobs = []          # (utt, start, len, own frames)
        for u, w in enumerate(wavs):
            for (s_, l, p) in chunk_plan(w.numel()):
                jobs.append((u, s_, l, self.n_frames(l + p)))
        by_bucket = {}
        for j in jobs:
            by_bucket.setdefault(self.bucket_frames(j[3]), []).append(j)
        pieces = {}
        for Tb, grp in by_bucket.items():
            Lb = C.HOP * Tb + 80
            assert self.n_frames(Lb) == Tb
            i = 0
            while i < len(grp):
                n = min(max_batch, len(grp) - i)
                if pow2_batches:
                    n = 1 << (n.bit_length() - 1)
                sub = grp[i:i + n]
                i += n
                buf = torch.zeros(len(sub), Lb, device=self.device, dtype=torch.float32)
                dsts, srcs = [], []
                for r, (u, s_, l, _t) in enumerate(sub):
                    k = min(l, Lb)
                    dsts.append(buf[r, :k]); srcs.append(wavs[u][s_:s_ + k])
                torch._foreach_copy_(dsts, srcs)             # one multi-tensor launch instead of one copy kernel per chunk
                exact = all(t == Tb for (_u, _s, _l, t) in sub)
                lens = None if exact else torch.tensor([t for (_u, _s, _l, t) in sub], dtype=torch.int32).to(self.device, non_blocking=True)
                out = self.encode_batch(buf, lens)
                for r, (u, s_, _l, t) in enumerate(sub):
                    pieces[(u, s_)] = out[r, :t]
        res = []
        for u, w in enumerate(wavs):
            parts = [pieces[(u, s_)] for (s_, _l, _p) in chunk_plan(w.numel())]
            res.append(cat_rows(parts) if parts else torch.empty(0, self.E, device=self.device))
        return res
