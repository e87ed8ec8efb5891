"""Host side of the conditioned HiFi-GAN generator ("SynthesizerTrn").

Mirrors ``SynthesizerTrn.forward`` + ``Generator.forward`` of the reference:
'mix' = hifigan/ddsp_models.py:108-233, 405-493 (additive-synth excitation, doubling
side channels), 'f0' = hifigan/ddsp_models_f0.py:106-216, 320-381 (sine excitation,
n_harmonic+2 side channels).  Weight norm is live in the reference at inference
(ddsp_hubconf.py:100-102); it is folded once here (w = g * v / ||v||).

Everything is channel-last [T, C] fp32.  Skip connections are zero-copy: producers
write straight into column blocks of the concat buffers (``ldo`` / column offset of
the conv kernel), so ``torch.cat`` never happens.  The transposed convolutions run as
one GEMM over K = (k/u)*Cin, N = u*Cout with a row-scatter epilogue.
"""
from __future__ import annotations

import os

import torch

from . import ops

LRELU = 0.1


def _fold(sd, name):
    if name + ".weight" in sd:
        return sd[name + ".weight"].float()
    return torch._weight_norm(sd[name + ".weight_v"].float(), sd[name + ".weight_g"].float(), 0)


import threading as _threading
_SERIAL = _threading.local()          # per host thread


class serial_resblocks:
    """Context (A/B aid, equality tests): the three ResBlock branches of a stage are launched one after the other — one launch per
    branch and step — instead of one grid per step (ops.conv_gemm_multi / ops.resblock_pair_multi).  Same kernels, same
    descriptors, same buffers: the waveform is bit-identical either way.  (Until round 4 this context switched off the
    per-branch STREAMS; those are gone — see Vocoder._forward.)"""
    def __enter__(self):
        self.prev = getattr(_SERIAL, "on", False); _SERIAL.on = True
    def __exit__(self, *a):
        _SERIAL.on = self.prev


class Vocoder:
    def __init__(self, state: dict, h: dict, kind: str = "mix", device="cuda"):
        assert kind in ("mix", "f0")
        self.h, self.kind, self.device = h, kind, torch.device(device)
        dev = self.device
        f0_ = lambda t: t.detach().float().contiguous().to(dev)
        f = lambda t: ops.attach_split(f0_(t)) if t.dim() == 2 else f0_(t)     # 2-D = packed GEMM weights
        self.rates, self.ksz = list(h["upsample_rates"]), list(h["upsample_kernel_sizes"])
        self.n_up = len(self.rates)
        self.hop = h["hop_size"]
        self.sr = h["sampling_rate"]
        nh, uic = h["n_harmonic"], h["upsample_initial_channel"]
        self.uic = uic
        self.lin_w, self.lin_b = f(state["dec.lin_pre.weight"]), f(state["dec.lin_pre.bias"])
        self.pre_w, self.pre_b = f(ops.pack_conv_weight(state["dec.conv_pre.weight"].float())), f(state["dec.conv_pre.bias"])
        # side (down) path channel counts: res[0] = cond, res[i+1] = output of down stage i
        if kind == "mix":
            self.side = [nh * 2 ** i for i in range(self.n_up + 1)]
        else:
            self.side = [nh + 2] * (self.n_up + 1)
        self.downs, self.rbd = [], []
        for i in range(self.n_up):
            j = self.n_up - 1 - i
            self.downs.append(dict(w=f(ops.pack_conv_weight(_fold(state, f"dec.downs.{i}"))), b=f(state[f"dec.downs.{i}.bias"]),
                                   k=self.ksz[j], u=self.rates[j]))
            nm = f"dec.resblocks_downs.{i}.convs.0"
            self.rbd.append(dict(w=f(ops.pack_conv_weight(_fold(state, nm))), b=f(state[nm + ".bias"])))
        self.cpre_w, self.cpre_b = f(ops.pack_conv_weight(state["dec.concat_pre.weight"].float())), f(state["dec.concat_pre.bias"])
        self.ups, self.ccv, self.res = [], [], []
        nk = len(h["resblock_kernel_sizes"])
        for i in range(self.n_up):
            u, k = self.rates[i], self.ksz[i]
            cout = uic // 2 ** (i + 1)
            self.ups.append(dict(w=f(ops.pack_convT_weight(_fold(state, f"dec.ups.{i}"), u)), b=f(state[f"dec.ups.{i}.bias"]),
                                 u=u, k=k, cout=cout, cin=uic // 2 ** i))
            self.ccv.append(f(ops.pack_conv_weight(state[f"dec.concat_conv.{i}.weight"].float())))
            blocks = []
            for j, (k_r, dil) in enumerate(zip(h["resblock_kernel_sizes"], h["resblock_dilation_sizes"])):
                nm = f"dec.resblocks.{i * nk + j}"
                convs = []
                for m, d in enumerate(dil):
                    w1_ = _fold(state, f"{nm}.convs1.{m}")
                    convs.append(dict(w1=f(ops.pack_conv_weight(w1_)), b1=f(state[f"{nm}.convs1.{m}.bias"]),
                                      # |convs1(lrelu(x)) + b1| <= max_n sum_k |w1[n, k]| * max|x| + max|b1|: the inner tensor of a
                                      # ResBlock pair is bounded through its input's range slot, not measured (see _forward)
                                      t1_bound=(float(w1_.abs().sum(dim=(1, 2)).max()), float(state[f"{nm}.convs1.{m}.bias"].abs().max())),
                                      w2=f(ops.pack_conv_weight(_fold(state, f"{nm}.convs2.{m}"))), b2=f(state[f"{nm}.convs2.{m}.bias"]),
                                      d=d))
                blocks.append(dict(k=k_r, convs=convs))
            self.res.append(blocks)
        self.post_w = f(ops.pack_conv_weight(state["dec.conv_post.weight"].float()))
        self.prenet_w = f(state["sin_prenet.weight"].float().reshape(-1, 3))
        self.prenet_b = f(state["sin_prenet.bias"])
        self._dyn = None
        self._graphs = {}          # frame bucket -> (hipGraph, static inputs, device frame count, static output); insertion order = LRU
        self._seen = set()
        self._graph_pool = None if not torch.cuda.is_available() else torch.cuda.graph_pool_handle()
        self._graph_pools = {0: self._graph_pool}
        self.max_graphs = 48            # (bucket, mode) instances PER tail stream (Vocoder._evict)
        self.use_graphs = True
        self.merge_branches = os.environ.get("KNNSVC_MERGE_BRANCHES", "1") != "0"
        self._h = None             # C-side model handle (_handle())

    def _merged(self) -> bool:
        """One grid per step for the three ResBlock branches of a stage (default); KNNSVC_MERGE_BRANCHES=0 / serial_resblocks():
        one launch per branch."""
        return self.merge_branches and not getattr(_SERIAL, "on", False)

    # -------------------------------------------------------------------------------------------
    def _conv(self, x, w, out, *, T_in, cin, cout, k, **kw):
        return ops.conv_gemm(x, w, out, n=cout, cin=cin, taps=k, t_in=T_in, dyn=self._dyn, **kw)

    BUCKET_FRAMES = 25          # frame counts are rounded up to a multiple of this (0.5 s) for the graph cache

    @torch.inference_mode()
    def forward(self, c: torch.Tensor, f0: torch.Tensor, harm: torch.Tensor | None = None) -> torch.Tensor:
        """c [N, hubert_dim], f0 [N], harm [N, 49] (mix only), all fp32 on the GPU -> waveform [N*hop].

        The generator is ~110 short kernel launches whose host-side launch cost exceeds their device time at 30 s and
        below, so its schedule is captured into a hipGraph and replayed.  Real utterances almost all differ in length, so a
        graph is captured per frame-count BUCKET (N rounded up to 25 frames), not per N: every launch inside takes its
        lengths from a device-side frame count (knnsvc_conv_desc.n_dyn: input rows past the valid length read as the
        convolutions' own zero padding, rows past it are not written), which makes the bucket's graph compute exactly what
        an exact-length run computes (`test_vocoder_bucket_graph_equals_exact_length`).  A shape is run eagerly at first
        sight and captured when it comes back; capture does not synchronise the device (ops.capture_graph), so meeting a new
        bucket inside the dataset-mode stream pipeline costs one eager pass, not a pipeline stall.  LRU cache; the graphs of ONE
        tail stream share a memory pool (they only ever replay one after the other), each further tail of the stream pipeline
        has its own instances and its own pool (pipeline.current_tail())."""
        N = c.shape[0]
        if not self.use_graphs or torch.cuda.is_current_stream_capturing():
            return self._forward(c, f0, harm)
        q = self.BUCKET_FRAMES
        Nb = -(-N // q) * q
        key = Nb if self._merged() else (Nb, "serial")
        from . import pipeline
        slot = pipeline.current_tail()          # tail stream index of the stream pipeline (0 outside one): graphs replayed from
        if slot:                                # different tail streams may overlap — one instance and one memory pool per tail
            key = (key, slot)
        ent = self._graphs.get(key)
        if ent is None:
            if key not in self._seen:                      # first sight: eager, exact length
                self._seen.add(key)
                return self._forward(c, f0, harm)
            dev = c.device
            sc = torch.zeros(Nb, c.shape[1], device=dev, dtype=torch.float32)
            sf = torch.zeros(Nb, device=dev, dtype=torch.float32)
            sh = torch.zeros(Nb, harm.shape[1], device=dev, dtype=torch.float32) if harm is not None else None
            nd = torch.full((1,), N, device=dev, dtype=torch.int32)
            if slot not in self._graph_pools:
                self._graph_pools[slot] = torch.cuda.graph_pool_handle()
            g, out = ops.capture_graph(lambda: self._forward(sc, sf, sh, n_dyn=nd), self.device, self._graph_pools[slot])
            ent = self._graphs[key] = (g, sc, sf, sh, nd, out, [None])
            self._evict(slot, keep=key)
        else:
            self._graphs[key] = self._graphs.pop(key)                 # most recently used
        g, sc, sf, sh, nd, out, last = ent
        sc[:N].copy_(c); sf[:N].copy_(f0)
        if sh is not None:
            sh[:N].copy_(harm)
        nd.fill_(N)
        g.replay()
        y = out[:N * self.hop].clone()
        last[0] = torch.cuda.current_stream(c.device)       # the entry may be destroyed once this stream has drained (a stream
        return y                                            # object, not an event: events held at interpreter exit crash in hipEventDestroy)

    def _evict(self, slot, keep=None) -> None:
        """LRU per tail stream (``max_graphs`` instances EACH: the key space is buckets x tails, and the tails replay
        independently).  An evicted key is forgotten altogether — its next sight is an eager pass again, not an immediate
        re-capture, so a length distribution wider than the cache degrades to eager passes instead of a capture per call — and
        only an entry whose last replay has finished is destroyed (its static buffers go back to the tail's pool).  The entry
        just captured (``keep``) is never a victim.  Under a steady pipeline the tails are never idle, so "skip the busy ones"
        alone would let the cache grow without bound: beyond twice the budget the oldest entry's stream is waited for once
        (a host wait on work that is already queued: it ends) and the entry goes."""
        mine = [k for k in self._graphs if k != keep and (k[1] if isinstance(k, tuple) and not isinstance(k[1], str) else 0) == slot]
        extra = len(mine) + (1 if keep is not None else 0) - self.max_graphs
        hard = len(mine) + (1 if keep is not None else 0) - 2 * self.max_graphs
        for k in mine:                                                    # insertion order = least recently used first
            if extra <= 0:
                break
            st = self._graphs[k][6][0]
            if st is not None and not st.query():
                if hard <= 0:
                    continue                                              # its stream is still busy: try the next oldest
                st.synchronize()                                          # bounded overshoot: this one has to go
            self._graphs.pop(k)
            self._seen.discard(k)
            extra -= 1
            hard -= 1

    # -------------------------------------------------------------------------------------------
    def _handle(self):
        """The C-side model (knnsvc_generator_create, include/knnsvc_hip.h "Whole-model entry points"): pointers to THIS object's
        packed weights behind which one call enqueues the whole launch sequence."""
        if self._h is not None:
            return self._h
        import ctypes as Ct
        from . import _lib

        def W(t):
            w = _lib.Weight()
            w.w = t.data_ptr()
            w2 = getattr(t, "_w2", None)
            w.w_f16x2 = w2.data_ptr() if w2 is not None else None
            w.w_f16x2_scale = float(getattr(t, "_w2_scale", 0.0)) if w2 is not None else 0.0
            return w
        h = self.h
        stages = (_lib.GenStage * self.n_up)()
        for i in range(self.n_up):
            st, up, dn, rb = stages[i], self.ups[i], self.downs[i], self.rbd[i]
            st.up, st.up_b, st.u, st.k, st.cin, st.cout = W(up["w"]), up["b"].data_ptr(), up["u"], up["k"], up["cin"], up["cout"]
            st.ccv = W(self.ccv[i])
            for j, blk in enumerate(self.res[i]):
                st.res_k[j] = blk["k"]
                for m, cv in enumerate(blk["convs"]):
                    pr = st.res[j][m]
                    pr.w1, pr.b1, pr.w2, pr.b2, pr.dil = W(cv["w1"]), cv["b1"].data_ptr(), W(cv["w2"]), cv["b2"].data_ptr(), cv["d"]
                    pr.t1_bound_mul, pr.t1_bound_add = float(cv["t1_bound"][0]), float(cv["t1_bound"][1])
            st.down, st.down_b, st.down_k, st.down_u = W(dn["w"]), dn["b"].data_ptr(), dn["k"], dn["u"]
            st.rbd, st.rbd_b = W(rb["w"]), rb["b"].data_ptr()
        d = _lib.GeneratorDesc()
        d.kind, d.n_up, d.hop, d.sample_rate = (0 if self.kind == "mix" else 1), self.n_up, self.hop, self.sr
        d.n_harm_in, d.uic, d.hubert_dim, d.hifi_dim = 49, self.uic, self.lin_w.shape[1], self.lin_w.shape[0]
        for i, v in enumerate(self.side):
            d.side[i] = v
        d.lin, d.lin_b, d.pre, d.pre_b = W(self.lin_w), self.lin_b.data_ptr(), W(self.pre_w), self.pre_b.data_ptr()
        d.cpre, d.cpre_b, d.post = W(self.cpre_w), self.cpre_b.data_ptr(), W(self.post_w)
        d.prenet_w, d.prenet_b = self.prenet_w.data_ptr(), self.prenet_b.data_ptr()
        d.stages = stages
        hd = Ct.c_void_p()
        ops.check(_lib.load().knnsvc_generator_create(Ct.byref(d), Ct.byref(hd)), "generator_create")
        self._h = hd
        return hd

    def __del__(self):
        try:
            if getattr(self, "_h", None) is not None:
                from . import _lib
                _lib.load().knnsvc_generator_free(self._h)
                self._h = None
        except Exception:
            pass

    def _handle_ok(self) -> bool:
        """The one-call path covers the PRODUCT configuration (f16x2 GEMMs with range slots, fused pairs at C = 32 / 64, three
        ResBlocks of three pairs per stage, branches as one grid); any A/B switch sends the forward down the host sequence."""
        return (self._merged() and ops.gemm_mode() == "f16x2" and ops.range_slots_on() and not ops._SLOT_DBG and
                os.environ.get("KNNSVC_FUSED_PAIR", "1") != "0" and os.environ.get("KNNSVC_FUSED_PAIR_128", "0") != "1" and
                os.environ.get("KNNSVC_VOCODER_HOST_SEQ") != "1" and
                all(len(blocks) == 3 and all(len(b["convs"]) == 3 for b in blocks) for blocks in self.res))

    def _forward(self, c: torch.Tensor, f0: torch.Tensor, harm: torch.Tensor | None = None, n_dyn: torch.Tensor | None = None) -> torch.Tensor:
        """``n_dyn`` (device int32 [1], <= N): the valid frame count of a forward laid out for N = a bucket's frames.
        SynthesizerTrn.forward + Generator.forward = ONE call into the library (knnsvc_generator_forward)."""
        if not self._handle_ok() or (self.kind == "mix" and (harm is None or harm.shape[1] != 49)):
            return self._forward_host(c, f0, harm, n_dyn)
        from . import _lib
        lib = _lib.load()
        hd = self._handle()
        N = c.shape[0]
        out = torch.empty(N * self.hop, device=c.device, dtype=torch.float32)
        nb = int(lib.knnsvc_generator_workspace_bytes(hd, N))
        ws = torch.empty(nb, device=c.device, dtype=torch.uint8)
        cc, ff = c.contiguous(), f0.contiguous()
        hh = harm.contiguous() if (self.kind == "mix") else None
        ops.check(lib.knnsvc_generator_forward(hd, cc.data_ptr(), ff.data_ptr(), hh.data_ptr() if hh is not None else None, N,
                                               n_dyn.data_ptr() if n_dyn is not None else None, out.data_ptr(), ws.data_ptr(), nb, ops._stream()),
                  "generator_forward")
        return out

    def _forward_host(self, c: torch.Tensor, f0: torch.Tensor, harm: torch.Tensor | None = None, n_dyn: torch.Tensor | None = None) -> torch.Tensor:
        """The same forward, launch by launch from the host (rounds 1-4; kept for the A/B switches and as the reference the one-call
        path is tested against)."""
        dev = c.device
        N = c.shape[0]
        dyn = self._dyn = None if n_dyn is None else (n_dyn, N)
        hop, n_up, uic = self.hop, self.n_up, self.uic
        L = N * hop
        new = lambda r, ch: torch.empty(r, ch, device=dev, dtype=torch.float32)
        # Range slots of the f16x2 GEMMs (include/knnsvc_hip.h, "Range"): one slot per logical tensor.  Every producer folds
        # max|out| into its output's slot (out_absmax), every consumer derives its activation scale from its input's slot
        # (x_absmax) — the generator's activations have no a-priori bound (residual sums over 4 x 9 ResBlock convs, arbitrary
        # trained weights), and this way no range can overflow fp16 and small-amplitude stages keep their bits, with no host
        # round trip (the whole forward stays one hipGraph).
        slots = torch.zeros(256 * ops.SLOT_W, device=dev, dtype=torch.float32)
        n_slot = [0]

        def slot():
            n_slot[0] += 1
            return slots[(n_slot[0] - 1) * ops.SLOT_W:n_slot[0] * ops.SLOT_W]
        # lengths of the time axis at each level of the side path: lens[0] = L ... lens[n_up] = N
        lens = [L]
        for i in range(n_up):
            lens.append(lens[-1] // self.downs[i]["u"])
        assert lens[-1] == N
        # concat buffers: up stage i consumes cat[i] = [ups_i output | res[n_up-1-i]]; one slot per buffer (both producers
        # fold into it; the side path reads its part before the main path has written the other — a smaller bound, still
        # a bound of what it reads)
        cat, cat_slot = [], []
        for i in range(n_up):
            ch = uic // 2 ** (i + 1)
            cat.append(new(lens[n_up - 1 - i], ch + self.side[n_up - 1 - i]))
            cat_slot.append(slot())
        cat_pre, cat_pre_slot = new(N, uic + self.side[n_up]), slot()

        def res_view(level):
            """(tensor view, ld, channels, slot) of res[level] inside its concat buffer."""
            if level == n_up:
                return cat_pre[:, uic:], cat_pre.shape[1], self.side[level], cat_pre_slot
            buf = cat[n_up - 1 - level]
            ch = uic // 2 ** (n_up - level)
            return buf[:, ch:], buf.shape[1], self.side[level], cat_slot[n_up - 1 - level]

        # ---- head of the main path: input projection + conv_pre -> cat_pre[:, :uic] ------------------------
        # (It shares nothing with the side path below but the concat buffer's slot.  Round 3/4 ran it on a stream of its own next
        #  to the side path (5.83 -> 5.70 ms); with the branch streams gone the generator is ONE stream again and the head simply
        #  runs first.)
        s_x0, s_c = slot(), slot()
        x0 = ops.linear(c.contiguous(), self.lin_w, self.lin_b, x_absmax=ops.absmax(c.contiguous(), s_c), out_absmax=s_x0, dyn=dyn)
        self._conv(x0, self.pre_w, cat_pre, T_in=N, cin=x0.shape[1], cout=uic, k=7, m=N, pad=3, bias=self.pre_b,
                   ldo=cat_pre.shape[1], x_absmax=s_x0, out_absmax=cat_pre_slot)
        # ---- excitation + sin_prenet -> res[0] -------------------------------------------------
        cond, ld0, c0, s0 = res_view(0)
        ops.additive_synth(f0.contiguous(), harm.contiguous() if self.kind == "mix" else None, self.prenet_w, self.prenet_b,
                           cond, ld0, hop=hop, sr=self.sr, mode=0 if self.kind == "mix" else 1, n_dyn=n_dyn)
        ops.absmax(cond[:, :c0], s0)
        # ---- side (down) path ----------------------------------------------------------------------
        for i in range(n_up):
            src, ld_s, c_s, s_src = res_view(i)
            dst, ld_d, c_d, s_dst = res_view(i + 1)
            dn = self.downs[i]
            t_in = lens[i]
            t_mid = t_in // dn["u"] + 1            # the conv yields one more row than the crop keeps; the
            mid, s_mid = new(t_mid, c_d), slot()    # k=3 resblock conv still reads it (ddsp_models.py:189-194)
            self._conv(src, dn["w"], mid, T_in=t_in, cin=c_s, cout=c_d, k=dn["k"], m=t_mid, stride=dn["u"],
                       pad=dn["k"] // 2, ldx=ld_s, bias=dn["b"], x_absmax=s_src, out_absmax=s_mid)
            rb = self.rbd[i]
            self._conv(mid, rb["w"], dst, T_in=t_mid, cin=c_d, cout=c_d, k=3, m=lens[i + 1], pad=1, bias=rb["b"],
                       a_slope=LRELU, resid=mid, ldr=c_d, ldo=ld_d, x_absmax=s_mid, out_absmax=s_dst)
        # ---- main path ------------------------------------------------------------------------------
        x, s_x = new(N, uic), slot()
        self._conv(cat_pre, self.cpre_w, x, T_in=N, cin=cat_pre.shape[1], cout=uic, k=3, m=N, pad=1, bias=self.cpre_b,
                   x_absmax=cat_pre_slot, out_absmax=s_x)
        t_cur = N
        for i in range(n_up):
            up = self.ups[i]
            u, k, cout, cin = up["u"], up["k"], up["cout"], up["cin"]
            R = k // u
            t_out = t_cur * u
            assert t_out == cat[i].shape[0]
            ld_c = cat[i].shape[1]
            ops.conv_gemm(x, up["w"], cat[i], m=t_cur + R - 1, n=u * cout, cin=cin, taps=R, stride=1, dil=-1, pad=0,
                          t_in=t_cur, bias=up["b"], bias_period=cout, a_slope=LRELU, ldo=ld_c,
                          convt_u=u, convt_cout=cout, convt_pad=(k - u) // 2, t_out=t_out, x_absmax=s_x, out_absmax=cat_slot[i], dyn=dyn)
            xc, s_xc = new(t_out, cout), slot()
            self._conv(cat[i], self.ccv[i], xc, T_in=t_out, cin=ld_c, cout=cout, k=3, m=t_out, pad=1,
                       x_absmax=cat_slot[i], out_absmax=s_xc)
            xs, s_xs = new(t_out, cout), slot()
            nblk = len(self.res[i])

            def branch(j, blk, out_buf, s_out, accumulate, div):
                """One ResBlock1 (hifigan/ddsp_models.py:13-44): three (dilated conv -> conv + residual) pairs on xc."""
                kr = blk["k"]
                t1, ra, rb_ = new(t_out, cout), new(t_out, cout), new(t_out, cout)
                cur, s_cur = xc, s_xc
                for m, cv in enumerate(blk["convs"]):
                    d = cv["d"]
                    last = m == len(blk["convs"]) - 1
                    if not (last and (accumulate or div != 1.0)) and ops.resblock_pair_ok(cout, kr, d):
                        # both convolutions of the pair in one launch, t1 in LDS only (knnsvc_resblock_pair): the narrow stages
                        dst = out_buf if last else (ra if cur is not ra else rb_)
                        s_dst = s_out if last else slot()
                        ops.resblock_pair(cur, cv["w1"], cv["b1"], cv["w2"], cv["b2"], dst, t=t_out, channels=cout, taps=kr, dil=d,
                                          slope=LRELU, x_absmax=s_cur, t1_bound=cv["t1_bound"], out_absmax=s_dst, dyn=self._dyn)
                        cur, s_cur = dst, s_dst
                        continue
                    # t1 = lrelu(convs1(lrelu(cur)) + b1) is not measured: its consumer bounds it by t1_bound applied to cur's slot
                    # (leaky ReLUs do not grow their argument) — one publishing launch per ResBlock pair instead of two
                    self._conv(cur, cv["w1"], t1, T_in=t_out, cin=cout, cout=cout, k=kr, m=t_out, dil=d,
                               pad=(kr * d - d) // 2, bias=cv["b1"], a_slope=LRELU, act=ops.ACT_LRELU, act_slope=LRELU,
                               x_absmax=s_cur)
                    dst = out_buf if last else (ra if cur is not ra else rb_)
                    s_dst = s_out if last else slot()
                    self._conv(t1, cv["w2"], dst, T_in=t_out, cin=cout, cout=cout, k=kr, m=t_out, pad=(kr - 1) // 2,
                               bias=cv["b2"], resid=cur, ldr=cout,
                               accumulate=(last and accumulate), div=(div if last else 1.0),
                               x_absmax=s_cur, x_bound=cv["t1_bound"], out_absmax=s_dst)
                    cur, s_cur = dst, s_dst

            # Each branch writes its own output; knnsvc_mean3 takes (rb2 + (rb1 + rb0)) / 3 — the association of the reference's
            # running sum (xs += resblock(x); x = xs / num_kernels, ddsp_models.py:218-227) — and publishes the stage's range slot.
            # The three ResBlocks (kernel sizes 3 / 7 / 11) only share their input, and step m of one depends only on step m - 1
            # of the same one: the launches of step m of ALL THREE are one grid (ops.conv_gemm_multi / ops.resblock_pair_multi:
            # blockIdx.y = branch).  That fills the chip where one branch's launch does not — the first stage's convolutions are
            # 470 workgroups for 768 slots — without any stream: rounds 3 and 4 ran the branches on three streams, and whether those
            # overlapped was up to HIP's stream -> hardware-queue mapping (bench: 34.8 or 38.3 ms per step depending on how many
            # streams the process had created before; off altogether under an RCCL group).  Same descriptors as the separate
            # launches (serial_resblocks() runs those): the waveform does not depend on the mode.
            if nblk != 3:                            # other configurations: the running sum in the last epilogues, as round 2
                for j, blk in enumerate(self.res[i]):
                    branch(j, blk, xs, s_xs, j > 0, float(nblk) if j == nblk - 1 else 1.0)
                x, s_x, t_cur = xs, s_xs, t_out
                continue
            outs = [new(t_out, cout) for _ in range(3)]
            if not self._merged():
                for j, blk in enumerate(self.res[i]):
                    branch(j, blk, outs[j], None, False, 1.0)
            else:
                order = sorted(range(3), key=lambda j: -self.res[i][j]["k"])       # most taps first: workgroups are dispatched y-major
                tmp = {j: (new(t_out, cout), new(t_out, cout), new(t_out, cout)) for j in range(3)}      # t1, ra, rb of each branch
                state = {j: (xc, s_xc) for j in range(3)}
                n_steps = len(self.res[i][0]["convs"])
                assert all(len(b["convs"]) == n_steps for b in self.res[i])
                for m in range(n_steps):
                    pairs, c1s, c2s = [], [], []
                    for j in order:
                        blk = self.res[i][j]
                        cv, kr = blk["convs"][m], blk["k"]
                        d = cv["d"]
                        t1, ra, rb_ = tmp[j]
                        cur, s_cur = state[j]
                        last = m == n_steps - 1
                        dst = outs[j] if last else (ra if cur is not ra else rb_)
                        s_dst = None if last else slot()
                        if ops.resblock_pair_ok(cout, kr, d):
                            ops.resblock_pair(cur, cv["w1"], cv["b1"], cv["w2"], cv["b2"], dst, t=t_out, channels=cout, taps=kr, dil=d,
                                              slope=LRELU, x_absmax=s_cur, t1_bound=cv["t1_bound"], out_absmax=s_dst, dyn=self._dyn,
                                              defer=pairs)
                        else:
                            self._conv(cur, cv["w1"], t1, T_in=t_out, cin=cout, cout=cout, k=kr, m=t_out, dil=d,
                                       pad=(kr * d - d) // 2, bias=cv["b1"], a_slope=LRELU, act=ops.ACT_LRELU, act_slope=LRELU,
                                       x_absmax=s_cur, defer=c1s)
                            self._conv(t1, cv["w2"], dst, T_in=t_out, cin=cout, cout=cout, k=kr, m=t_out, pad=(kr - 1) // 2,
                                       bias=cv["b2"], resid=cur, ldr=cout, x_absmax=s_cur, x_bound=cv["t1_bound"], out_absmax=s_dst,
                                       defer=c2s)
                        state[j] = (dst, s_dst)
                    ops.resblock_pair_multi(pairs)
                    ops.conv_gemm_multi(c1s)
                    ops.conv_gemm_multi(c2s)
            ops.mean3(outs[0], outs[1], outs[2], float(nblk), xs, out_absmax=s_xs, dyn=dyn)
            x, s_x, t_cur = xs, s_xs, t_out
        y = new(t_cur, 1)
        self._conv(x, self.post_w, y, T_in=t_cur, cin=x.shape[1], cout=1, k=7, m=t_cur, pad=3, a_slope=0.01,
                   act=ops.ACT_TANH, x_absmax=s_x)
        assert n_slot[0] * ops.SLOT_W <= slots.numel()
        return y.reshape(-1)
