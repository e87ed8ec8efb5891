"""Oracle: conditioned HiFi-GAN generator (reference hifigan/ddsp_models.py:13-44,
81-94, 108-233, 405-493 'mix'; hifigan/ddsp_models_f0.py:106-216, 320-381 'f0').
Flat state dict with the reference's names; weight norm folded with
torch._weight_norm exactly as the live parametrisation does.  Test infrastructure only."""
from __future__ import annotations

import torch
import torch.nn.functional as F

from .synth_ref import additive_synth, sine_excitation

LRELU = 0.1


def _w(sd, name):
    if name + ".weight" in sd:
        return sd[name + ".weight"]
    return torch._weight_norm(sd[name + ".weight_v"], sd[name + ".weight_g"], 0)


def _b(sd, name):
    return sd.get(name + ".bias")


def _pad(k, d=1):
    return int((k * d - d) / 2)


def resblock1(sd, name, x, k, dil):
    """3 x [lrelu -> dilated conv -> lrelu -> conv -> +x] (ddsp_models.py:37-44)."""
    for m, d in enumerate(dil):
        xt = F.leaky_relu(x, LRELU)
        xt = F.conv1d(xt, _w(sd, f"{name}.convs1.{m}"), _b(sd, f"{name}.convs1.{m}"), padding=_pad(k, d), dilation=d)
        xt = F.leaky_relu(xt, LRELU)
        xt = F.conv1d(xt, _w(sd, f"{name}.convs2.{m}"), _b(sd, f"{name}.convs2.{m}"), padding=_pad(k, 1))
        x = xt + x
    return x


def resblock3(sd, name, x):
    """one [lrelu -> conv k3 d1 -> +x] (ddsp_models.py:81-94: only dilation[0] is built)."""
    xt = F.leaky_relu(x, LRELU)
    xt = F.conv1d(xt, _w(sd, f"{name}.convs.0"), _b(sd, f"{name}.convs.0"), padding=1)
    return xt + x


def generator(sd, h, c, cond):
    """Generator.forward (ddsp_models.py:176-233).  c [B,N,hubert], cond [B,Cc,N*hop] -> [B,1,N*hop]."""
    rates, ksz = h["upsample_rates"], h["upsample_kernel_sizes"]
    n_up = len(rates)
    x = F.linear(c, sd["dec.lin_pre.weight"], sd["dec.lin_pre.bias"]).permute(0, 2, 1)
    x = F.conv1d(x, sd["dec.conv_pre.weight"], sd["dec.conv_pre.bias"], padding=3)
    se = cond
    res = [se]
    for i in range(n_up):
        j = n_up - 1 - i
        n_in = se.size(2)
        se = F.conv1d(se, _w(sd, f"dec.downs.{i}"), _b(sd, f"dec.downs.{i}"), stride=rates[j], padding=ksz[j] // 2)
        se = resblock3(sd, f"dec.resblocks_downs.{i}", se)
        se = se[:, :, : n_in // rates[j]]
        res.append(se)
    x = torch.cat([x, se], 1)
    x = F.conv1d(x, sd["dec.concat_pre.weight"], sd["dec.concat_pre.bias"], padding=1)
    nk = len(h["resblock_kernel_sizes"])
    for i in range(n_up):
        x = F.leaky_relu(x, LRELU)
        x = F.conv_transpose1d(x, _w(sd, f"dec.ups.{i}"), _b(sd, f"dec.ups.{i}"), stride=rates[i],
                               padding=(ksz[i] - rates[i]) // 2)
        x = torch.cat([x, res[n_up - 1 - i]], 1)
        x = F.conv1d(x, sd[f"dec.concat_conv.{i}.weight"], None, padding=1)
        xs = None
        for j, (k, d) in enumerate(zip(h["resblock_kernel_sizes"], h["resblock_dilation_sizes"])):
            r = resblock1(sd, f"dec.resblocks.{i * nk + j}", x, k, d)
            xs = r if xs is None else xs + r
        x = xs / nk
    x = F.leaky_relu(x)                    # default slope 0.01 (ddsp_models.py:229)
    x = F.conv1d(x, sd["dec.conv_post.weight"], None, padding=3)
    return torch.tanh(x)


@torch.inference_mode()
def synthesizer(sd, h, kind, c, f0, harm=None):
    """SynthesizerTrn.forward.  c [B,N,hubert], f0 [B,N,1], harm [B,N,49] (mix only) -> [B,1,N*hop]."""
    if kind == "mix":
        exc = additive_synth(f0, harm, h["sampling_rate"], h["hop_size"]).transpose(1, 2)
    else:
        exc = sine_excitation(f0, h["sampling_rate"], h["hop_size"])
    cond = F.conv1d(exc, sd["sin_prenet.weight"], sd["sin_prenet.bias"], padding=1)
    return generator(sd, h, c, cond)
