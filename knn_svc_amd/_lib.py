"""ctypes binding of libknnsvc_hip.so (the C ABI in include/knnsvc_hip.h).

There is deliberately no fallback: if the library is missing or a symbol is
absent, importing an op raises.  The library is built in-tree by
``__graft_entry__.build()`` / ``make -C knn_svc_amd/csrc``.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("KNNSVC_LIB") or os.path.join(_HERE, "libknnsvc_hip.so")      # KNNSVC_LIB: an A/B build (csrc/Makefile)
ABI_VERSION = 17

vp, i32, i64, f32, sz = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_size_t


class ConvDesc(C.Structure):
    """struct knnsvc_conv_desc (field order and types must match the header)."""
    _fields_ = [
        ("x", vp), ("x_bstride", i64), ("x_gstride", i64), ("ldx", i32), ("t_in", i32),
        ("cin", i32), ("taps", i32), ("stride", i32), ("dil", i32), ("pad", i32),
        ("a_slope", f32),
        ("w", vp), ("w_gstride", i64), ("n", i32),
        ("bias", vp), ("bias_gstride", i64), ("bias_period", i32),
        ("out", vp), ("o_bstride", i64), ("o_gstride", i64), ("ldo", i32), ("m", i32),
        ("act", i32), ("act_slope", f32),
        ("resid", vp), ("r_bstride", i64), ("r_gstride", i64), ("ldr", i32),
        ("accumulate", i32), ("div", f32),
        ("batches", i32), ("groups", i32),
        ("convt_u", i32), ("convt_cout", i32), ("convt_pad", i32), ("t_out", i32),
        ("w_bf16x3", vp),
        ("w_f16x2", vp), ("w_f16x2_scale", f32), ("a_f16x2_scale", f32),
        ("x_f16x2", i32), ("out_f16x2", i32),
        ("x_absmax", vp), ("w_absmax", vp), ("out_absmax", vp), ("out_f16x2_scale", f32),
        ("n_dyn", vp), ("dyn_t_in_mul", i32), ("dyn_t_in_add", i32), ("dyn_m_mul", i32), ("dyn_m_add", i32), ("dyn_t_out_mul", i32),
        ("x_bound_mul", f32), ("x_bound_add", f32),
        ("fixed_tile", i32),
    ]


class PairDesc(C.Structure):
    """struct knnsvc_pair_desc (field order and types must match the header)."""
    _fields_ = [
        ("x", vp), ("ldx", i32), ("t", i32), ("channels", i32), ("taps", i32), ("dil", i32),
        ("w1_f16x2", vp), ("w1_scale", f32), ("b1", vp),
        ("w2_f16x2", vp), ("w2_scale", f32), ("b2", vp),
        ("out", vp), ("ldo", i32),
        ("slope", f32),
        ("x_absmax", vp), ("t1_bound_mul", f32), ("t1_bound_add", f32), ("a1_scale", f32), ("a2_scale", f32),
        ("out_absmax", vp),
        ("n_dyn", vp), ("dyn_mul", i32),
    ]


class Weight(C.Structure):
    """struct knnsvc_weight"""
    _fields_ = [("w", vp), ("w_f16x2", vp), ("w_f16x2_scale", f32), ("pad_", i32)]


class WavlmConv(C.Structure):
    """struct knnsvc_wavlm_conv"""
    _fields_ = [("w", Weight), ("ln_g", vp), ("ln_b", vp), ("dim", i32), ("k", i32), ("stride", i32), ("cin", i32),
                ("out_split", i32), ("pad_", i32)]


class WavlmLayer(C.Structure):
    """struct knnsvc_wavlm_layer"""
    _fields_ = [("ln1_g", vp), ("ln1_b", vp), ("ln2_g", vp), ("ln2_b", vp),
                ("wqkv", Weight), ("bqkv", vp), ("wo", Weight), ("bo", vp), ("w1", Weight), ("b1", vp), ("w2", Weight), ("b2", vp),
                ("gate_w", vp), ("gate_b", vp), ("grep_a", vp),
                ("xn_split", i32), ("xn2_split", i32), ("h_split", i32), ("attn_f16", i32)]


class WavlmDesc(C.Structure):
    """struct knnsvc_wavlm_desc"""
    _fields_ = [("n_conv", i32), ("n_layers", i32), ("conv", C.POINTER(WavlmConv)), ("layers", C.POINTER(WavlmLayer)),
                ("ln_g", vp), ("ln_b", vp), ("feats_split", i32), ("pad_", i32),
                ("proj", Weight), ("proj_b", vp),
                ("pos", Weight), ("pos_b", vp), ("pos_groups", i32), ("pos_k", i32),
                ("pos_a_scale", f32), ("E", i32), ("H", i32), ("ffn", i32),
                ("layer_mix", C.POINTER(f32))]


class GenPair(C.Structure):
    """struct knnsvc_gen_pair"""
    _fields_ = [("w1", Weight), ("b1", vp), ("w2", Weight), ("b2", vp), ("dil", i32), ("t1_bound_mul", f32), ("t1_bound_add", f32), ("pad_", i32)]


class GenStage(C.Structure):
    """struct knnsvc_gen_stage"""
    _fields_ = [("up", Weight), ("up_b", vp), ("u", i32), ("k", i32), ("cin", i32), ("cout", i32),
                ("ccv", Weight),
                ("res_k", i32 * 3), ("pad_", i32), ("res", (GenPair * 3) * 3),
                ("down", Weight), ("down_b", vp), ("down_k", i32), ("down_u", i32),
                ("rbd", Weight), ("rbd_b", vp)]


class GeneratorDesc(C.Structure):
    """struct knnsvc_generator_desc"""
    _fields_ = [("kind", i32), ("n_up", i32), ("hop", i32), ("sample_rate", i32), ("n_harm_in", i32), ("uic", i32), ("hubert_dim", i32), ("hifi_dim", i32),
                ("side", i32 * 8),
                ("lin", Weight), ("lin_b", vp), ("pre", Weight), ("pre_b", vp),
                ("cpre", Weight), ("cpre_b", vp), ("post", Weight),
                ("prenet_w", vp), ("prenet_b", vp),
                ("stages", C.POINTER(GenStage))]


# name -> (restype, argtypes); every symbol the header declares
SIGNATURES = {
    "knnsvc_abi_version": (i32, []),
    "knnsvc_last_error": (C.c_char_p, []),
    "knnsvc_conv_gemm_last_kernel": (C.c_char_p, []),
    "knnsvc_conv_gemm_last_epilogue": (C.c_char_p, []),
    "knnsvc_conv_gemm": (i32, [C.POINTER(ConvDesc), vp]),
    "knnsvc_conv_gemm_multi": (i32, [C.POINTER(ConvDesc), i32, vp]),
    "knnsvc_mean3": (i32, [vp, vp, vp, i64, f32, vp, vp, vp, i64, vp]),
    "knnsvc_axpy": (i32, [vp, i64, f32, i32, vp, vp]),
    "knnsvc_resblock_pair": (i32, [C.POINTER(PairDesc), vp]),
    "knnsvc_resblock_pair_multi": (i32, [C.POINTER(PairDesc), i32, vp]),
    "knnsvc_split_weight_bf16x3": (i32, [vp, i64, i32, vp, vp]),
    "knnsvc_split_weight_f16x2": (i32, [vp, i64, i32, f32, vp, vp]),
    "knnsvc_split_f16x2_dyn": (i32, [vp, i64, i32, vp, vp, vp]),
    "knnsvc_absmax": (i32, [vp, i64, i32, i32, vp, vp]),
    "knnsvc_wavlm_create": (i32, [C.POINTER(WavlmDesc), C.POINTER(vp)]),
    "knnsvc_wavlm_free": (i32, [vp]),
    "knnsvc_wavlm_frames": (i64, [vp, i64]),
    "knnsvc_wavlm_workspace_bytes": (sz, [vp, i32, i64]),
    "knnsvc_wavlm_encode": (i32, [vp, vp, i32, i64, vp, vp, vp, vp, sz, vp]),
    "knnsvc_generator_create": (i32, [C.POINTER(GeneratorDesc), C.POINTER(vp)]),
    "knnsvc_generator_free": (i32, [vp]),
    "knnsvc_generator_workspace_bytes": (sz, [vp, i64]),
    "knnsvc_generator_forward": (i32, [vp, vp, vp, vp, i64, vp, vp, vp, sz, vp]),
    "knnsvc_layernorm": (i32, [vp, i64, i32, i32, vp, vp, i32, vp, i32, vp]),
    "knnsvc_wavlm_conv0": (i32, [vp, i32, i64, vp, i32, i32, i32, vp, vp, vp, i32, vp]),
    "knnsvc_wavlm_gate": (i32, [vp, i64, i32, i32, i32, vp, vp, vp, vp, i32, vp]),
    "knnsvc_wavlm_attention": (i32, [vp, vp, vp, i32, i32, i32, vp, i32, i32, vp, vp]),
    "knnsvc_mask_rows": (i32, [vp, i32, i32, i32, i32, vp, vp]),
    "knnsvc_row_norms": (i32, [vp, i64, i32, i32, vp, vp, vp, vp]),
    "knnsvc_knn_workspace_bytes": (sz, [i64, i64, i32]),
    "knnsvc_knn_topk": (i32, [vp, vp, vp, i64, vp, vp, vp, i64, i32, i32, i64, i64, i64, vp, vp, vp, sz, vp, i32, vp]),
    "knnsvc_knn_rescore": (i32, [vp, i64, i32, vp, vp, vp, vp, vp, vp, i64, i32, i64, i64, i64, i32, vp, vp, vp]),
    "knnsvc_knn_select": (i32, [vp, i64, vp, vp, i64, vp, vp, i64, i32, i64, i64, vp, vp, vp]),
    "knnsvc_knn_merge": (i32, [vp, vp, i32, i64, i32, vp, vp, vp]),
    "knnsvc_knn_screen": (i32, [vp, vp, vp, vp, i64, vp, vp, vp, vp, i64, i32, vp, vp, i64, i64, i64, vp, vp, i32, vp, vp, i32, vp]),
    "knnsvc_knn_refine": (i32, [vp, vp, i32, i64, i32, vp, i32, vp, vp, vp, i32, vp, vp]),
    "knnsvc_log_f0_median": (i32, [vp, i64, vp, vp, vp]),
    "knnsvc_shift_f0": (i32, [vp, i64, vp, vp, vp, vp]),
    "knnsvc_reload_knobs": (i32, []),
    "knnsvc_probe_dispatch": (i32, [i32, i32, vp]),
    "knnsvc_f0_rerank": (i32, [vp, i64, i32, vp, vp, vp, vp]),
    "knnsvc_concat_reselect": (i32, [vp, vp, vp, i64, vp, vp, i64, i32, vp, vp, i32, f32, vp, vp]),
    "knnsvc_smooth_workspace_bytes": (sz, [i64]),
    "knnsvc_smooth_weights": (i32, [vp, i64, vp, i64, i32, i32, f32, vp, i32, vp, vp, vp, sz, vp]),
    "knnsvc_weighted_gather": (i32, [vp, vp, i64, i32, vp, i32, i32, i32, vp, vp]),
    "knnsvc_round_f16": (i32, [vp, i64, vp, vp]),
    "knnsvc_amp_ratio": (i32, [vp, i32, vp, i32, i64, vp, i64, i32, i32, vp, vp]),
    "knnsvc_flac_info": (i32, [vp, i64, vp, vp, vp, vp, vp]),
    "knnsvc_flac_decode": (i32, [vp, i64, vp, i64, vp]),
    "knnsvc_flac_encode": (i32, [vp, i32, i64, i32, i32, vp, vp, i64, vp]),
    "knnsvc_f0_harvest_workspace": (i32, [i64, i32, f32, f32, f32, vp, vp]),
    "knnsvc_f0_harvest": (i32, [vp, i64, i32, f32, f32, f32, f32, vp, i64, vp, i64, vp, vp]),
    "knnsvc_reflect_pad": (i32, [vp, i64, i32, vp, vp]),
    "knnsvc_reflect_pad_batch": (i32, [vp, vp, i32, i32, vp, i64, vp]),
    "knnsvc_spec_harm": (i32, [vp, i64, i32, i32, vp, i32, vp, vp, vp]),
    "knnsvc_complex_mag": (i32, [vp, i64, i32, i32, vp, vp]),
    "knnsvc_harmonic_amps": (i32, [vp, vp, i64, i32, i32, vp, vp]),
    "knnsvc_additive_synth": (i32, [vp, vp, i64, i32, i32, i32, i32, vp, vp, i32, vp, i32, vp, vp, vp, vp]),
}

_lib = None


class KnnSvcError(RuntimeError):
    pass


def load():
    """Load (once) and type the shared library; raises if it is absent — no CPU fallback exists."""
    global _lib
    if _lib is not None:
        return _lib
    # torch must load ITS bundled HIP runtime first: libknnsvc_hip.so then binds to that same
    # libamdhip64.so.7 (matched by SONAME).  Loaded the other way round the process ends up with two
    # HIP runtimes and torch's device pointers / streams mean nothing to ours.
    import torch  # noqa: F401
    if not os.path.isfile(LIB_PATH):
        raise KnnSvcError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  knn_svc_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the .so does not export a declared symbol
        fn.restype, fn.argtypes = res, args
    got = lib.knnsvc_abi_version()
    if got != ABI_VERSION:
        raise KnnSvcError(f"libknnsvc_hip ABI {got} != expected {ABI_VERSION}")
    _lib = lib
    return lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = load().knnsvc_last_error().decode("utf-8", "replace")
        raise KnnSvcError(f"libknnsvc_hip {what} failed (code {rc}): {msg}")
