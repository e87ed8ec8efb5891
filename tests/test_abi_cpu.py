"""CPU: the C-ABI library builds, loads, and exports exactly what include/knnsvc_hip.h declares
(no compute calls — there is no GPU here)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "knnsvc_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(knnsvc_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as g
    g.build()
    from knn_svc_amd import _lib
    lib = _lib.load()
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in the header but not exported"
    assert sorted(_lib.SIGNATURES) == names, "ctypes table and header disagree"
    assert lib.knnsvc_abi_version() == _lib.ABI_VERSION
    assert lib.knnsvc_knn_workspace_bytes(1500, 30000, 32) > 0
    assert lib.knnsvc_smooth_workspace_bytes(1500) > 0


def test_conv_desc_layout_matches_header():
    """ctypes.Structure field order/types mirror struct knnsvc_conv_desc (checked by size: 4- and 8-byte
    members with natural alignment give the same total on both sides)."""
    from knn_svc_amd._lib import ConvDesc
    src = open(os.path.join(ROOT, "include", "knnsvc_hip.h")).read()
    body = re.search(r"typedef struct knnsvc_conv_desc \{(.*?)\} knnsvc_conv_desc;", src, re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        m = re.match(r"(const float\*|const void\*|const int32_t\*|float\*|int64_t|int32_t|float)\s+(\w+)$", decl)
        assert m, decl
        fields.append((m.group(2), m.group(1)))
    assert [f for f, _ in fields] == [f for f, _ in ConvDesc._fields_]
    cmap = {"const float*": ctypes.c_void_p, "const void*": ctypes.c_void_p, "const int32_t*": ctypes.c_void_p, "float*": ctypes.c_void_p, "int64_t": ctypes.c_int64,
            "int32_t": ctypes.c_int32, "float": ctypes.c_float}
    for (name, ctype), (pname, ptype) in zip(fields, ConvDesc._fields_):
        assert cmap[ctype] is ptype, (name, ctype, ptype)


def test_ops_refuse_cpu_tensors():
    import pytest
    import torch
    from knn_svc_amd import ops
    from knn_svc_amd._lib import KnnSvcError
    with pytest.raises(KnnSvcError):
        ops.knn_topk(torch.zeros(4, 8), torch.zeros(8, 8), 2)


def test_cli_flags_match_reference_surface():
    from knn_svc_amd.inference import build_parser, output_dir_for
    a = build_parser().parse_args(["s", "t"])
    assert (a.ckpt_type, a.post_opt, a.topk, a.device, a.prioritize_f0, a.tgt_loudness_db, a.dur_limit) == \
        ("mix", "no_post_opt", 4, "cuda", True, -16, None)
    assert output_dir_for("/d/A", "/d/B", "mix", "post_opt_0.2", None) == "/d/A_to_B_mix_post_opt_post_opt_0.2/"
    assert output_dir_for("/d/A", "/d/B", "mix", "no_post_opt", 600).startswith("/d/duration_limit_600_A_to_B")


def test_packers_roundtrip():
    import torch
    import torch.nn.functional as F
    from knn_svc_amd import ops
    g = torch.Generator().manual_seed(0)
    w = torch.randn(6, 4, 5, generator=g)
    x = torch.randn(1, 4, 20, generator=g)
    ref = F.conv1d(x, w, padding=2)[0].T
    wp = ops.pack_conv_weight(w)                                  # [6, 5*4]
    cols = torch.stack([F.pad(x[0].T, (0, 0, 2, 2))[t:t + 5].reshape(-1) for t in range(20)])
    assert torch.allclose(cols @ wp.T, ref, atol=1e-5)
    wt = torch.randn(4, 3, 6, generator=g)                        # ConvTranspose1d [Cin, Cout, k], u = 2 -> 3 taps
    u, pad, R = 2, 2, 3
    yt = F.conv_transpose1d(x, wt, stride=u, padding=pad)[0].T    # [40, 3]
    assert yt.shape[0] == 20 * u
    wpt = ops.pack_convT_weight(wt, u)                            # [u*3, R*4]
    xz = torch.cat([torch.zeros(R, 4), x[0].T, torch.zeros(R, 4)])           # x[q] lives at row q + R
    out = torch.zeros(20 * u, 3)
    for q in range(20 + R - 1):
        a = torch.cat([xz[q + R - r] for r in range(R)])          # taps r: x[q - r]
        row = (wpt @ a).reshape(u, 3)
        for ph in range(u):
            o = q * u + ph - pad
            if 0 <= o < 20 * u:
                out[o] = row[ph]
    assert torch.allclose(out, yt, atol=1e-5)


def test_pool_cache_lru_and_keys(tmp_path):
    import numpy as np
    import torch
    from knn_svc_amd import pool_cache
    a = tmp_path / "a.wav"; a.write_bytes(b"x" * 10); np.save(tmp_path / "a_f0.npy", np.zeros(3))
    k1 = pool_cache.file_key(a, ("enc", 6))
    assert k1 == pool_cache.file_key(a, ("enc", 6)) and k1 != pool_cache.file_key(a, ("enc", 2))
    a.write_bytes(b"x" * 11)
    assert pool_cache.file_key(a, ("enc", 6)) != k1                       # content change -> new identity
    c = pool_cache.PoolCache(budget_bytes=3 * 400)
    ent = lambda v: dict(feats=torch.full((100,), float(v)), f0=None)      # 400 bytes each
    for i in range(3):
        c.put(("f", i), ent(i))
    assert c.get(("f", 0)) is not None                                    # refresh 0 -> 1 is now the oldest
    c.put(("f", 3), ent(3))
    assert c.get(("f", 1)) is None and c.get(("f", 0)) is not None and c.get(("f", 3)) is not None
    assert c.used == 3 * 400 and c.hits == 3 and c.misses == 1
    off = pool_cache.PoolCache(budget_bytes=0)
    off.put(("f", 0), ent(0))
    assert off.get(("f", 0)) is None


def test_pool_store_disk_tier(tmp_path):
    """On-disk tier of the pool-feature store: written atomically per file, found again by a fresh store,
    invalidated by any change of the key, and a damaged file is a miss, not an error."""
    import numpy as np
    import torch
    from knn_svc_amd import pool_cache
    wav = tmp_path / "a.wav"; wav.write_bytes(b"RIFF" + bytes(100))
    np.save(tmp_path / "a_f0.npy", np.zeros(3, np.float32))
    key = pool_cache.file_key(wav, ("uid1", 6))
    dkey = pool_cache.file_key(wav, ("abcdef", 6))
    val = {f: torch.arange(12, dtype=torch.float32).reshape(3, 4) + i for i, f in enumerate(pool_cache.FIELDS)}
    c1 = pool_cache.PoolCache(budget_bytes=1 << 20, disk_dir=str(tmp_path / "store"))
    assert c1.get(key, dkey) is None and c1.misses == 1
    c1.put(key, val, dkey)
    assert c1.disk_writes == 1 and len(list((tmp_path / "store").glob("*.npz"))) == 1
    c2 = pool_cache.PoolCache(budget_bytes=1 << 20, disk_dir=str(tmp_path / "store"))      # "another process"
    got = c2.get(pool_cache.file_key(wav, ("uid7", 6)), dkey)
    assert got is not None and c2.disk_hits == 1 and all(torch.equal(got[f], val[f]) for f in pool_cache.FIELDS)
    assert c2.get(pool_cache.file_key(wav, ("uid7", 6))) is got                          # promoted to the resident tier
    assert c2.get(("x",), pool_cache.file_key(wav, ("other-weights", 6))) is None         # other weights: miss
    np.save(tmp_path / "a_f0.npy", np.zeros(4, np.float32))                               # f0 track changed: miss
    assert c2.get(("y",), pool_cache.file_key(wav, ("abcdef", 6))) is None
    pth = next((tmp_path / "store").glob("*.npz")); pth.write_bytes(b"garbage")
    c3 = pool_cache.PoolCache(budget_bytes=0, disk_dir=str(tmp_path / "store"))
    assert c3.get(("z",), dkey) is None


def test_struct_layouts_match_the_header_as_a_c_compiler_sees_it(tmp_path):
    """Every struct that crosses the boundary: sizeof and the offset of its last member as gcc lays the header out == what the
    ctypes mirror in knn_svc_amd/_lib.py uses (a wrong field order or a missing pad shows up here, not as a corrupted launch)."""
    import subprocess
    from knn_svc_amd import _lib
    pairs = [("knnsvc_conv_desc", _lib.ConvDesc), ("knnsvc_pair_desc", _lib.PairDesc), ("knnsvc_weight", _lib.Weight),
             ("knnsvc_wavlm_conv", _lib.WavlmConv), ("knnsvc_wavlm_layer", _lib.WavlmLayer), ("knnsvc_wavlm_desc", _lib.WavlmDesc),
             ("knnsvc_gen_pair", _lib.GenPair), ("knnsvc_gen_stage", _lib.GenStage), ("knnsvc_generator_desc", _lib.GeneratorDesc)]
    src = '#include <stdio.h>\n#include <stddef.h>\n#include "knnsvc_hip.h"\nint main(void) {\n'
    for cname, st in pairs:
        last = st._fields_[-1][0]
        src += f'  printf("{cname} %zu %zu\\n", sizeof({cname}), offsetof({cname}, {last}));\n'
    src += "  return 0;\n}\n"
    (tmp_path / "l.c").write_text(src)
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(tmp_path / "l.c"), "-o", str(tmp_path / "l")], check=True)
    out = subprocess.run([str(tmp_path / "l")], check=True, capture_output=True, text=True).stdout.split("\n")
    for line, (cname, st) in zip(out, pairs):
        name, size, off = line.split()
        assert name == cname and int(size) == ctypes.sizeof(st) and int(off) == getattr(st, st._fields_[-1][0]).offset, (line, ctypes.sizeof(st))


def test_wavlm_handle_host_side_without_a_gpu():
    """knnsvc_wavlm_create / frames / workspace_bytes / free are host code: the frame law of the conv stack (G9: 480 320 padded
    samples -> 1500 frames) and a workspace plan come out of the handle without touching a device; encode refuses a workspace
    that is too small before it launches anything."""
    from knn_svc_amd import _lib, config as C
    lib = _lib.load()
    convs = (_lib.WavlmConv * 7)()
    for i, (dim, k, s) in enumerate(C.conv_layers(C.WAVLM_LARGE)):
        convs[i].dim, convs[i].k, convs[i].stride, convs[i].cin, convs[i].out_split = dim, k, s, (1 if i == 0 else 512), 1
        convs[i].w.w = 8; convs[i].ln_g = 8; convs[i].ln_b = 8          # (never dereferenced on the host)
    layers = (_lib.WavlmLayer * 6)()
    d = _lib.WavlmDesc()
    d.n_conv, d.n_layers, d.conv, d.layers = 7, 6, convs, layers
    d.ln_g = d.ln_b = 8; d.proj.w = 8; d.pos.w = 8; d.pos_groups, d.pos_k = 16, 128
    d.E, d.H, d.ffn = 1024, 16, 4096
    h = ctypes.c_void_p()
    assert lib.knnsvc_wavlm_create(ctypes.byref(d), ctypes.byref(h)) == 0 and h.value
    assert lib.knnsvc_wavlm_frames(h, 480320) == 1500 and lib.knnsvc_wavlm_frames(h, 16000 + 320) == 50
    one, eight = lib.knnsvc_wavlm_workspace_bytes(h, 1, 480320), lib.knnsvc_wavlm_workspace_bytes(h, 8, 480320)
    assert one > 96063 * 512 * 4 and 7 * one < eight < 8.5 * one             # the first conv layer's output dominates; linear in the batch
    assert lib.knnsvc_wavlm_workspace_bytes(h, 1, 300) == 0                   # shorter than the receptive field
    rc = lib.knnsvc_wavlm_encode(h, 256, 1, 480320, None, 256, 256, 256, 1024, None)
    assert rc != 0 and b"workspace" in lib.knnsvc_last_error()
    d.H = 15
    h2 = ctypes.c_void_p()
    assert lib.knnsvc_wavlm_create(ctypes.byref(d), ctypes.byref(h2)) != 0 and b"head_dim" in lib.knnsvc_last_error()
    assert lib.knnsvc_wavlm_free(h) == 0
