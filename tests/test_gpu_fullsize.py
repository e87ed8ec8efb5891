"""GPU: BASELINE-size runs checked through size-independent properties and sampled oracle rows, plus the
edge cases of the path (ragged chunks, tiny pools, unvoiced tracks, single frames)."""
import numpy as np
import pytest
import torch

from knn_svc_amd import config as C, synthetic as S

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _smooth(x):
    return (x + torch.roll(x, 1, 0) + torch.roll(x, 2, 0)) / 3


@pytest.fixture(scope="module")
def north_star_features():
    """1500 query / 30000 pool rows of 1024-d features (BASELINE north-star point), temporally smooth."""
    q = _smooth(S.clustered_features(1500, 1024, 1, n_centres=80))
    p = _smooth(S.clustered_features(30000, 1024, 2, n_centres=80))
    _, f0 = S.synth_clip(30000 * 320, 3); pf0 = torch.from_numpy(f0[:30000].copy())
    _, f0 = S.synth_clip(1500 * 320, 4); qf0 = torch.from_numpy(f0[:1500].copy() * 1.2)
    harm = torch.rand(30000, 49, generator=torch.Generator().manual_seed(5)) * 0.05
    return q, p, qf0, pf0, harm


@pytest.mark.parametrize("mode", ["f16x2", "fp32", "f16x2-norescore"])
def test_knn_north_star_size_against_the_reference_fixture(golden, monkeypatch, mode):
    """Fixture G3c: the reference's own top-32 (fast_cosine_dist + torch.topk, 20 query rows at a time) at 1500 x 30 000 — the size
    the fused route (epochs of knn_screen + knn_refine, no [Nq, Np] matrix) was built for.  Since round 5 the order of a list comes
    from EXACT dot products (knnsvc_knn_rescore: fp64 accumulation, rounded once, then the reference's formula), so this path is
    the quiet side: its distances sit closer to the exact ones than the reference's own, and every disagreement with the
    reference's lists must sit inside the two roundings together.  Both routes (f16x2 screening on the matrix cores, the fp32-MFMA
    tile kernel) feed the same re-score and must give the SAME lists; the third case switches the re-score off and documents what
    it buys (round 4's numbers)."""
    from knn_svc_amd import ops
    g = golden("g3c_knn_north_star")
    q = S.clustered_features(int(g["nq"]), 1024, int(g["q_seed"]))
    p = S.clustered_features(int(g["np_"]), 1024, int(g["p_seed"]))
    monkeypatch.setenv("KNNSVC_KNN", "fp32" if mode == "fp32" else "f16x2")
    monkeypatch.setenv("KNNSVC_KNN_RESCORE", "0" if mode.endswith("norescore") else "1")
    fused0, dot0 = ops.KNN_ROUTE_COUNTS["fused"], ops.KNN_ROUTE_COUNTS["dot"]
    idx, dist = ops.knn_topk(q.to(DEV), p.to(DEV), 32)
    if mode != "fp32":
        assert ops.KNN_ROUTE_COUNTS["fused"] > fused0 and ops.KNN_ROUTE_COUNTS["dot"] == dot0      # the route under test
    idx, dist = idx.cpu(), dist.cpu()
    ref_idx = torch.from_numpy(g["idx"]).long()

    def exact(ix):            # fp64 cosine distances of the listed pairs only (the full 1500 x 30 000 fp64 matrix costs a minute of CPU)
        qd, pr = q.double(), p.double()[ix]                                      # [nq, 32, 1024]
        return (1.0 - torch.einsum("qd,qkd->qk", qd, pr) / (qd.norm(dim=1)[:, None] * pr.norm(dim=2))).numpy()
    da, db = exact(ref_idx), exact(idx)
    # how far each side's fp32 distances sit from the exact ones (on its own list): the reference's formula (cdist through a
    # matrix product, clamp, sqrt, ...) is only defined up to that, and so is the ORDER of neighbours closer together than it
    err_ref = float(np.abs(da - g["dist"].astype(np.float64)).max())
    err_gpu = float(np.abs(db - dist.numpy().astype(np.float64)).max())
    a_, b_ = ref_idx.numpy(), idx.numpy()
    st = dict(top4=float(np.mean(np.all(a_[:, :4] == b_[:, :4], axis=1))), allk=float(np.mean(np.all(a_ == b_, axis=1))),
              sets=float(np.mean([set(a_[i]) == set(b_[i]) for i in range(len(a_))])), max_gap=float(np.abs(da - db).max()),
              unexplained=int(np.sum(np.abs(da - db) > err_ref + err_gpu)))          # knn_ref.topk_agreement on the listed pairs
    print(f"kNN at the north-star size vs the reference fixture [{mode}]: {st}; largest |fp32 - exact| distance: reference {err_ref:.2e}, here {err_gpu:.2e}")
    assert st["unexplained"] == 0, st
    assert float((dist - torch.from_numpy(g["dist"])).abs().max()) < 5e-6
    assert bool((dist[:, 1:] >= dist[:, :-1]).all())
    if mode.endswith("norescore"):
        # round 4, measured (deterministic): the reference's distances are up to 6.8e-7 off the exact ones, the screening products' up
        # to 1.22e-6 (96 fp32 roundings along K); ordered lists differ in 5.6 % of the rows, the sets in 1 of 1500, the first four in 2
        assert err_gpu <= 2.0 * err_ref, (err_gpu, err_ref)
        assert st["top4"] >= 0.998 and st["sets"] >= 0.999 and st["allk"] >= 0.93 and st["max_gap"] <= 1.2e-6, st
        return
    # with the exact re-score this path is the quiet side.  Measured (r05, deterministic): this path's distances are
    # 2.1e-7 from exact (the formula's own roundings on a correctly rounded dot product), the reference's 6.8e-7; sets equal in
    # 1500 of 1500 rows (r04: 1499), first four in 1499 (1498), ordered top-32 in 96.87 % (94.4 %) — the largest EXACT gap
    # between two swapped neighbours is 2.9e-7 (r04: 1.03e-6): what is left is decided by the reference's own BLAS rounding
    assert err_gpu <= 0.5 * err_ref, (err_gpu, err_ref)
    assert st["top4"] >= 0.9993 and st["sets"] == 1.0 and st["allk"] >= 0.968 and st["max_gap"] <= 3.5e-7, st
    TestKnnModesAgree.lists[mode] = (idx, dist)
    if len(TestKnnModesAgree.lists) == 2:          # both screening routes end in the same exact re-score: identical lists, bit for bit
        (ia, da_), (ib, db_) = TestKnnModesAgree.lists["f16x2"], TestKnnModesAgree.lists["fp32"]
        same = float((ia == ib).all(1).float().mean())
        print(f"f16x2-screened vs fp32-screened lists after the exact re-score: {same:.5f} of the rows identical")
        assert same == 1.0 and torch.equal(da_, db_)


class TestKnnModesAgree:
    lists = {}


def test_selection_chain_north_star_size_against_the_reference_fixture(golden):
    """Fixture G4c (the reference itself at 1500 x 30 000, tests/gen_golden.py select_ns): given the reference's neighbour lists,
    the f0 shift, the stable f0 re-rank and BOTH frame-sequential concat re-selections reproduce the reference's output on every
    one of the 1500 frames; the search on these (temporally smooth) features agrees with the reference's lists up to fp32
    rounding gaps (see test_knn_north_star_size_against_the_reference_fixture)."""
    from knn_svc_amd import ops
    from tests.gen_golden_inputs import north_star_inputs
    g = golden("g4c_select_north_star")
    q, p, qf0, pf0 = north_star_inputs()
    qd, pd, qf0d, pf0d = q.to(DEV), p.to(DEV), qf0.to(DEV), pf0.to(DEV)
    nn32 = torch.from_numpy(g["nn32"]).long().to(DEV)
    sh = ops.shift_f0(qf0d, ops.log_f0_median(qf0d), ops.log_f0_median(pf0d))
    assert float((sh.cpu() - torch.from_numpy(g["shifted"])).abs().max() / torch.from_numpy(g["shifted"]).abs().max()) < 5e-6
    shifted = torch.from_numpy(g["shifted"]).to(DEV)                  # from here on the reference's own values
    rk = ops.f0_rerank(nn32, shifted, pf0d)
    assert float((rk[:, :4].cpu() == torch.from_numpy(g["ranked4"]).long()).all(1).float().mean()) == 1.0
    qn, _ = ops.row_norms(qd); pn, _ = ops.row_norms(pd)
    a = ops.concat_reselect(nn32[:, :4].contiguous(), qd, qn, pd, pn, concat_weight=0.2)
    b = ops.concat_reselect(torch.from_numpy(g["ranked4"]).long().to(DEV), qd, qn, pd, pn, shifted, pf0d, concat_weight=0.2)
    ma = float((a.cpu() == torch.from_numpy(g["sel_plain"]).long()).all(1).float().mean())
    mb = float((b.cpu() == torch.from_numpy(g["sel_f0"]).long()).all(1).float().mean())
    idx, _ = ops.knn_topk(qd, pd, 32)
    top4 = float((idx[:, :4].cpu() == torch.from_numpy(g["nn32"]).long()[:, :4]).all(1).float().mean())
    print(f"north-star selection chain vs the reference: concat plain {ma:.4f}, pitched {mb:.4f} of 1500 frames; search top-4 rows equal {top4:.4f}")
    assert ma == 1.0 and mb == 1.0
    assert top4 >= 0.995


def test_smooth_weights_north_star_size_against_the_reference_fixture(golden):
    """Fixture G5c: the reference's compute_wavlm_weight / compute_extended_weight on 1500 frames (its own re-selected neighbours,
    30 000-row pools).  Same iteration counts; weights within the bar of the small fixture (g5)."""
    from knn_svc_amd import ops
    from tests.gen_golden_inputs import north_star_inputs
    g, g4 = golden("g5c_smooth_north_star"), golden("g4c_select_north_star")
    _q, p, _qf0, _pf0 = north_star_inputs()
    w, it = ops.smooth_weights(torch.from_numpy(g4["sel_plain"]).long().to(DEV), p.to(DEV), 0.1, return_iters=True)
    e = float((w.cpu() - torch.from_numpy(g["w_wavlm"])).abs().max())
    ph = torch.rand(30000, 49, generator=torch.Generator().manual_seed(int(g["harm_seed"]))) * 0.05
    wh, ith = ops.smooth_weights(torch.from_numpy(g4["sel_f0"]).long().to(DEV), ph.to(DEV), 1000.0, return_iters=True)
    eh = float((wh.cpu() - torch.from_numpy(g["w_harm"])).abs().max())
    print(f"Adam loops at the north-star size vs the reference: WavLM weights max|d| {e:.2e}, iterations {int(it)} (reference {int(g['iters_wavlm'])}); "
          f"harmonics max|d| {eh:.2e}, iterations {int(ith)} (reference {int(g['iters_harm'])})")
    assert int(it) == int(g["iters_wavlm"]) and int(ith) == int(g["iters_harm"])
    assert e < 4e-4 and eh < 4e-4


def test_knn_full_size_properties(north_star_features):
    from knn_svc_amd import ops
    from oracle import knn_ref
    q, p, *_ = north_star_features
    qd, pd = q.to(DEV), p.to(DEV)
    idx, dist = ops.knn_topk(qd, pd, 32)
    idx_c, dist_c = idx.cpu(), dist.cpu()
    # sortedness, range, uniqueness
    assert bool((dist_c[:, 1:] >= dist_c[:, :-1]).all())
    assert int(idx_c.min()) >= 0 and int(idx_c.max()) < 30000
    assert all(len(set(r.tolist())) == 32 for r in idx_c[::50])
    # sampled rows against the oracle (the reference's own formula)
    rows = torch.arange(0, 1500, 47)
    ref_i, ref_d = knn_ref.knn_topk(q[rows], p, 32)
    st = knn_ref.topk_agreement(ref_i, idx_c[rows], knn_ref.cosine_dist_f64(q[rows], p), tau=5e-7)
    print("full-size kNN, 32 sampled rows:", st)
    # ratcheted to the measured values (r03: every sampled row equals the oracle's ordered top-32, largest inversion 0)
    assert st["unexplained"] == 0 and st["top4"] == 1.0 and st["sets"] == 1.0 and st["allk"] >= 0.99 and st["max_gap"] <= 5e-7
    assert float((dist_c[rows] - ref_d).abs().max()) < 5e-6
    # determinism: the same launch twice is bit-identical
    idx2, dist2 = ops.knn_topk(qd, pd, 32)
    assert torch.equal(idx, idx2) and torch.equal(dist, dist2)
    # shard invariance at full size: 8 unequal shards merged == single search (device-count invariance)
    cuts = [0, 3000, 7000, 11111, 15000, 19999, 24000, 27000, 30000]
    pi, pdists = [], []
    for a, b in zip(cuts[:-1], cuts[1:]):
        i, d = ops.knn_topk(qd, pd[a:b].contiguous(), 32, idx_offset=a)
        pi.append(i); pdists.append(d)
    mi, md = ops.knn_merge(torch.stack(pdists), torch.stack(pi))
    assert torch.equal(mi, idx) and torch.equal(md, dist)
    # idempotence: pool rows queried against the pool find themselves first, at distance ~0
    self_i, self_d = ops.knn_topk(pd[:256].contiguous(), pd, 4)
    assert bool((self_i[:, 0].cpu() == torch.arange(256)).all()) and float(self_d[:, 0].abs().max()) < 1e-5


def test_match_full_size_against_oracle_pieces(north_star_features):
    """Whole match stage at 1500 x 30000: index stages vs the oracle, Adam by its objective."""
    from knn_svc_amd.matching import match_features
    from oracle import select_ref
    q, p, qf0, pf0, harm = north_star_features
    of, hw, sf0, dbg = match_features(q.to(DEV), qf0.to(DEV), p.to(DEV), pf0.to(DEV), harm.to(DEV), "mix", "post_opt_0.2",
                                      return_debug=True)
    nn32 = dbg["nn32"].cpu()
    sh = select_ref.shift_query_f0(qf0, pf0)
    assert float(((sf0.cpu() - sh).abs() / (sh.abs() + 1e-9)).max()) < 5e-6
    ranked = select_ref.rerank_by_f0(sf0.cpu(), pf0, nn32)       # same shifted f0 -> same keys
    # the frame-sequential re-selection over ALL 1500 frames (round 3 checked the first 300; the oracle's loop takes seconds)
    sel = select_ref.concat_reselect(nn32[:, :4].clone(), q, p, concat_weight=0.2)
    match = float((dbg["idx_wavlm"].cpu() == sel).all(1).float().mean())
    sel2 = select_ref.concat_reselect(ranked[:, :4].clone(), q, p, sf0.cpu(), pf0, concat_weight=0.2)
    match2 = float((dbg["idx_harm"].cpu() == sel2).all(1).float().mean())
    print(f"concat re-selection, all 1500 frames: plain {match:.4f}, pitched {match2:.4f}")
    assert match == 1.0 and match2 == 1.0
    for w, idx, pool, scale in ((dbg["w_wavlm"], dbg["idx_wavlm"], p, 0.1), (dbg["w_harm"], dbg["idx_harm"], harm, 1000.0)):
        w, idx = w.cpu(), idx.cpu()
        assert float((w.sum(1) - 1).abs().max()) < 1e-5 and float(w.min()) >= 0

        def loss(wt):
            g = {s: pool[torch.clamp(idx + s, 0, len(pool) - 1)] for s in (-1, 0, 1)}
            e = {s: (g[s] * wt[..., None]).sum(1) for s in g}
            return float(scale * ((e[-1][1:] - e[0][:-1]) ** 2).mean(-1).mean() + scale * ((e[0][1:] - e[1][:-1]) ** 2).mean(-1).mean())
        l_opt, l_uniform = loss(w), loss(torch.full_like(w, 0.25))
        print(f"smoothness loss: optimised {l_opt:.5f} vs uniform {l_uniform:.5f}")
        assert l_opt < l_uniform
    # weighted sums are what the weights say
    ref = (p[dbg["idx_wavlm"].cpu().reshape(-1)].reshape(1500, 4, 1024) * dbg["w_wavlm"].cpu()[..., None]).sum(1)
    assert float((of.cpu() - ref).abs().max()) < 1e-4


def test_vocoder_30s_deterministic_and_graph_equals_eager():
    from knn_svc_amd.vocoder import Vocoder
    voc = Vocoder(S.seeded_state(S.generator_param_spec(C.HIFIGAN_V1, "mix"), 2), C.HIFIGAN_V1, "mix", DEV)
    g = torch.Generator().manual_seed(0)
    N = 1500
    c = torch.randn(N, 1024, generator=g).to(DEV); harm = (torch.rand(N, 49, generator=g) * 0.02).to(DEV)
    _, f0 = S.synth_clip(N * 320, 5); f0 = torch.from_numpy(f0[:N].copy()).to(DEV)
    y1 = voc.forward(c, f0, harm)            # captures the hipGraph
    y2 = voc.forward(c, f0, harm)            # replay
    voc.use_graphs = False
    y3 = voc.forward(c, f0, harm)            # eager
    assert y1.numel() == N * 320 and bool(torch.isfinite(y1).all()) and float(y1.abs().max()) <= 1.0
    assert torch.equal(y1, y2) and torch.equal(y1, y3)


def test_wavlm_large_chunk_independence_and_ragged_tail():
    """A 30 s chunk encodes to the same 1500 frames alone, in a batch, and in front of a ragged tail;
    tails of <= 320 samples are dropped (ddsp_prematch_dataset.py:277-285).  "The same" is bit for bit as long as the
    launches pick the same GEMM kernel; a batch large enough for the 256x256-tile kernel (16x16x32 MFMAs: 32 k per
    instruction) sums over K in another grouping than the 128x128-tile kernel a lone chunk gets, so across that boundary
    the frames agree to fp32 accumulation noise (with KNNSVC_QUAD=0 every launch takes the small tile: bit-equal again)."""
    from knn_svc_amd.wavlm import WavLMEncoder, chunk_plan
    cfg = C.WAVLM_LARGE
    enc = WavLMEncoder(S.seeded_state(S.wavlm_param_spec(cfg, 2), seed=1), cfg, DEV, n_layers=2)
    w, _ = S.synth_clip(30 * 16000 + 5000, 21)
    wg = torch.from_numpy(w).to(DEV)
    full = enc.full_features(wg)
    assert full.shape == (1500 + enc.n_frames(5000 + 320 - 5000 % 320), 1024)
    alone = enc.full_features(wg[:480000])
    assert alone.shape[0] == 1500 and torch.equal(alone, full[:1500])
    many = enc.encode_many([wg, wg[:480000], wg[:480000 + 300]])
    scale = float(full.abs().max())
    d0, d1 = float((many[0] - full).abs().max()) / scale, float((many[1] - alone).abs().max()) / scale
    print(f"batched vs lone chunk: max |difference| / max |feature| = {d0:.2e}, {d1:.2e}")
    assert d0 < 5e-6 and d1 < 5e-6                          # measured 1.3e-6
    import os
    from knn_svc_amd import ops as _kops
    os.environ["KNNSVC_QUAD"] = "0"
    _kops.reload_knobs()                                     # the dispatcher reads its switches once
    try:
        enc2 = WavLMEncoder(S.seeded_state(S.wavlm_param_spec(cfg, 2), seed=1), cfg, DEV, n_layers=2)
        f2 = enc2.full_features(wg)
        m2 = enc2.encode_many([wg, wg[:480000], wg[:480000 + 300]])
        assert torch.equal(m2[0], f2) and torch.equal(m2[1], f2[:1500])
    finally:
        del os.environ["KNNSVC_QUAD"]
        _kops.reload_knobs()
    assert many[2].shape[0] == 1500                       # 300-sample tail dropped
    assert [l for (_s, l, _p) in chunk_plan(480000 + 300)] == [480000]
    assert [l for (_s, l, _p) in chunk_plan(480000 + 321)] == [480000, 321]


def test_edge_cases():
    from knn_svc_amd import ops
    from knn_svc_amd._lib import KnnSvcError
    from knn_svc_amd.matching import match_features
    # pool smaller than k: the reference's topk(k=32) raises; so do we
    q = torch.randn(5, 64, device=DEV); p = torch.randn(20, 64, device=DEV)
    with pytest.raises(KnnSvcError):
        ops.knn_topk(q, p, 32)
    # single query frame, smallest legal pool, every frame unvoiced except two
    q = S.clustered_features(1, 64, 1, n_centres=3).to(DEV); p = S.clustered_features(32, 64, 2, n_centres=3).to(DEV)
    qf0 = torch.tensor([220.0], device=DEV); pf0 = torch.zeros(32, device=DEV); pf0[3] = 200.0; pf0[9] = 180.0
    harm = torch.rand(32, 49, device=DEV)
    of, hw, sf0 = match_features(q, qf0, p, pf0, harm, "mix", "post_opt_0.2")
    assert of.shape == (1, 64) and hw.shape == (1, 49) and bool(torch.isfinite(of).all()) and bool(torch.isfinite(hw).all())
    assert abs(float(sf0[0]) - 180.0) < 1e-3              # lower median of {180, 200} in the log domain
    # two frames: one adjacent pair for the smoothness loop
    q2 = S.clustered_features(2, 64, 3, n_centres=3).to(DEV)
    of, hw, sf0 = match_features(q2, torch.tensor([0.0, 150.0], device=DEV), p, pf0, harm, "mix", "post_opt_0.2")
    assert float(sf0[0]) == 0.0 and bool(torch.isfinite(of).all())
    # additive synth: all-unvoiced track gives the 1e-7 mask floor, not NaN
    cond = torch.empty(4 * 320, 32, device=DEV)
    exc = ops.additive_synth(torch.zeros(4, device=DEV), torch.rand(4, 49, device=DEV), torch.randn(32, 3, device=DEV),
                             torch.zeros(32, device=DEV), cond, 32, want_exc=True)
    assert bool(torch.isfinite(exc).all()) and float(exc.abs().max()) < 1e-3


def test_knn_multi_chunk_pool_and_mask_across_chunks():
    """Pools above 262 144 rows are searched in balanced chunks whose lists are folded by knnsvc_knn_merge (every buffer
    resource stays below 1 GiB).  300 000 x 1024 (1.2 GB): sampled rows against the oracle, the chunked result equals a
    search over explicit shards, and a self-mask that straddles the chunk boundary behaves like an unchunked one."""
    from knn_svc_amd import ops
    from oracle import knn_ref
    np_rows = 300_000
    base = _smooth(S.clustered_features(30_000, 1024, 12, n_centres=120))
    g = torch.Generator().manual_seed(13)
    p = torch.cat([base * (1.0 + 0.01 * i) + 0.02 * torch.randn(30_000, 1024, generator=g) for i in range(10)], 0)
    q = _smooth(S.clustered_features(400, 1024, 14, n_centres=120))
    pd, qd = p.to(DEV), q.to(DEV)
    assert len(ops.prepare_knn_pool(pd, 32)) == 2                                # the two-chunk route
    idx, dist = ops.knn_topk(qd, pd, 32)
    idx_c, dist_c = idx.cpu(), dist.cpu()
    assert bool((dist_c[:, 1:] >= dist_c[:, :-1]).all()) and int(idx_c.max()) < np_rows and int(idx_c.min()) >= 0
    assert int((idx_c >= 150_000).sum()) > 0 and int((idx_c < 150_000).sum()) > 0  # neighbours come from both chunks
    rows = torch.arange(0, 400, 37)
    ref_i, ref_d = knn_ref.knn_topk(q[rows], p, 32)
    st = knn_ref.topk_agreement(ref_i, idx_c[rows], knn_ref.cosine_dist_f64(q[rows], p), tau=5e-7)
    print("300k-row pool, sampled rows:", st)
    assert st["unexplained"] == 0 and st["top4"] == 1.0 and st["sets"] == 1.0 and float((dist_c[rows] - ref_d).abs().max()) < 5e-6
    # explicit shards at other boundaries give the same merged lists
    cuts = [0, 70_000, 200_000, np_rows]
    parts = [ops.knn_topk(qd, pd[a:b].contiguous(), 32, idx_offset=a) for a, b in zip(cuts[:-1], cuts[1:])]
    mi, md = ops.knn_merge(torch.stack([d for _, d in parts]), torch.stack([i for i, _ in parts]))
    assert torch.equal(mi, idx) and torch.equal(md, dist)
    # a mask across the chunk boundary (rows 149 990 .. 150 010 compete at distance exactly 1)
    lo, hi = 149_990, 150_010
    qm = p[lo:hi].to(DEV).contiguous()                                          # the masked rows query themselves
    mi2, md2 = ops.knn_topk(qm, pd, 32, mask=(lo, hi))
    inside = (mi2 >= lo) & (mi2 < hi)
    assert bool((md2[inside] == 1.0).all())
    assert not bool(((mi2[:, 0] >= lo) & (mi2[:, 0] < hi)).any())                # themselves no longer first
    del pd
    torch.cuda.empty_cache()


def test_prematch_speaker_full_size_properties():
    """per_spk_extract's per-speaker body at BASELINE pool size (20 utterances x 1500 frames): bookkeeping and the
    invariants the reference's consumer relies on (hifigan/ddsp_meldataset.py:473-499), plus sampled rows against the
    oracle's masked search."""
    from knn_svc_amd import prematch
    from oracle import prematch_ref
    n_utt, T = 20, 1500
    feats = _smooth(S.clustered_features(n_utt * T, 1024, 21, n_centres=90))
    g = torch.Generator().manual_seed(22)
    spec = torch.rand(n_utt * T, 200, generator=g) + 0.01
    _, f0 = S.synth_clip(n_utt * T * 320, 23)
    f0 = torch.from_numpy(f0[:n_utt * T].copy())
    harm = torch.rand(n_utt * T, 49, generator=g) * 0.05
    mk = lambda x: {f"u{i:02d}": x[i * T:(i + 1) * T].contiguous().to(DEV) for i in range(n_utt)}
    res = prematch.match_speaker(mk(feats), mk(spec), mk(f0), mk(harm))
    pool_h = res["pool"].cpu()
    assert torch.equal(pool_h, feats.half().float())                       # fp16-rounded pool, bit for bit
    assert [it["slice"] for it in res["items"]] == [(i * T, (i + 1) * T) for i in range(n_utt)]
    for i in (0, 7, 19):
        it = res["items"][i]
        nn, nnf, ar, w = (it[k].cpu() for k in ("nearest_nbrs", "nearest_nbrs_f0_priority", "amp_ratio", "harmonics_best_weight_para"))
        s, e = it["slice"]
        assert nn.shape == (T, 32) and not bool(((nn >= s) & (nn < e)).any())          # own utterance masked out
        assert bool((nnf.sort(1).values == nn.sort(1).values).all())                    # the f0 re-sort is a permutation
        assert bool((ar > 0).all()) and bool(torch.isfinite(ar).all())
        assert float((w.sum(1) - 1).abs().max()) < 1e-5 and bool((w >= 0).all())
        rows = torch.arange(0, T, 211)
        ref = prematch_ref.self_knn(feats[s:e][rows], pool_h, s, e)
        first4 = float((ref[:, :4] == nn[rows][:, :4]).all(1).float().mean())
        sets = float(np.mean([set(a.tolist()) == set(b.tolist()) for a, b in zip(ref, nn[rows])]))
        print(f"prematch speaker, utterance {i}: first-4 rows equal {first4:.3f}, top-32 sets equal {sets:.3f}")
        assert first4 == 1.0 and sets >= 0.99              # measured (r03): 1.000 / 1.000 on all three utterances
        ra = prematch_ref.amp_ratio(spec[s:e][rows], spec, nnf[rows][:, :4])
        assert float(((ra - ar[rows]).abs() / ra).max()) < 2e-6
