"""CPU, world_size 2 over gloo: dataset mode (``KNeighborsVC.bulk_match``) under both ways of sharing a run between
ranks (BASELINE cfg 4; reference loop ddsp_matcher.py:1073-1133):

  * default — speaker pairs dealt round-robin, every rank alone on its pairs, no collective;
  * KNNSVC_POOL_SHARD=1 — every rank walks every pair, the target pool and the source files are encoded in shares,
    the kNN is searched per shard (lists exchanged with one all-to-all) and the utterances are dealt over the ranks.

Both must write exactly the files (same names, same samples) a single process writes.  The HIP kernels are replaced by
CPU stand-ins injected at the module seams (a projection "encoder", the oracle's kNN / re-rank / re-selection, a toy
"vocoder"): what is under test is the host logic — file shares, collectives, item ownership, output naming — which is
the same code that runs over RCCL on the GPUs."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
E = 64


class FakeEncoder:
    """Duck-typed WavLMEncoder: frames of 400 samples at hop 320 projected to E dims (deterministic, CPU)."""
    uid, n_layers, E = 1, 6, E
    cfg = {"encoder_layers": 24}
    device = torch.device("cpu")
    encoded = 0                      # frames this process encoded (to show that the shares are shares)

    def __init__(self):
        g = torch.Generator().manual_seed(5)
        self.proj = torch.randn(400, E, generator=g)

    def n_frames(self, n):
        return (n - 400) // 320 + 1

    def set_layer_mix(self, weights):
        assert weights is None          # the live path's one-hot on layer 6

    def weights_fingerprint(self):
        return "fake"

    def encode_many(self, wavs, max_batch=8, pow2_batches=False):
        from knn_svc_amd.wavlm import chunk_plan
        out = []
        for w in wavs:
            parts = []
            for (s, l, p) in chunk_plan(w.numel()):
                x = torch.nn.functional.pad(w[s:s + l], (0, p))
                parts.append(x.unfold(0, 400, 320) @ self.proj)
            out.append(torch.cat(parts, 0))
            FakeEncoder.encoded += out[-1].shape[0]
        return out


class FakeVocoder:
    def forward(self, c, f0, harm=None):
        y = 0.5 * torch.tanh(c[:, :8].mean(1) + 1e-3 * f0 + (harm.sum(1) if harm is not None else 0.0))
        return y.repeat_interleave(320)


def _cpu_local_topk(q, pool, k, offset):
    from oracle import knn_ref
    idx, d = knn_ref.knn_topk(q, pool, k)
    return idx + offset, d


def _cpu_merge(part_dist, part_idx):
    parts, nq, k = part_dist.shape
    d = part_dist.permute(1, 0, 2).reshape(nq, parts * k)
    i = part_idx.permute(1, 0, 2).reshape(nq, parts * k)
    order = torch.argsort(i, dim=1, stable=True)
    d, i = d.gather(1, order), i.gather(1, order)
    order = torch.argsort(d, dim=1, stable=True)[:, :k]
    return i.gather(1, order), d.gather(1, order)


def _cpu_match_features(query_seq, query_f0, matching_list, matching_f0, harmonics_list, ckpt_type, post_opt,
                        return_debug=False, nn32=None, nan_flags=None, pool_prep=None):
    from oracle import select_ref
    if nn32 is None:
        nn32 = _cpu_local_topk(query_seq, matching_list, 32, 0)[0]
    sh = select_ref.shift_query_f0(query_f0, matching_f0)
    idx = nn32[:, :4]
    idx2 = select_ref.rerank_by_f0(sh, matching_f0, nn32)[:, :4]
    of = matching_list[idx.reshape(-1)].reshape(-1, 4, matching_list.shape[1]).mean(1)
    hw = harmonics_list[idx2.reshape(-1)].reshape(-1, 4, harmonics_list.shape[1]).mean(1)
    return of, hw, sh


def _cpu_side_features(wav, f0_host, T):
    f0 = torch.from_numpy(np.ascontiguousarray(f0_host[:T]))
    fr = wav[:T * 320].reshape(T, 320)
    harm = torch.stack([fr.abs().mean(1) * (k + 1) for k in range(49)], 1)
    return f0, harm, fr[:, :200].abs().contiguous()


def _inject():
    from knn_svc_amd import dist as kd, matching
    matching.side_features = _cpu_side_features
    matching.match_features = _cpu_match_features
    matching.prepare_pool = lambda P, split=True: None
    matching.batched_knn = lambda q_all, P, prep, max_blocks=0: (_cpu_local_topk(q_all, P, 32, 0)[0], None)
    kd._hip_local_topk = _cpu_local_topk
    kd._hip_merge = _cpu_merge
    matching._POOL_CACHE = None


def make_dataset(root):
    """3 speakers x 4 utterances (2.5-4 s, one longer than 30 s is not needed here) with f0 caches."""
    from knn_svc_amd import audio_io, synthetic as S
    for s in range(3):
        d = os.path.join(root, f"spk{s}")
        os.makedirs(d, exist_ok=True)
        for u in range(4):
            n = 16000 * 2 + 4000 * u + 777 * s + 13
            w, f0 = S.synth_clip(n, 50 + 10 * s + u)
            audio_io.write_wav_pcm16(os.path.join(d, f"u{u}.wav"), w, 16000)
            np.save(os.path.join(d, f"u{u}_f0.npy"), f0.astype(np.float32))


def _run_bulk(root, out_dir):
    from knn_svc_amd.matcher import KNeighborsVC
    vc = KNeighborsVC(FakeEncoder(), FakeVocoder(), {"sampling_rate": 16000}, device="cpu")
    return vc.bulk_match(root, root, out_dir, ckpt_type="mix", post_opt="no_post_opt", duration_limit=None)


def _worker(rank, ws, port, root, out_dir, shard, res):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    os.environ["KNNSVC_POOL_CACHE_GB"] = "0"          # every pair encodes its files: the encoded-frame count is then comparable
    if shard:
        os.environ["KNNSVC_POOL_SHARD"] = "1"
    else:
        os.environ.pop("KNNSVC_POOL_SHARD", None)
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    _inject()
    written = _run_bulk(root, out_dir)
    res[rank] = (written, FakeEncoder.encoded)
    dist.destroy_process_group()


def _launch(root, out_dir, shard, port):
    ctx = mp.get_context("spawn")
    res = ctx.Manager().dict()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, root, out_dir, shard, res)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    return res[0], res[1]


def _read_all(files):
    from knn_svc_amd import audio_io
    return {os.path.relpath(f, os.path.dirname(os.path.dirname(os.path.dirname(f)))): audio_io.read_wav(f)[0] for f in files}


def test_bulk_match_pool_shard_and_pair_share_equal_single_process(tmp_path, monkeypatch):
    sys.path.insert(0, ROOT)
    root = str(tmp_path / "data")
    make_dataset(root)
    # single process, no process group: the reference's order, everything on one rank
    monkeypatch.setenv("KNNSVC_POOL_CACHE_GB", "0")
    monkeypatch.delenv("KNNSVC_POOL_SHARD", raising=False)
    from knn_svc_amd import dist as kd, matching
    saved = (matching.side_features, matching.match_features, matching.prepare_pool, kd._hip_local_topk, kd._hip_merge)
    saved_bk = matching.batched_knn
    try:
        _inject()
        FakeEncoder.encoded = 0
        single = _run_bulk(root, str(tmp_path / "single"))
        enc_single = FakeEncoder.encoded
    finally:
        matching.side_features, matching.match_features, matching.prepare_pool, kd._hip_local_topk, kd._hip_merge = saved
        matching.batched_knn = saved_bk
        matching._POOL_CACHE = None
    assert len(single) == 3 * 2 * 4                                       # 6 ordered pairs x 4 utterances
    ref = _read_all(single)
    port = 35500 + (os.getpid() % 2000)
    for shard in (False, True):
        out_dir = str(tmp_path / ("shard" if shard else "pairs"))
        (w0, e0), (w1, e1) = _launch(root, out_dir, shard, port + int(shard))
        assert w0 == w1 and len(w0) == len(single) == len(set(w0))       # gathered list: every file once, same on both ranks
        got = _read_all(w0)
        assert sorted(got) == sorted(ref)
        for k in ref:
            assert got[k].shape == ref[k].shape and np.array_equal(got[k], ref[k]), k
        # the work really is shared: neither rank encoded everything, together they encoded what one process encodes
        assert 0 < e0 < enc_single and 0 < e1 < enc_single and e0 + e1 == enc_single, (e0, e1, enc_single)


def _a2a_worker(rank, ws, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    from knn_svc_amd import dist as kd, synthetic as S
    pool = S.clustered_features(700, 64, 9, n_centres=12)
    cut = 437
    mine = (pool[:cut] if rank == 0 else pool[cut:]).contiguous()
    # replicated queries with owners: rank 0 owns the first 17 rows, rank 1 the next 30 — uneven on purpose
    q_all = S.clustered_features(47, 64, 321, n_centres=12)
    rows = [17, 30]
    idx, d = kd.sharded_knn_owned(q_all, rows, mine, 8, _cpu_local_topk, _cpu_merge)
    lo = sum(rows[:rank])
    fi, fd = _cpu_local_topk(q_all[lo:lo + rows[rank]], pool, 8, 0)
    ok = idx.shape == (rows[rank], 8) and bool(torch.equal(d, fd)) and bool((idx == fi).float().mean() > 0.99)
    # an owner with no rows at all
    idx0, d0 = kd.sharded_knn_owned(q_all[:9], [9, 0], mine, 8, _cpu_local_topk, _cpu_merge)
    ok = ok and idx0.shape[0] == (9 if rank == 0 else 0)
    t = torch.arange(10, dtype=torch.float32).reshape(5, 2) + 100 * rank
    got = kd.all_to_all_rows(t, [2, 3], [2, 2] if rank == 0 else [3, 3])
    want = torch.cat([t0[:2] if rank == 0 else t0[2:] for t0 in (torch.arange(10.).reshape(5, 2), torch.arange(10.).reshape(5, 2) + 100)])
    out[rank] = ok and bool(torch.equal(got, want))
    dist.destroy_process_group()


def test_sharded_knn_owned_all_to_all_gloo_world2():
    """The list exchange of the pool-sharded search is an all-to-all: a rank receives only the lists of the query rows it
    owns (uneven ownership and uneven pool shards), and the merged result equals the unsharded search."""
    ctx = mp.get_context("spawn")
    out = ctx.Manager().dict()
    port = 37500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_a2a_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert out[0] and out[1]
