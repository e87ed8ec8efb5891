"""GPU, world_size 2 on ONE card: the multi-rank paths (BASELINE cfg 4 = dataset mode with the pool sharded,
cfg 5 = one pool sharded under many queries) with the REAL HIP kernels on every rank.  RCCL refuses two ranks on one device, so
the group is gloo and knn_svc_amd.dist stages the collectives through host memory (dist._host_staged) — the kernels, the
shard arithmetic, the ownership of items and the merge are the production code, only the transport differs.

The ranks are child processes; this process must not have touched the GPU before it starts them (a GPU-initialised process
must not be the origin of an exec on the GPU boxes), so the module sorts first among the GPU test files and skips itself
when something initialised the device earlier in the same interpreter."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _clean_parent():
    if torch.cuda.is_initialized():
        pytest.skip("ranks are spawned from a process that has not initialised the GPU: run this module first / on its own")


def _make_dataset(root):
    sys.path.insert(0, ROOT)
    from knn_svc_amd import audio_io, synthetic as S
    for s in range(3):
        d = os.path.join(root, f"spk{s}")
        os.makedirs(d, exist_ok=True)
        for u in range(3):
            n = 16000 + 2400 * u + 777 * s + 13
            w, f0 = S.synth_clip(n, 70 + 10 * s + u)
            audio_io.write_wav_pcm16(os.path.join(d, f"u{u}.wav"), w, 16000)
            np.save(os.path.join(d, f"u{u}_f0.npy"), f0.astype(np.float32))


def _bulk_worker(rank, ws, port, root, out_dir, shard, res):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    os.environ["KNNSVC_POOL_CACHE_GB"] = "0"
    if shard:
        os.environ["KNNSVC_POOL_SHARD"] = "1"
    else:
        os.environ.pop("KNNSVC_POOL_SHARD", None)
    import torch.distributed as dist
    if ws > 1:
        dist.init_process_group("gloo", rank=rank, world_size=ws)
    from knn_svc_amd import config as C, synthetic as S
    from knn_svc_amd.matcher import KNeighborsVC
    from knn_svc_amd.vocoder import Vocoder
    from knn_svc_amd.wavlm import WavLMEncoder
    dev = "cuda:0"
    cfg, h = C.WAVLM_TINY, C.HIFIGAN_TINY
    enc = WavLMEncoder(S.seeded_state(S.wavlm_param_spec(cfg), seed=11), cfg, dev, n_layers=2)
    vc = KNeighborsVC(enc, Vocoder(S.seeded_state(S.generator_param_spec(h, "mix"), 63), h, "mix", dev), h, dev)
    written = vc.bulk_match(root, root, out_dir, ckpt_type="mix", post_opt="post_opt_0.2", duration_limit=None)
    torch.cuda.synchronize()
    res[rank] = list(written)
    if ws > 1:
        dist.destroy_process_group()


def _launch(target, ws, args):
    ctx = mp.get_context("spawn")
    res = ctx.Manager().dict()
    procs = [ctx.Process(target=target, args=(r, ws) + args + (res,)) for r in range(ws)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(420)
        assert p.exitcode == 0, p.exitcode
    return [res[r] for r in range(ws)]


def _read_all(files):
    sys.path.insert(0, ROOT)
    from knn_svc_amd import audio_io
    return {os.path.relpath(f, os.path.dirname(os.path.dirname(os.path.dirname(f)))): audio_io.read_wav(f)[0] for f in files}


def test_bulk_match_two_ranks_on_one_gpu_equal_one_rank(tmp_path):
    """BASELINE cfg 4 at world_size 2 with the HIP kernels: speaker pairs dealt over the ranks (no collective) and
    KNNSVC_POOL_SHARD=1 (pool encoded in shares, per-shard search, all-to-all of the lists, utterances dealt out) both write
    the files one rank writes — same names, same samples."""
    _clean_parent()
    root = str(tmp_path / "data")
    _make_dataset(root)
    port = 36200 + (os.getpid() % 1500)
    (single,) = _launch(_bulk_worker, 1, (port, root, str(tmp_path / "single"), False))
    assert len(single) == 3 * 2 * 3
    ref = _read_all(single)
    for shard in (False, True):
        out_dir = str(tmp_path / ("shard" if shard else "pairs"))
        w0, w1 = _launch(_bulk_worker, 2, (port + 1 + int(shard), root, out_dir, shard))
        assert w0 == w1 and len(w0) == len(single) == len(set(w0))
        got = _read_all(w0)
        assert sorted(got) == sorted(ref)
        worst = 0.0
        for k in ref:
            assert got[k].shape == ref[k].shape, k
            worst = max(worst, float(np.abs(got[k] - ref[k]).max()))
        print(f"pool_shard={shard}: {len(got)} files, max |difference| to the one-rank run {worst:.3g}")
        assert worst <= 1e-5, (shard, worst)


def _knn_worker(rank, ws, port, res):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    from knn_svc_amd import dist as kd, ops, synthetic as S
    dev = "cuda:0"
    pool = S.clustered_features(9000, 256, 9, n_centres=40).to(dev)
    cut = 5437                                                                  # uneven shards
    mine = (pool[:cut] if rank == 0 else pool[cut:]).contiguous()
    ok = True
    # (a) every rank brings its own queries, the pool is sharded: result = unsharded search
    nq = 300
    q = S.clustered_features(nq, 256, 100 + rank, n_centres=40).to(dev)
    idx, d = kd.sharded_knn(q, mine, 32)
    fi, fd, _f = ops.knn_topk(q, pool, 32, check_nan=False, return_flag=True)
    ok = ok and bool(torch.equal(d, fd)) and bool(torch.equal(idx, fi))
    # (b) replicated queries with owners (dataset mode x pool shard): a rank receives only its rows
    q_all = S.clustered_features(257, 256, 321, n_centres=40).to(dev)
    rows = [100, 157]
    idx, d = kd.sharded_knn_owned(q_all, rows, mine, 32)
    lo = sum(rows[:rank])
    fi, fd, _f = ops.knn_topk(q_all[lo:lo + rows[rank]], pool, 32, check_nan=False, return_flag=True)
    ok = ok and idx.shape == (rows[rank], 32) and bool(torch.equal(d, fd)) and bool(torch.equal(idx, fi))
    # (c) variable-size row gather and the deferred NaN check (every rank raises together)
    g = kd.all_gather_rows_var(mine[:, :8].contiguous())
    ok = ok and bool(torch.equal(g, pool[:, :8]))
    for dst in (0, 1):              # rows to one rank only (strong scaling: the owner of a conversion's back half)
        g1 = kd.gather_rows_var(mine[:, :8].contiguous(), [cut, 9000 - cut], dst)
        ok = ok and ((g1 is None) if rank != dst else (g1.is_cuda and bool(torch.equal(g1, pool[:, :8]))))
    kd.raise_if_any_nan()
    qn = q.clone(); qn[3, 5] = float("nan")
    raised = False
    if rank == 1:
        kd.sharded_knn(qn, mine, 32)
    else:
        kd.sharded_knn(q, mine, 32)
    try:
        kd.raise_if_any_nan()
    except ops.KnnSvcError:
        raised = True
    res[rank] = bool(ok and raised)
    dist.destroy_process_group()


def test_sharded_knn_two_ranks_on_one_gpu_equals_unsharded_search():
    """BASELINE cfg 5's search at world_size 2: per-shard fused distance / top-k (HIP), lists exchanged, merged with the
    single-GPU ordering — indices and distances bit-equal to the unsharded search, uneven shards and row ownership; a NaN in
    one rank's queries raises on every rank."""
    _clean_parent()
    port = 38200 + (os.getpid() % 1500)
    r0, r1 = _launch(_knn_worker, 2, (port,))
    assert r0 and r1


def _cfg4_worker(rank, ws, port, res):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    from knn_svc_amd import dist as kd, ops, synthetic as S
    dev = "cuda:0"
    sizes = [11250 + d for d in (300, -300, 77, -77)]                            # 45 000 rows in four uneven shards
    pool = S.clustered_features(45000, 1024, 9, n_centres=80)
    o = sum(sizes[:rank])
    mine = pool[o:o + sizes[rank]].contiguous().to(dev)
    q = S.clustered_features(400, 1024, 321, n_centres=80).to(dev)              # one utterance, the same on every rank
    fused0 = ops.KNN_ROUTE_COUNTS["fused"]
    idx, d = kd.sharded_knn(q, mine, 32, replicated=True, counts=sizes)
    took_fused = ops.KNN_ROUTE_COUNTS["fused"] > fused0
    kd.raise_if_any_nan()
    ok = True
    if rank == 0:                                                                # the unsharded search, once
        fi, fd = ops.knn_topk(q, pool.to(dev), 32)
        ok = bool(torch.equal(idx, fi)) and bool(torch.equal(d, fd))
    whole = kd.all_gather_rows_var(mine[:, :16].contiguous(), sizes)
    ok = ok and bool(torch.equal(whole.cpu(), pool[:, :16]))
    res[rank] = bool(ok and took_fused)
    dist.destroy_process_group()


def test_cfg4_size_pool_in_four_uneven_shards_equals_unsharded_search():
    """BASELINE cfg 4's search at its pool size with the HIP kernels: a 45 000-row speaker pool (1024-d) in four uneven shards, one
    utterance's 400 query frames replicated on four ranks sharing the card (gloo, host-staged collectives), every shard searched on
    the fused route and re-scored exactly, the [400, 32] lists all-gathered and merged: indices and distances BIT-EQUAL to the
    unsharded search of the whole pool (the exact re-score makes a pair's distance independent of what else its shard holds)."""
    _clean_parent()
    port = 39900 + (os.getpid() % 1500)
    res = _launch(_cfg4_worker, 4, (port,))
    assert all(res), res


def test_bench_self_launches_its_ranks_rehearsal():
    """`python bench.py --gpus 2` with no RANK in the environment — the shape of the driver's own command — starts its two ranks
    itself (child processes of torch.distributed.run, before anything touches the GPU) and relays rank 0's JSON line.  Rehearsal
    mode: both ranks on cuda:0 under gloo (KNNSVC_BENCH_REHEARSE=1), so the N > 1 code path runs end to end on a one-GPU box."""
    import json
    import subprocess
    _clean_parent()
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(KNNSVC_BENCH_REHEARSE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for mode in ("weak", "strong"):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--scaling", mode,
                            "--no-cpu-baseline", "--no-other-configs"], env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, (mode, r.stderr[-3000:])
        assert "launching 2 ranks" in r.stderr
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1, r.stdout[-2000:]
        line = json.loads(lines[0])
        assert line["n_gpus"] == 2 and line["rehearsal"] is True and line["scaling"] == mode and line["value"] > 0, line
