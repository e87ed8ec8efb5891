"""CPU, world_size 2 over gloo: pool-row sharding + all-gather merge gives the single-process result.
The local top-k / merge are the oracle here (no GPU in this container); on the GPU box the same
``sharded_knn`` runs with the HIP kernels (tests/test_gpu_kernels.py::test_knn_shard_merge_equals_single)."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _cpu_local_topk(q, pool, k, offset):
    from oracle import knn_ref
    idx, d = knn_ref.knn_topk(q, pool, k)
    return idx + offset, d


def _cpu_merge(part_dist, part_idx):
    parts, nq, k = part_dist.shape
    d = part_dist.permute(1, 0, 2).reshape(nq, parts * k)
    i = part_idx.permute(1, 0, 2).reshape(nq, parts * k)
    # (distance, lower index) lexicographic order == the HIP kernels' packed-key order
    order = torch.argsort(i, dim=1, stable=True)
    d, i = d.gather(1, order), i.gather(1, order)
    order = torch.argsort(d, dim=1, stable=True)[:, :k]
    return i.gather(1, order), d.gather(1, order)


def _worker(rank, ws, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    from knn_svc_amd import dist as kd, synthetic as S
    pool = S.clustered_features(600, 64, 7, n_centres=12)
    npr = 300
    q = S.clustered_features(40, 64, 100 + rank, n_centres=12)
    idx, d = kd.sharded_knn(q, pool[rank * npr:(rank + 1) * npr].contiguous(), 8, _cpu_local_topk, _cpu_merge)
    ref_i, ref_d = _cpu_local_topk(q, pool, 8, 0)
    ok = bool(torch.equal(d, ref_d)) and bool((idx == ref_i).float().mean() > 0.99)
    gathered = kd.all_gather_rows(torch.full((2, 3), float(rank)))
    ok = ok and gathered.shape == (4, 3) and float(gathered[2, 0]) == 1.0
    out[rank] = ok
    dist.destroy_process_group()


def test_sharded_knn_gloo_world2():
    ctx = mp.get_context("spawn")
    mgr = ctx.Manager()
    out = mgr.dict()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert out[0] and out[1]


def test_single_process_passthrough():
    sys.path.insert(0, ROOT)
    from knn_svc_amd import dist as kd, synthetic as S
    q = S.clustered_features(10, 64, 1, n_centres=4); p = S.clustered_features(100, 64, 2, n_centres=4)
    idx, d = kd.sharded_knn(q, p, 8, _cpu_local_topk, _cpu_merge)
    ref_i, ref_d = _cpu_local_topk(q, p, 8, 0)
    assert torch.equal(idx, ref_i) and torch.equal(d, ref_d)


def _share_worker(rank, ws, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    from knn_svc_amd import dist as kd
    pairs = [(f"s{i}", f"t{j}") for i in range(3) for j in range(3) if i != j]      # bulk_match's pair list
    mine = kd.my_share(pairs)
    out[rank] = (mine, kd.gather_paths([f"/o/{a}/{b}.wav" for a, b in mine]))
    dist.destroy_process_group()


def test_work_sharing_gloo_world2():
    """Dataset mode / prematch shard their independent units (speaker pairs, speaker folders) round-robin over ranks with
    no data-path collective: the shares partition the list, and every rank ends up with the full list of written files."""
    ctx = mp.get_context("spawn")
    out = ctx.Manager().dict()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_share_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    (m0, g0), (m1, g1) = out[0], out[1]
    pairs = [(f"s{i}", f"t{j}") for i in range(3) for j in range(3) if i != j]
    assert sorted(m0 + m1) == sorted(pairs) and not set(m0) & set(m1) and abs(len(m0) - len(m1)) <= 1
    assert m0 == pairs[0::2] and m1 == pairs[1::2]
    assert g0 == g1 and len(g0) == len(pairs) and g0[:len(m0)] == [f"/o/{a}/{b}.wav" for a, b in m0]


def test_work_sharing_single_process():
    sys.path.insert(0, ROOT)
    from knn_svc_amd import dist as kd
    assert kd.my_share([1, 2, 3]) == [1, 2, 3] and kd.gather_paths(["a"]) == ["a"]
    t = torch.arange(6.).reshape(3, 2)
    assert kd.gather_rows_var(t, [3], 0) is t


def _uneven_worker(rank, ws, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    from knn_svc_amd import dist as kd, synthetic as S
    pool = S.clustered_features(700, 64, 9, n_centres=12)
    cut = 437                                               # shards of 437 and 263 rows
    mine = pool[:cut] if rank == 0 else pool[cut:]
    q = S.clustered_features(30, 64, 200 + rank, n_centres=12)
    idx, d = kd.sharded_knn(q, mine.contiguous(), 8, _cpu_local_topk, _cpu_merge)
    ref_i, ref_d = _cpu_local_topk(q, pool, 8, 0)
    whole = kd.all_gather_rows_var(mine.contiguous())
    ok = bool(torch.equal(d, ref_d)) and bool((idx == ref_i).float().mean() > 0.99) and bool(torch.equal(whole, pool)) \
        and kd.shard_rows(mine.shape[0], mine.device) == [437, 263]
    # one conversion against a sharded pool: the SAME queries on every rank, searched once per shard
    qr = S.clustered_features(25, 64, 777, n_centres=12)
    ri, rd = kd.sharded_knn(qr, mine.contiguous(), 8, _cpu_local_topk, _cpu_merge, replicated=True)
    fi, fd = _cpu_local_topk(qr, pool, 8, 0)
    ok = ok and bool(torch.equal(rd, fd)) and bool((ri == fi).float().mean() > 0.99)
    # rows to ONE rank only (the owner of a conversion's back half, bench.py --scaling strong), both ways round
    for dst in (0, 1):
        g = kd.gather_rows_var(mine.contiguous(), [437, 263], dst)
        ok = ok and ((g is None) if rank != dst else bool(torch.equal(g, pool)))
    # contiguous file shares reproduce the single-process order
    lo, hi = kd.contiguous_share(11)
    ok = ok and (lo, hi) == ((0, 6) if rank == 0 else (6, 11))
    out[rank] = ok
    dist.destroy_process_group()


def test_sharded_knn_uneven_shards_gloo_world2():
    """BASELINE cfg 4: a real speaker pool does not divide evenly over the ranks — global rows follow the cumulative
    shard sizes and the row all-gather pads to the largest shard."""
    ctx = mp.get_context("spawn")
    out = ctx.Manager().dict()
    port = 33500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_uneven_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert out[0] and out[1]


def _w8_worker(rank, ws, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    from knn_svc_amd import dist as kd, synthetic as S
    ok = True
    # ---- bench.py --scaling strong at N = 8: 20 pool clips in contiguous, balanced shares (3, 3, 3, 3, 2, 2, 2, 2)
    lo, hi = kd.contiguous_share(20)
    clips = [3, 3, 3, 3, 2, 2, 2, 2]
    ok = ok and hi - lo == clips[rank] and lo == sum(clips[:rank])
    rows_per_clip = 150                                       # (1500 in the bench; the host logic does not depend on it)
    counts = [c * rows_per_clip for c in clips]
    pool = S.clustered_features(20 * rows_per_clip, 64, 9, n_centres=24)
    mine = pool[lo * rows_per_clip:hi * rows_per_clip].contiguous()
    ok = ok and kd.shard_rows(mine.shape[0], mine.device) == counts
    q = S.clustered_features(40, 64, 777, n_centres=24)       # the replicated source
    ri, rd = kd.sharded_knn(q, mine, 8, _cpu_local_topk, _cpu_merge, replicated=True, counts=counts)
    fi, fd = _cpu_local_topk(q, pool, 8, 0)
    ok = ok and bool(torch.equal(rd, fd)) and bool((ri == fi).float().mean() > 0.99)
    for conv in range(10):                                     # the owner of a conversion's back half rotates over the ranks
        g = kd.gather_rows_var(mine, counts, conv % ws)
        ok = ok and ((g is None) if rank != conv % ws else bool(torch.equal(g, pool)))
    ok = ok and bool(torch.equal(kd.all_gather_rows_var(mine, counts), pool))
    # ---- BASELINE cfg 4's shape: a 45 000-row speaker pool in 8 uneven shards, every rank's own queries (weak form: all-to-all merge)
    sizes = [5625 + d for d in (40, -40, 13, -13, 7, -7, 100, -100)]
    assert sum(sizes) == 45000
    big = S.clustered_features(45000, 32, 5, n_centres=40)
    o = sum(sizes[:rank])
    shard = big[o:o + sizes[rank]].contiguous()
    qq = S.clustered_features(12, 32, 300 + rank, n_centres=40)
    idx, d = kd.sharded_knn(qq, shard, 32, _cpu_local_topk, _cpu_merge)
    ref_i, ref_d = _cpu_local_topk(qq, big, 32, 0)
    ok = ok and bool(torch.equal(d, ref_d)) and bool((idx == ref_i).float().mean() > 0.99)
    whole = kd.all_gather_rows_var(shard)                     # (counts read back: one tiny all-gather)
    ok = ok and whole.shape == big.shape and bool(torch.equal(whole, big))
    # independent units dealt round-robin (dataset mode's speaker pairs)
    ok = ok and kd.my_share(list(range(19))) == list(range(rank, 19, 8))
    out[rank] = ok
    dist.destroy_process_group()


def test_world_size_8_uneven_strong_split_and_cfg4_pool_shape():
    """VERDICT r4 #6a (first-run insurance for the 8-GPU node): the host logic of both scaling modes at world size 8 over
    gloo — the uneven 20-clip strong split (3, 3, 3, 3, 2, 2, 2, 2) with replicated queries, the rotating owner's point-to-point
    row gather, the padded uneven all-gather (the ONE form, also on RCCL), and cfg 4's 45 000-row pool in eight uneven shards with
    per-rank queries and the all-to-all merge.  Every result equals the single-process one."""
    ctx = mp.get_context("spawn")
    out = ctx.Manager().dict()
    port = 35500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_w8_worker, args=(r, 8, port, out)) for r in range(8)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    assert all(out[r] for r in range(8)), dict(out)
