"""Multi-GPU layer: pool-row sharding with an RCCL all-gather merge.

The reference is single-device.  Here every rank (one process per GPU, ``torch.distributed``
backend "nccl" == RCCL over xGMI) owns a contiguous range of pool rows:

  1. queries of all ranks are all-gathered (Nq x 4 KB each — small);
  2. each rank runs the fused distance/top-k kernel against ITS shard only and reports
     (distance, global row) lists — no pool bytes cross the fabric for the search;
  3. one all-to-all of the [Nq_total, 32] lists (8 B per entry: every rank receives only the lists
     of ITS queries, one block per shard), then an 8-way merge with the same (distance, lower
     index) ordering as the single-GPU kernel, so results do not depend on the number of devices
     (replicated queries whose result every rank needs use an all-gather instead);
  4. the rows the later stages read (selected neighbours and their +/-1 neighbours) are served
     from an all-gathered copy of the pool features / f0 / harmonics (<= ~200 MB per speaker).

xGMI is point to point (7 links per GPU), so the collectives are plain all-gathers / all-to-alls
whose per-peer messages travel on their own link; there is no ring or tree to tune.
The local top-k and the merge are injectable so that the sharding logic is testable on CPU
with gloo (tests/test_dist_cpu.py) — the product path always uses the HIP kernels.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def _host_staged(t: torch.Tensor) -> bool:
    """A gloo group given DEVICE tensors: the collective travels through host memory.  This is the rehearsal mode — several
    ranks sharing one GPU (RCCL refuses two ranks on one device) run the real HIP kernels and the real multi-rank host
    logic, only the transport differs (tests/test_gpu_dist2.py).  Production groups are "nccl" (= RCCL over xGMI)."""
    return t.is_cuda and dist.get_backend() == "gloo"


def _all_gather_into(out: torch.Tensor, t: torch.Tensor, async_op=False):
    if _host_staged(t):
        o = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(o, t.contiguous().cpu())
        out.copy_(o)
        return None
    return dist.all_gather_into_tensor(out, t.contiguous(), async_op=async_op)


def all_gather_rows(t: torch.Tensor) -> torch.Tensor:
    """Concatenate equal-shaped [n, ...] tensors of every rank along dim 0 (rank order)."""
    _r, ws = world()
    if not (dist.is_available() and dist.is_initialized()):
        return t                      # single process without a process group
    out = torch.empty((ws * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    _all_gather_into(out, t)
    return out


def all_gather_rows_async(t: torch.Tensor):
    """all_gather_rows started now and waited for later: returns (out, wait) where wait() makes the CURRENT stream wait
    for the collective (no host block).  Kernels enqueued between the two calls run while the bytes travel — the pool
    all-gather of a step flies under the rank's local kNN search."""
    _r, ws = world()
    if not (dist.is_available() and dist.is_initialized()):
        return t, (lambda: None)
    out = torch.empty((ws * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    work = _all_gather_into(out, t, async_op=True)
    return out, (work.wait if work is not None else (lambda: None))


def shard_rows(n_local: int, device) -> list:
    """Row counts of every rank's shard (one tiny all-gather); [n_local] without a process group."""
    if not (dist.is_available() and dist.is_initialized()):
        return [int(n_local)]
    _r, ws = world()
    mine = torch.tensor([int(n_local)], dtype=torch.int64, device=device)
    out = torch.empty(ws, dtype=torch.int64, device=device)
    _all_gather_into(out, mine)
    return [int(v) for v in out.tolist()]


def all_gather_rows_var(t: torch.Tensor, counts=None) -> torch.Tensor:
    """all_gather_rows for shards of DIFFERENT row counts (a real speaker pool rarely divides evenly): every rank pads to the
    largest shard, ONE all_gather_into_tensor, the padding rows are dropped.  The same code on RCCL and on gloo — until round 4
    the RCCL branch used ``dist.all_gather`` into uneven views, a form that had never executed anywhere (no multi-GPU node so
    far); the padding costs at most one shard-size difference per rank (one 30 s clip: 6 MB) and the form below is the one
    the gloo tests (world sizes 2 and 8) and the one-card rehearsals exercise.  ``counts`` = shard_rows(...) if already known."""
    if not (dist.is_available() and dist.is_initialized()):
        return t
    counts = counts if counts is not None else shard_rows(t.shape[0], t.device)
    if len(set(counts)) == 1:
        return all_gather_rows(t)
    mx = max(counts)
    pad = torch.zeros((mx,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    pad[:t.shape[0]] = t
    g = all_gather_rows(pad).view((len(counts), mx) + tuple(t.shape[1:]))
    return torch.cat([g[r, :c] for r, c in enumerate(counts)], 0)


def gather_rows_var(t: torch.Tensor, counts, dst: int):
    """The rows of every rank's shard (``counts[r]`` rows on rank r), in rank order, on rank ``dst`` ONLY -> the whole tensor
    there, None elsewhere.  Point-to-point (one grouped isend / irecv batch): a pool that only its owner matches against is not
    replicated (bench.py --scaling strong: the owner of a conversion rotates over the ranks).  gloo with device tensors
    (one-card rehearsal) stages through host memory like the collectives above."""
    rank, ws = world()
    if ws == 1:
        return t
    counts = [int(c) for c in counts]
    assert t.shape[0] == counts[rank], (t.shape, counts, rank)
    staged = _host_staged(t)
    src = t.contiguous().cpu() if staged else t.contiguous()
    if rank != dst:
        if counts[rank]:
            for r in dist.batch_isend_irecv([dist.P2POp(dist.isend, src, dst)]):
                r.wait()
        return None
    out = torch.empty((sum(counts),) + tuple(t.shape[1:]), dtype=t.dtype, device=src.device)
    views = out.split(counts, 0)
    views[rank].copy_(src)
    ops_ = [dist.P2POp(dist.irecv, views[r], r) for r in range(ws) if r != rank and counts[r]]
    if ops_:
        for r in dist.batch_isend_irecv(ops_):
            r.wait()
    return out.to(t.device) if staged else out


_NAN_FLAGS = []      # device flags of the per-shard searches since the last raise_if_any_nan()


def _hip_local_topk(q, pool, k, offset):
    from . import ops
    idx, d, flag = ops.knn_topk(q, pool, k, idx_offset=offset, check_nan=False, return_flag=True)
    _NAN_FLAGS.append(flag)          # read later, once (a read here would synchronise every step of a stream pipeline)
    return idx, d


def raise_if_any_nan():
    """The reference exits when any distance is NaN (lib_ongaku_test.py:166-169).  The sharded searches defer that check:
    this reads their flags once, takes the maximum over the ranks (so that every rank raises together) and raises."""
    if not _NAN_FLAGS:
        return
    fl = torch.stack([x.reshape(()) for x in _NAN_FLAGS])
    f = torch.stack([(fl & 1).max(), ((fl >> 1) & 1).max()])       # [NaN distance, candidate-buffer overflow of the fused route]
    _NAN_FLAGS.clear()
    if dist.is_available() and dist.is_initialized():
        if _host_staged(f):
            f = f.cpu()
        dist.all_reduce(f, op=dist.ReduceOp.MAX)
    nan, over = (int(v) for v in f.tolist())
    from . import ops
    if nan:
        raise ops.KnnSvcError("containing nan")
    if over:            # on every rank together: the caller repeats the searches on the dot-matrix route (ops.retry_on_overflow)
        raise ops.KnnOverflow("fused kNN route: candidate buffer overflow on some rank")


def _hip_merge(part_dist, part_idx):
    from . import ops
    return ops.knn_merge(part_dist, part_idx)


def all_to_all_rows(t: torch.Tensor, send_rows, recv_rows) -> torch.Tensor:
    """Rows of ``t`` [sum(send_rows), ...] are dealt out in rank order (send_rows[r] rows go to rank r); returns the
    rows received, [sum(recv_rows), ...], grouped by sender in rank order.  One ``all_to_all_single`` (RCCL: every
    peer pair uses its own xGMI link), so a rank receives only what it asked for."""
    out = torch.empty((int(sum(recv_rows)),) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    if _host_staged(t):
        o = torch.empty(out.shape, dtype=out.dtype)
        dist.all_to_all_single(o, t.contiguous().cpu(), output_split_sizes=[int(r) for r in recv_rows],
                               input_split_sizes=[int(r) for r in send_rows])
        out.copy_(o)
        return out
    dist.all_to_all_single(out, t.contiguous(), output_split_sizes=[int(r) for r in recv_rows],
                           input_split_sizes=[int(r) for r in send_rows])
    return out


def sharded_knn_owned(q_all: torch.Tensor, owner_rows, pool_local: torch.Tensor, k: int = 32, local_topk=None, merge=None,
                      counts=None):
    """Replicated queries with OWNERS: ``q_all`` [sum(owner_rows), D] is the same on every rank and ordered rank-major
    (the first owner_rows[0] rows belong to rank 0, ...).  Every rank searches all of them in its own pool shard; the
    [., k] (distance, global row) lists are then exchanged with ONE all-to-all so that each rank receives, from every
    shard, only the lists of the rows it owns (ws x fewer bytes than all-gathering every list to every rank: BASELINE
    cfg 5 moves 98 MB instead of 786 MB per rank), and merges them.  Returns (idx, dist) for this rank's own rows."""
    local_topk = local_topk or _hip_local_topk
    merge = merge or _hip_merge
    rank, ws = world()
    if not (dist.is_available() and dist.is_initialized()):
        return local_topk(q_all, pool_local, k, 0)
    owner_rows = [int(r) for r in owner_rows]
    assert len(owner_rows) == ws and sum(owner_rows) == q_all.shape[0], (owner_rows, q_all.shape)
    counts = list(counts) if counts is not None else shard_rows(pool_local.shape[0], q_all.device)
    assert len(counts) == ws and counts[rank] == pool_local.shape[0], (counts, rank, pool_local.shape)
    idx, dst = local_topk(q_all, pool_local, k, sum(counts[:rank]))            # every row vs my shard, global ids
    mine = owner_rows[rank]
    recv = [mine] * ws                                                         # my rows' lists, one block per shard
    # ONE exchange for both halves of the lists: (distance bits, index) packed as three 32-bit words per entry.  Every collective
    # is a point where the searching stream waits for a kernel on RCCL's stream to be scheduled — next to a chip full of
    # another conversion's generator that costs a few hundred microseconds each, whatever the message size
    # (tools/bench_1rank_ab.sh: a one-rank group alone cost the stream pipeline 7 % of its throughput).
    packed = torch.empty(idx.shape + (3,), dtype=torch.int32, device=idx.device)
    packed[..., 0] = dst.view(torch.int32)
    packed[..., 1:] = idx.view(torch.int32).view(idx.shape + (2,))
    got = all_to_all_rows(packed, owner_rows, recv).view(ws, mine, k, 3)
    if mine == 0:
        return idx[:0], dst[:0]
    d_parts = got[..., 0].contiguous().view(torch.float32)
    i_parts = got[..., 1:].contiguous().view(torch.int64).view(ws, mine, k)
    return merge(d_parts, i_parts)


def sharded_knn(q_local: torch.Tensor, pool_local: torch.Tensor, k: int = 32, local_topk=None,
                merge=None, replicated: bool = False, counts=None):
    """Top-k of every rank's queries against the union of all ranks' pool shards.

    q_local [nq, D] (same nq on every rank), pool_local [np_r, D]: shards may differ in size (each must hold >= k
    rows); the global row of local row j on rank r is sum(np_0 .. np_{r-1}) + j, i.e. the row order of
    all_gather_rows_var(pool_local).  Returns (idx [nq, k] global rows, dist [nq, k]) for THIS rank's queries: the
    queries are all-gathered (KBs), searched in every shard, and the lists come back through one all-to-all
    (``sharded_knn_owned``) — each rank receives only its own queries' lists.
    ``replicated``: every rank holds the SAME queries and every rank needs the merged result (one conversion against a
    sharded pool, BASELINE cfg 4 single file): they are not gathered, each rank searches them once in its shard, the
    lists are all-gathered and every rank ends up with the same merged lists.
    ``counts``: the shard sizes if the caller already knows them (shard_rows() reads them back to the host — a
    synchronisation a stream pipeline must not have inside its steps)."""
    local_topk = local_topk or _hip_local_topk
    merge = merge or _hip_merge
    rank, ws = world()
    if not (dist.is_available() and dist.is_initialized()):
        return local_topk(q_local, pool_local, k, 0)
    nq = q_local.shape[0]                 # (a 1-rank group still walks the collective path: it is the same code)
    counts = list(counts) if counts is not None else shard_rows(pool_local.shape[0], q_local.device)
    assert len(counts) == ws and counts[rank] == pool_local.shape[0], (counts, rank, pool_local.shape)
    if replicated:
        idx, dst = local_topk(q_local, pool_local, k, sum(counts[:rank]))
        packed = torch.empty((1,) + idx.shape + (3,), dtype=torch.int32, device=idx.device)      # one all-gather for both halves
        packed[0, ..., 0] = dst.view(torch.int32)
        packed[0, ..., 1:] = idx.view(torch.int32).view(idx.shape + (2,))
        got = all_gather_rows(packed)
        return merge(got[..., 0].contiguous().view(torch.float32), got[..., 1:].contiguous().view(torch.int64).view((ws,) + idx.shape))
    q_all = all_gather_rows(q_local)                                           # [ws*nq, D], rank-major
    return sharded_knn_owned(q_all, [nq] * ws, pool_local, k, local_topk, merge, counts)


def contiguous_share(n: int):
    """[lo, hi) of n ordered units for this rank: contiguous, balanced ranges in rank order, so that concatenating the
    ranks' parts reproduces the single-process order (pool files keep their global row order — the concat cost's
    "next frame" is row + 1)."""
    rank, ws = world()
    base, rem = divmod(n, ws)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def my_share(items):
    """Independent work units (speaker folders in prematch, (source speaker, target speaker) pairs in dataset mode,
    source clips in BASELINE cfg 5) are dealt round-robin over the ranks — the per-utterance / per-speaker stages of
    SURVEY.md §8e shard with no collective at all.  Every rank must call this with the same, identically ordered list."""
    rank, ws = world()
    return [it for i, it in enumerate(items) if i % ws == rank]


def gather_paths(paths):
    """All ranks' lists of written files, concatenated in rank order (host objects; one small all_gather_object)."""
    if not (dist.is_available() and dist.is_initialized()):
        return list(paths)
    _rank, ws = world()
    out = [None] * ws
    dist.all_gather_object(out, list(paths))
    return [p for part in out for p in part]
