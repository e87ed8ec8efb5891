"""Tail-stream budget of the stream pipeline (knn_svc_amd/pipeline.py): a function of the live streams with a hard cap below
the measured five-stream cliff (VERDICT r3 #2c).  Host logic only — no GPU."""
from knn_svc_amd import pipeline


class _S:
    pass


def _with(streams, rccl, fn):
    keep, keep_r = pipeline._STREAMS[:], pipeline.rccl_streams
    objs = [_S() for _ in streams]
    pipeline._STREAMS[:] = [((lambda o=o: o), p, k) for o, (p, k) in zip(objs, streams)]
    pipeline.rccl_streams = lambda: rccl
    try:
        return fn()
    finally:
        pipeline._STREAMS[:] = keep
        pipeline.rccl_streams = keep_r


def test_hard_cap_is_below_the_cliff():
    assert _with([], (0, 0), lambda: [pipeline.tail_budget(n) for n in (0, 1, 3, 4, 5, 16)]) == [1, 1, 3, 4, 4, 4]


def test_tails_partners_and_branches_do_not_count_against_the_budget():
    live = [(-1, "tail")] * 3 + [(-1, "partner")] * 3 + [(0, "lane")] * 3 + [(0, "knn")]
    assert _with(live, (0, 0), lambda: pipeline.tail_budget(3)) == 3
    c = _with(live, (0, 0), lambda: pipeline.stream_census())
    assert c == {"high": 6, "normal": 4, "rccl": 0, "tails": 3}


def test_foreign_high_priority_streams_and_rccl_shrink_it():
    assert _with([(-1, "other")] * 2, (0, 0), lambda: pipeline.tail_budget(4)) == 2
    assert _with([(-1, "other")] * 7, (0, 0), lambda: pipeline.tail_budget(4)) == 1
    assert _with([], (1, 0), lambda: pipeline.tail_budget(3)) == 2            # a normal-priority RCCL stream: conservative 2
    assert _with([], (1, 1), lambda: pipeline.tail_budget(4)) == 2
    assert _with([(-1, "other")] * 3, (1, 1), lambda: pipeline.tail_budget(4)) == 1
