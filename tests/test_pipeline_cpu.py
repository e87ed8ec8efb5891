"""Tail-stream budget of the stream pipeline (knn_svc_amd/pipeline.py): a function of the live streams with a hard cap below
the measured five-stream cliff (VERDICT r3 #2c).  Host logic only — no GPU."""
from knn_svc_amd import pipeline


class _S:
    pass


def _with(streams, rccl, fn):
    keep, keep_r = pipeline._STREAMS[:], pipeline.rccl_streams
    pipeline._STREAMS[:] = [(None, p, k) for (p, k) in streams]
    pipeline.rccl_streams = lambda: rccl
    try:
        return fn()
    finally:
        pipeline._STREAMS[:] = keep
        pipeline.rccl_streams = keep_r


def test_streams_of_a_dead_pipeline_stop_counting():
    class P:
        pass
    import weakref
    p = P()
    keep = pipeline._STREAMS[:]
    try:
        pipeline._STREAMS[:] = [(weakref.ref(p), -1, "other"), (None, -1, "other")]
        assert len(pipeline._live()) == 2
        del p
        assert pipeline._live() == [(-1, "other")]
    finally:
        pipeline._STREAMS[:] = keep


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


def test_request_queue_isolates_a_batch_that_fails_as_a_whole():
    """serving.RequestQueue._run with a stand-in converter: a request that only fails INSIDE the conversion takes its batch
    down once; the batch is then re-run request by request and only the offender's Future carries the exception."""
    import torch
    from concurrent.futures import Future
    from knn_svc_amd import serving

    class Conv:
        class vc:
            device = torch.device("cpu")
        calls = []

        def load_checked(self, s):
            if s == "unloadable":
                raise ValueError("bad request")
            return s, None

        def convert(self, sources, loaded=None):
            self.calls.append(list(sources))
            if "poison" in sources:
                raise RuntimeError("containing nan")
            return [torch.tensor([float(len(s))]) for s in sources]

    rq = serving.RequestQueue.__new__(serving.RequestQueue)
    rq.conv, rq.isolated = Conv(), 0
    batch = [(s, Future()) for s in ("a", "unloadable", "poison", "bcd")]
    rq._run(batch)
    assert rq.isolated == 1 and Conv.calls == [["a", "poison", "bcd"], ["a"], ["poison"], ["bcd"]]
    assert float(batch[0][1].result()) == 1.0 and float(batch[3][1].result()) == 3.0
    assert isinstance(batch[1][1].exception(), ValueError) and isinstance(batch[2][1].exception(), RuntimeError)


def test_cat_rows_is_a_view_for_adjacent_parts_and_a_copy_otherwise():
    import torch
    from knn_svc_amd.wavlm import cat_rows
    b = torch.arange(60.).reshape(3, 5, 4)
    whole = cat_rows([b[i, :5] for i in range(3)])
    assert whole.data_ptr() == b.data_ptr() and torch.equal(whole, b.reshape(15, 4))
    assert cat_rows([b[1]]).data_ptr() == b[1].data_ptr()
    ragged = [b[0, :4], b[1, :5]]                                  # a gap between the parts: a real concatenation
    out = cat_rows(ragged)
    assert out.data_ptr() != b.data_ptr() and torch.equal(out, torch.cat(ragged))
    other = [b[0], torch.ones(2, 4)]                               # different buffers
    assert torch.equal(cat_rows(other), torch.cat(other))


def test_new_stream_passes_over_colliding_candidates_and_prefers_the_streams_that_matter(monkeypatch):
    """pipeline.new_stream's choice among torch's pool streams by MEASURED contention (round 5), with the measurement and the pool
    replaced by tables: a candidate that collides with a stream it has to run beside is passed over; when no candidate is clean the
    one that is clean against the ``must`` streams wins over one that is clean against the rest; KNNSVC_STREAM_PROBE=0 takes the
    first pool stream unmeasured."""
    import torch

    class FakeStream:
        n = 0

        def __init__(self, device=None, priority=0):
            FakeStream.n += 1
            self.cuda_stream, self.priority, self.device = FakeStream.n, priority, torch.device("cuda", 0)
    monkeypatch.setattr(torch.cuda, "Stream", FakeStream)
    monkeypatch.setattr(torch.cuda, "is_current_stream_capturing", lambda: False)
    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a, **k: None)
    lane, tail, other = FakeStream(), FakeStream(), FakeStream()          # ids 1, 2, 3
    table = {}
    monkeypatch.setattr(pipeline, "stream_contention", lambda a, b, dev: table.get((a.cuda_stream, b.cuda_stream), 1.5))
    keep_s, keep_k = pipeline._STREAMS[:], pipeline._KNOWN[:]
    try:
        # candidates 4 and 5 share a queue with the lane (2.1) / collide in dispatch with the tail (3.0); 6 is clean
        table.update({(1, 4): 2.1, (2, 5): 3.0})
        s = pipeline.new_stream(0, kind="partner", overlap_with=[lane, tail, other], must=[lane, tail])
        assert s.cuda_stream == 6
        # no clean candidate at all: 7, 9, 11, ... collide with the lane, the even ones only with `other` -> an even one is taken
        table.clear()
        for c in range(7, 40):
            table[(1, c) if c % 2 else (3, c)] = 2.1
        s = pipeline.new_stream(0, kind="partner", overlap_with=[lane, tail, other], must=[lane, tail])
        assert s.cuda_stream % 2 == 0 and s.cuda_stream >= 8
        first = FakeStream.n + 1
        monkeypatch.setenv("KNNSVC_STREAM_PROBE", "0")
        assert pipeline.new_stream(0, kind="partner", overlap_with=[lane], must=[lane]).cuda_stream == first
    finally:
        pipeline._STREAMS[:] = keep_s
        pipeline._KNOWN[:] = keep_k
