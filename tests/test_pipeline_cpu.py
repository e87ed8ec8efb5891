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
