"""Stream scheduler for back-to-back conversions (the reference has no equivalent: its
loops in ddsp_matcher.py:1073-1150 run one utterance at a time on the default stream).

A conversion is two kinds of GPU work: chip-wide kernels (WavLM encoder, kNN, vocoder) and the
frame-sequential recurrences of the match stage (concat re-selection, Adam smoothness weights), which are
single-workgroup kernels that occupy 2-4 of the 256 CUs for ~11 ms.  Run one after the other, the chip idles
during the recurrences.  ``LanePipeline`` enqueues the *head* of item i on lane stream i mod L and the *tail*
of item i on one tail stream behind an event, so the recurrences of one item run underneath the chip-wide
work of its neighbours.  Every item still does all of its work; only the order of enqueueing changes, and the
results are bit-identical to the sequential order (same kernels, same inputs).

Constraint: a captured hipGraph owns its input / output / scratch buffers, so the SAME graph must not be replayed from
two streams at once (`WavLMEncoder.encode_batch`, `Vocoder.forward`): heads that encode use one lane (bench.py), heads that
only match may use several (dataset mode).  Replaying one encoder graph from two lanes aborts the process (observed), it
does not merely race.  The tail runs on THREE tail streams when there are several lanes (item i on tail i mod 3): the
generator keeps one graph instance and one memory pool per tail (`current_tail()`), so up to three generators of different
items are in flight — each is a chain of short launches that leave most of the chip idle on their own.

HIP maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4); two streams that share a queue
serialise.  knn_svc_amd/__init__.py raises the default to 8 before the runtime starts.
"""
from __future__ import annotations

import threading

import torch


_TAIL = threading.local()


def current_tail() -> int:
    """Index of the tail stream the calling code runs on (0 outside LanePipeline.run or with one tail)."""
    return getattr(_TAIL, "k", 0)


def _tensors(obj):
    if isinstance(obj, torch.Tensor):
        yield obj
    elif isinstance(obj, dict):
        for v in obj.values():
            yield from _tensors(v)
    elif isinstance(obj, (list, tuple)):
        for v in obj:
            yield from _tensors(v)


# ---- stream census.  HIP maps streams onto hardware queues; the measurements of round 3 (profiles/r03_tail_streams_sweep.txt) fit
# FOUR queues for high-priority streams: up to four tail streams are fine, a FIFTH shares a queue with a stream it waits for and
# the pipeline collapses (cfg 5 share 3170 -> 530 xRT).  Every stream the package creates goes through new_stream(), so the number
# of live high-priority streams is known, and the tail count is a function of it with a hard cap below the cliff.
MAX_TAILS = 4                      # never more, whatever KNNSVC_TAILS says: the fifth is the cliff
HIPRI_QUEUES = 4
_STREAMS = []                      # (owner or None, priority, kind): owner = weak reference to the LanePipeline that made it


_PROBE = {"alone_ms": {}, "tries": 0, "rejected": 0, "log": []}
_KNOWN = []                        # (owner weakref or None, kind, stream): the package's live streams, for overlap_with="all"
PROBE_BLOCKS, PROBE_SPIN = 200_000, 2000          # the probe launch: 200 000 one-wave workgroups spinning ~1 us each (~47 us alone)


def _probe_launch(streams, device) -> float:
    """ms from the first launch's start to the last one's end: one dispatch-bound probe launch (knnsvc_probe_dispatch) per stream."""
    from . import _lib
    lib = _lib.load()
    with torch.cuda.device(device):
        for st in streams:
            st.synchronize()
        ev = []
        for st in streams:
            with torch.cuda.stream(st):
                e0 = torch.cuda.Event(enable_timing=True); e0.record()
                lib.knnsvc_probe_dispatch(PROBE_BLOCKS, PROBE_SPIN, st.cuda_stream)
                e1 = torch.cuda.Event(enable_timing=True); e1.record()
                ev.append((e0, e1))
        for st in streams:
            st.synchronize()
        return max(ev[0][0].elapsed_time(e1) for _e0, e1 in ev)


def stream_contention(a, b, device) -> float:
    """How badly do streams a and b get in each other's way?  -> (time of one dispatch-bound launch on EACH, started together) /
    (time of one alone).  HIP maps streams onto a few hardware queues, and which streams share a queue — or a dispatch pipe — is an
    accident of how many streams the PROCESS has made before: round 4's bench step moved between 34.8 and 38.3 ms with it
    (profiles/r04_rank1_rccl_ab.txt), round 5's — without the generator's branch streams — still between 34.8 and 38.5
    (profiles/r05_stream_robustness_ab_before_probe.txt).  So it is MEASURED.  Three classes show up on MI355X (47 us alone):
    ~1.5 (70 us: side by side), ~2.1 (98 us: one hardware queue, strictly one after the other — a single-thread spin kernel on each
    shows the same pairs) and ~3 (140-150 us: they overlap, but their workgroup dispatch collides: a pair like that — encoder lane
    against the match stage's partner stream — is what the 37 ms placements had and the 34.5 ms ones did not)."""
    key = (torch.device(device).index, PROBE_BLOCKS)
    if key not in _PROBE["alone_ms"]:
        _probe_launch([b], device)                       # (first launch: code object load)
        _PROBE["alone_ms"][key] = min(_probe_launch([b], device) for _ in range(3))
    alone = _PROBE["alone_ms"][key]
    together = _probe_launch([a, b], device)
    return together / alone


CONTENTION_OK = 1.75               # between the "side by side" (~1.5) and the "one queue" (~2.1) classes


def new_stream(device, priority: int = 0, kind: str = "", owner=None, overlap_with=(), must=()) -> "torch.cuda.Stream":
    """Every stream of the package.  Streams made for a LanePipeline count while that pipeline is alive (weak reference to the
    PIPELINE — torch's stream objects themselves do not survive being weakly referenced: the process dies in the garbage
    collector); all others are cached for the life of the process by their makers.
    ``overlap_with``: streams whose kernels this one has to run BESIDE — a list, or "all" = the package's live lane / tail / partner /
    kNN streams on the device.  torch hands out pool streams round-robin and HIP maps them onto hardware queues; a candidate whose
    measured contention (stream_contention) with any listed stream is above CONTENTION_OK is set aside and the next pool stream is
    tried, up to 16; if none passes, the least contended candidate is taken — least against ``must`` first (the streams that
    matter most: a partner's own stream, the lanes and tails), then against the rest.  KNNSVC_STREAM_PROBE=0: the first pool
    stream, unmeasured."""
    import os
    import weakref
    s = torch.cuda.Stream(device=device, priority=priority)
    if isinstance(overlap_with, str):
        idx = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
        _KNOWN[:] = [e for e in _KNOWN if e[0] is None or e[0]() is not None]
        overlap_with = [st for _o, k, st in _KNOWN if st.device.index == idx and k in ("lane", "tail", "partner", "knn")]
    must = [m for m in must if m is not None]
    rest = [o for o in overlap_with if not any(o.cuda_stream == m.cuda_stream for m in must)]
    if (must or rest) and os.environ.get("KNNSVC_STREAM_PROBE", "1") != "0" and not torch.cuda.is_current_stream_capturing():
        best = None
        torch.cuda.synchronize(device)      # the probe times launches: nothing else may be running (once per stream, at its creation)
        klass = lambda r: 0 if r <= CONTENTION_OK else (1 if r <= 2.4 else 2)       # side by side / one queue / colliding dispatch
        for _try in range(16):
            _PROBE["tries"] += 1
            w_must = max((stream_contention(o, s, device) for o in must), default=0.0)
            w_rest = max((stream_contention(o, s, device) for o in rest), default=0.0)
            score = (klass(w_must), klass(w_rest), w_must, w_rest)
            _PROBE["log"].append((kind, int(priority), round(w_must, 2), round(w_rest, 2)))
            if best is None or score < best[0]:
                best = (score, s)
            if score[:2] == (0, 0):
                break
            _PROBE["rejected"] += 1
            s = torch.cuda.Stream(device=device, priority=priority)
        s = best[1]
        if os.environ.get("KNNSVC_STREAM_PROBE_LOG") == "1":
            import sys
            print(f"[stream probe] {kind} stream (priority {priority}): contention {best[0][2]:.2f} against its {len(must)} partner / lane / tail stream(s), "
                  f"{best[0][3]:.2f} against {len(rest)} other(s), after {_try + 1} candidate(s)", file=sys.stderr)
    _STREAMS.append((weakref.ref(owner) if owner is not None else None, int(priority), kind))
    _KNOWN.append((weakref.ref(owner) if owner is not None else None, kind, s))
    return s


def _live():
    _STREAMS[:] = [e for e in _STREAMS if e[0] is None or e[0]() is not None]
    return [(p, k) for _o, p, k in _STREAMS]


def rccl_streams():
    """(streams, high-priority streams) a live RCCL process group brings: torch's ProcessGroupNCCL runs its collectives on one
    stream per device taken from the stream pool, high priority only if the group was created with is_high_priority_stream."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_backend() != "nccl":
        return 0, 0
    hi = 0
    try:
        pg = dist.distributed_c10d._get_default_group()
        hi = 1 if pg._get_backend(torch.device("cuda")).options.is_high_priority_stream else 0
    except Exception:               # private API: unknown -> assume the worse case
        hi = 1
    return 1, hi


def stream_census(device=None):
    """Live streams this process created through the package, by priority class, plus RCCL's."""
    live = _live()
    n_rccl, hi_rccl = rccl_streams()
    return {"high": sum(1 for p, _k in live if p < 0) + hi_rccl, "normal": sum(1 for p, _k in live if p >= 0) + n_rccl - hi_rccl,
            "rccl": n_rccl, "tails": sum(1 for p, k in live if k == "tail")}


def tail_budget(requested: int, device=None) -> int:
    """Tail streams a new pipeline may create: the request, capped (a) at MAX_TAILS, (b) at the high-priority queues that are
    not already taken by high-priority streams OTHER than tails and their partner / branch streams (those follow a tail's
    work in order and were part of every measurement), (c) at 2 under an RCCL group — its stream is one more live stream next to
    lanes and tails, and the five-stream cliff has only been measured without it (no multi-GPU node so far): one more of
    headroom until it has."""
    c = stream_census(device)
    n = max(1, min(int(requested), MAX_TAILS))
    live = _live()
    other_hi = sum(1 for p, k in live if p < 0 and k not in ("tail", "partner", "branch")) + rccl_streams()[1]
    n = max(1, min(n, HIPRI_QUEUES - other_hi))
    if c["rccl"] and n > 2:
        n = 2
    return n


_VETTED = {}                       # (device index, kind, position, priority) -> stream: chosen once by measurement, reused by every pipeline


def vetted_stream(device, kind: str, position: int, priority: int, must) -> "torch.cuda.Stream":
    """The position-th lane / tail stream of this device.  Chosen ONCE (new_stream: measured contention against the package's other
    streams) and handed to every LanePipeline that asks — dataset mode and the serving converter make a pipeline per call, and a
    stream set that was measured to run side by side is worth keeping (their partner streams, cached per stream, then stay too)."""
    dev = torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    key = (idx, kind, int(position), int(priority))
    if key not in _VETTED:
        _VETTED[key] = new_stream(dev, priority=priority, kind=kind, overlap_with="all", must=must)
    return _VETTED[key]


class LanePipeline:
    def __init__(self, device, lanes: int = 1):
        if lanes < 1:
            raise ValueError("lanes must be >= 1")
        self.device = torch.device(device)
        # ONE lane (one conversion after the other: bench.py's north-star pipeline, special_match): lane, tail and their partner
        # streams are few enough to give every one a hardware queue it does not fight over — they are chosen by measurement, once
        # (vetted_stream).  SEVERAL lanes (dataset mode, serving): 3 lanes + their 3 partners + 3 tails + the kNN stream are more
        # streams than the runtime has hardware queues, some must share, and the greedy measured choice came out WORSE than torch's
        # round-robin pool order (same box, bench.py's cfg 5 share / cfg 3: 2591 / 2119 xRT against 2868 / 2553): those keep the
        # pool order, a fresh set per pipeline.
        self.measured = lanes == 1
        self.lanes = []
        for i in range(lanes):
            self.lanes.append(vetted_stream(self.device, "lane", i, 0, list(self.lanes)) if self.measured
                              else new_stream(self.device, kind="lane", owner=self))
        # The tail carries the single-workgroup recurrences: high priority puts it (and its partner stream, see
        # matching._side_stream) on hardware queues of their own — normal-priority streams can collide with each
        # other on a queue (more so once RCCL has created its streams) but never with these — and lets a lone
        # workgroup take the first CU that frees up.
        import os
        pr = int(os.environ.get("KNNSVC_TAIL_PRIORITY", "-1"))
        # THREE tail streams, item i's tail on stream i mod 3 (KNNSVC_TAILS=n): the generator is ~110 launches most of which are a
        # single round of workgroups waiting on latencies, so one generator leaves most of the chip's issue slots idle and two or
        # three in flight (different items) fill them — cfg 5 share 2620 -> 3140 xRT with two tails, 3070-3140 with three / four;
        # dataset mode on 5-10 s utterances (cfg 3) 1405 -> 1840 -> 2080 -> 2080 (tools/cfg5_product_bench.py, tools/cfg3_bench.py).
        # A consumer that replays hipGraphs keeps one graph instance and one memory pool per tail (Vocoder.forward, current_tail()).
        # Only with several lanes (dataset mode, serving: the generator runs with its ResBlock branches in series there): next to
        # the generator's own branch streams the extra tails oversubscribe the hardware queues — the depth-2 pipeline of bench.py
        # (one lane) went from 35 to 470-750 ms per step with three tails.
        # Why not more: with HIP's 8 hardware queues four tails are a little faster still (cfg 3: 2140 -> 2400 xRT) but FIVE collapse
        # (cfg 5 share 3170 -> 530 xRT: streams that depend on each other end up sharing queues), and 12 / 16 queues are slower at
        # any tail count (tools/final_profiles.sh, profiles/r03_tail_streams_sweep.txt).  The numbers fit FOUR hardware queues for
        # high-priority streams: four tails fill them, a fifth shares one with a tail it waits for — and the generator's own
        # branch streams inherit the tail's priority, which is why tails + parallel branches collapsed at 3 x 3 streams.  Three
        # leaves one such queue spare for whatever else the process created at that priority; fewer when the caller asks for
        # more than three lanes.
        self.n_tails_requested = int(os.environ.get("KNNSVC_TAILS", "0")) or (min(3, max(1, 6 - lanes)) if lanes > 1 else 1)
        n_tails = tail_budget(self.n_tails_requested, self.device)
        self.tail_streams = []
        for i in range(n_tails):
            self.tail_streams.append(vetted_stream(self.device, "tail", i, pr, self.lanes + self.tail_streams) if self.measured
                                     else new_stream(self.device, priority=pr, kind="tail", owner=self))
        self.tail_stream = self.tail_streams[0]

    def run(self, items, head, tail=None):
        """results[i] = tail(item_i, head(item_i)) (or head(item_i) without a tail), in item order.

        ``head`` and ``tail`` must only enqueue GPU work (no host synchronisation — a ``.item()`` inside them
        would serialise the pipeline); they see their lane / the tail stream as the current stream."""
        cur = torch.cuda.current_stream(self.device)
        for s in self.lanes + self.tail_streams:
            s.wait_stream(cur)
        out = []
        for i, item in enumerate(items):
            lane = self.lanes[i % len(self.lanes)]
            with torch.cuda.stream(lane):
                h = head(item)
                done = lane.record_event()
            if tail is None:
                out.append(h)
                continue
            ts = self.tail_streams[i % len(self.tail_streams)]
            with torch.cuda.stream(ts):
                ts.wait_event(done)
                for t in _tensors(h):
                    t.record_stream(ts)
                _TAIL.k = i % len(self.tail_streams)
                try:
                    out.append(tail(item, h))
                finally:
                    _TAIL.k = 0
        for s in self.lanes + self.tail_streams:
            cur.wait_stream(s)
        for t in _tensors(out):
            t.record_stream(cur)
        return out
