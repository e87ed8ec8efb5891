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
two lanes at once (`WavLMEncoder.encode_batch`, `Vocoder.forward`): heads that encode use one lane (bench.py), heads that
only match may use several (dataset mode), the generator runs on the single tail stream.  Replaying one encoder graph
from two lanes aborts the process (observed), it does not merely race.

HIP maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4); two streams that share a queue
serialise.  knn_svc_amd/__init__.py raises the default to 8 before the runtime starts.
"""
from __future__ import annotations

import torch


def _tensors(obj):
    if isinstance(obj, torch.Tensor):
        yield obj
    elif isinstance(obj, dict):
        for v in obj.values():
            yield from _tensors(v)
    elif isinstance(obj, (list, tuple)):
        for v in obj:
            yield from _tensors(v)


class LanePipeline:
    def __init__(self, device, lanes: int = 1):
        if lanes < 1:
            raise ValueError("lanes must be >= 1")
        self.device = torch.device(device)
        self.lanes = [torch.cuda.Stream(device=self.device) for _ in range(lanes)]
        # The tail carries the single-workgroup recurrences: high priority puts it (and its partner stream, see
        # matching._side_stream) on hardware queues of their own — normal-priority streams can collide with each
        # other on a queue (more so once RCCL has created its streams) but never with these — and lets a lone
        # workgroup take the first CU that frees up.
        import os
        self.tail_stream = torch.cuda.Stream(device=self.device, priority=int(os.environ.get("KNNSVC_TAIL_PRIORITY", "-1")))

    def run(self, items, head, tail=None):
        """results[i] = tail(item_i, head(item_i)) (or head(item_i) without a tail), in item order.

        ``head`` and ``tail`` must only enqueue GPU work (no host synchronisation — a ``.item()`` inside them
        would serialise the pipeline); they see their lane / the tail stream as the current stream."""
        cur = torch.cuda.current_stream(self.device)
        for s in self.lanes + [self.tail_stream]:
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
            with torch.cuda.stream(self.tail_stream):
                self.tail_stream.wait_event(done)
                for t in _tensors(h):
                    t.record_stream(self.tail_stream)
                out.append(tail(item, h))
        for s in self.lanes + [self.tail_stream]:
            cur.wait_stream(s)
        for t in _tensors(out):
            t.record_stream(cur)
        return out
