"""Device-resident pool-feature store.

The reference re-encodes both speaker pools from scratch for every (source speaker, target speaker)
pair in dataset mode and has its own pickle cache force-disabled (ddsp_prematch_dataset.py:1086-1087,
ddsp_matcher.py:1073-1112) — more than 90 % of the work of a ``bulk_match`` run.  With 288 GB of HBM the
per-file tensors (WavLM features 4 KB/frame, f0, harmonics, spectrum) simply stay on the device:
one entry per audio file, keyed by the file's identity (absolute path, size, mtime) and the encoder
that produced it, evicted least-recently-used beyond a byte budget.

    KNNSVC_POOL_CACHE_GB   budget in GiB (default 64, 0 disables the store)
"""
from __future__ import annotations

import os
from collections import OrderedDict


def file_key(path, encoder_tag) -> tuple:
    st = os.stat(path)
    f0p = os.path.splitext(str(path))[0] + "_f0.npy"
    f0s = os.stat(f0p) if os.path.isfile(f0p) else None
    return (os.path.abspath(str(path)), st.st_size, st.st_mtime_ns, None if f0s is None else (f0s.st_size, f0s.st_mtime_ns),
            encoder_tag)


class PoolCache:
    def __init__(self, budget_bytes: int | None = None):
        if budget_bytes is None:
            budget_bytes = int(float(os.environ.get("KNNSVC_POOL_CACHE_GB", "64")) * (1 << 30))
        self.budget = budget_bytes
        self.used = 0
        self.entries: "OrderedDict[tuple, tuple]" = OrderedDict()
        self.hits = self.misses = 0

    @staticmethod
    def _nbytes(value) -> int:
        return sum(t.numel() * t.element_size() for t in value.values() if t is not None)

    def get(self, key):
        e = self.entries.get(key)
        if e is None:
            self.misses += 1
            return None
        self.entries.move_to_end(key)
        self.hits += 1
        return e[0]

    def put(self, key, value: dict) -> None:
        if self.budget <= 0:
            return
        n = self._nbytes(value)
        if n > self.budget:
            return
        if key in self.entries:
            self.used -= self.entries.pop(key)[1]
        self.entries[key] = (value, n)
        self.used += n
        while self.used > self.budget and self.entries:
            _k, (_v, m) = self.entries.popitem(last=False)
            self.used -= m

    def clear(self) -> None:
        self.entries.clear()
        self.used = 0
