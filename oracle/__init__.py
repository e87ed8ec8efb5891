"""CPU oracle for the kNN-SVC inference hot path.  TEST INFRASTRUCTURE ONLY.

This package restates, in plain torch-CPU / numpy, the algorithm of every
reference function on the path (each function cites the reference file:line
it follows).  It is the checker for the HIP path — nothing here is shipped or
measured as the product:

* allowed importers: ``tests/``, ``__graft_entry__.smoke()`` and the
  ``cpu_baseline`` leg of ``bench.py``;
* ``knn_svc_amd`` never imports it and fails loudly when the HIP library is
  missing instead of falling back to anything on the CPU.

Pinning: the reference ships no tests or golden vectors for this path
(SURVEY.md §4), so the oracle is pinned against outputs of the reference
itself, imported in the build container by ``tests/gen_golden.py`` (seeded
inputs, outputs committed under ``tests/golden/``).  Two boundaries stay
**parity unpinned** because their third-party implementation is absent
offline: ``torchaudio`` (load/resample/Spectrogram — restated from its
documented defaults, cross-checked against a direct DFT) and
``pyworld.harvest`` (never needed when ``<stem>_f0.npy`` exists).
"""
