"""knn_svc_amd — MI355X-native kNN-SVC inference path.

(The spec names the package ``knn-svc_amd``; a hyphen is not a legal Python
identifier, so the importable name uses an underscore.)

Python here is the host side only: it mirrors the reference's entry points
(``ddsp_hubconf.knn_vc``, ``KNeighborsVC.special_match`` / ``bulk_match`` /
``vocode``, the ``ddsp_inference.py`` CLI) and calls ``libknnsvc_hip.so`` — a
C-ABI library of hand-written gfx950 kernels (``csrc/``, ``include/knnsvc_hip.h``)
— through ctypes.  There is no CPU fallback: every op raises if the library
is missing.
"""
import os as _os

# One hardware queue per HIP stream (the runtime's default is 4): the stream scheduler in pipeline.py and the
# two-branch match stage use up to 7 streams, and two streams on one queue serialise.  Read by the HIP runtime
# when it initialises, i.e. at the first torch.cuda call — importing this package first is enough.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

__all__ = ["config", "synthetic", "audio_io"]
