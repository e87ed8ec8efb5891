import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(ROOT / "tests" / "golden" / f"{name}.npz")
    return load


# Order of the GPU suite: the driver runs `pytest -m gpu -x`, so a late failure hides everything behind it.  The tests that compare
# with reference-generated goldens (models, end-to-end, prematch) run first, then the oracle comparisons at benchmark sizes, then the
# kernel-level and self-consistency tests.
# test_gpu_dist2 has to stay in front: it spawns its ranks and must do so before this interpreter has initialised the GPU.
_GPU_ORDER = ["test_gpu_dist2", "test_gpu_models", "test_gpu_product", "test_gpu_fulllength", "test_gpu_fullsize", "test_gpu_f0",
              "test_gpu_range", "test_gpu_kernels", "test_gpu_race"]


def pytest_collection_modifyitems(config, items):
    def rank(item):
        name = Path(str(item.fspath)).stem
        return _GPU_ORDER.index(name) if name in _GPU_ORDER else -1        # CPU files keep their place in front
    items.sort(key=rank)                                                    # stable: order inside a file is kept
