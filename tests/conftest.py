import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950); run with -m gpu on the GPU box")


@pytest.fixture(scope="session")
def hip_lib():
    """The in-tree HIP library; building it is part of __graft_entry__.build()."""
    from deepmod_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return _lib.load()


@pytest.fixture(scope="session")
def gpu_device(hip_lib):
    n = hip_lib.dm_device_count()
    if n < 1:
        pytest.fail("GPU test selected but no gfx950 device is visible (no fallback path exists)")
    return 0


def trained_like_weights():
    """Weights with the statistics of a TRAINED BiLSTM (tests/golden/make_trained_like.py: the exact architecture trained with torch
    autograd in the build container; the reference's own .data shards are absent): name -> float32 array, like synth.synthetic_weights."""
    import numpy as np
    z = np.load(os.path.join(GOLDEN, "trained_like_weights.npz"))
    return {k.replace("|", "/"): np.ascontiguousarray(z[k], dtype=np.float32) for k in z.files}
