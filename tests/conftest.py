import os
import sys

import pytest

# PyTorch-ROCm bundles its own copy of the HIP runtime (torch/lib/libamdhip64.so); libmsplat.so links the system one.  Whichever
# is loaded FIRST serves the whole process; when libmsplat.so came first, torch's later CUDA initialisation failed with "No HIP
# GPUs are available" (seen r4 on a test selection in which no earlier test had imported torch).  bench.py imports torch first too.
try:
    import torch  # noqa: F401
except ImportError:
    pass

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """libmsplat.so + the oracle must exist; build them once (hipcc cross-compiles without a GPU)."""
    import __graft_entry__ as g
    g.build()
    yield


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def has_gpu():
    try:
        import ctypes
        hip = ctypes.CDLL("libamdhip64.so")
        n = ctypes.c_int(0)
        return hip.hipGetDeviceCount(ctypes.byref(n)) == 0 and n.value > 0
    except OSError:
        return False
