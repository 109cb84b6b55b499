"""A shortened run of tools/soak.py: frames whose kernel selection changes from one to the next (visible count above / below
2 M, heavy chunks present / absent), one at a time and four in flight -- every render of a pose bit-identical to its first,
order checks at (0, 0), counts stable.  The full run (3000 frames per mode) is `python tools/soak.py` on the GPU box.
Runs as its own process, like bench.py: GPU_MAX_HW_QUEUES has to be in the environment before HIP initialises."""
import os
import subprocess
import sys

import pytest

from tests.conftest import ROOT

pytestmark = pytest.mark.gpu


def test_soak_frames_with_changing_kernel_selection():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "soak.py"), "--frames", "500"], cwd=ROOT,
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    assert "soak: OK" in p.stdout
