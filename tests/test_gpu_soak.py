"""A shortened run of tools/soak.py: frames whose kernel selection changes from one to the next (visible count above / below
2 M, heavy chunks present / absent), one at a time and four in flight -- every render of a pose bit-identical to its first,
order checks at (0, 0), counts stable.  The full run (3000 frames per mode) is `python tools/soak.py` on the GPU box."""
import importlib.util
import os

import pytest

from tests.conftest import ROOT

pytestmark = pytest.mark.gpu


def test_soak_frames_with_changing_kernel_selection():
    spec = importlib.util.spec_from_file_location("msplat_soak", os.path.join(ROOT, "tools", "soak.py"))
    soak = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(soak)
    assert soak.main(["--frames", "500"]) == 0
