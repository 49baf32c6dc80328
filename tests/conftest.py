import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950) device; run with -m gpu")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def gpu_device():
    """The device under test.  On a GPU box a missing device / library is a hard failure, not a skip."""
    import torch
    assert torch.cuda.is_available(), "GPU tests need a ROCm device"
    name = torch.cuda.get_device_properties(0).gcnArchName
    assert name.startswith("gfx950"), "expected gfx950 (MI355X), got %s" % name
    return torch.device("cuda", 0)
