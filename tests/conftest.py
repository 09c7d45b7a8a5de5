import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs an MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session")
def gpu():
    """Skip-free guard: GPU tests must fail, not skip, when the engine cannot run."""
    import torch
    assert torch.cuda.is_available(), "a -m gpu test ran on a machine without a GPU"
    return torch.cuda.get_device_name(0)
