import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(autouse=True)
def _default_stream_per_test():
    """TcarEngine makes a process-wide high-priority stream the CURRENT stream (engine.use_priority_stream).  The op-level
    tests pass stream = NULL to the C-ABI, so every test starts (synchronised) on the default stream again."""
    yield
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.synchronize()
            torch.cuda.set_stream(torch.cuda.default_stream())
    except ImportError:
        pass
