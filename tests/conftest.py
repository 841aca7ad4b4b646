import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(autouse=True)
def _seed_per_test(request):
    """Every test starts from a seed derived from its own id.  The fused models take their dropout seed from ``torch.initial_seed()``
    (like the reference takes its masks from the global generator); left unseeded that is a fresh random number per process, and the
    fp32-vs-fp64 gradient comparisons then ran on a different dropout mask every time -- on a few masks in a hundred an activation sits
    within rounding of a ReLU kink, the two precisions take different branches and a max-norm gate of 5e-4 trips (seen once in six runs
    of tests/test_fcstgnn_gpu.py)."""
    import zlib

    import torch
    torch.manual_seed(zlib.crc32(request.node.nodeid.encode()) & 0x7FFFFFFF)
    yield
