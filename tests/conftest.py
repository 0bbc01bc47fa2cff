import os
import sys

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")   # what the package's __init__ sets, here before any test initialises the device

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _limit_threads():
    # the tiny-model oracle runs are dominated by thread-pool overhead on a 256-thread host: a moderate pool is several times faster
    # (the SD-v1.5-size tests raise it again for their module)
    try:
        import torch
        torch.set_num_threads(min(16, os.cpu_count() or 1))
    except Exception:
        pass


def pytest_configure(config):
    _limit_threads()
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")
