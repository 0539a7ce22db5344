import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    # the oracle is small-tensor fp32 torch: on the many-core GPU box the default (one thread per core) is several times
    # SLOWER than a modest pool because of fork/join overhead
    try:
        import torch
        torch.set_num_threads(min(16, os.cpu_count() or 1))
    except Exception:
        pass
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no HIP device is visible")
    from videovanish_amd import hip
    hip.lib()  # fails loudly if libvvhip.so is missing
    return torch.device("cuda:0")
