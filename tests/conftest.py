import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def node():
    """Initialised library on cuda:0 (GPU tests only)."""
    import torch
    assert torch.cuda.is_available(), "GPU test without a GPU"
    import starneig_amd as S
    torch.cuda.set_device(0)
    torch.zeros(1, device="cuda")          # create the primary context first
    if not S.node_initialized():
        S.node_init(1, 1, S.NO_MESSAGES)
    yield S
    S.node_finalize()
