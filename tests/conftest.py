import os
import sys

import pytest

# PyTorch-ROCm ships its own HIP / HSA runtime.  When a process uses both torch
# and libsmvs_hip.so (the multi-GPU tests: torch.distributed is the RCCL
# plumbing), torch has to come first so that ONE runtime serves both; loaded
# the other way round the two runtimes coexist and torch sees no GPU.
try:
    import torch  # noqa: F401
except ImportError:
    pass

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure; built on first use)."""
    from oracle import pyoracle
    pyoracle.lib()
    return pyoracle
