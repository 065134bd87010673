import os
import sys

import pytest

# PyTorch bundles its own HIP runtime (same SONAME as /opt/rocm's): whichever is loaded first serves the whole process,
# and torch cannot initialise the GPU on the other one.  Tests that hand torch tensors to the library (device-pointer
# entry points, RCCL exchange) therefore need torch imported before libfhestring_hip.so, like bench.py does.
try:
    import torch  # noqa: F401
except ImportError:  # pragma: no cover
    torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SEED = 0xF5E57121


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    config.addinivalue_line("markers", "slow: minutes of GPU time (still part of -m gpu)")


@pytest.fixture(scope="session")
def oracle_keys():
    from oracle import core
    return core.Keys(SEED)


@pytest.fixture(scope="session")
def oracle_sk(oracle_keys):
    from oracle import core
    return core.ServerKey(oracle_keys)
