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


# The driver runs `pytest tests -m gpu` under a 900 s step limit; round 4's suite took 519 s and had grown every round
# (VERDICT r4 item 8).  Budget: 450 s.  Two guards: the NUMBER of GPU tests is capped here (a new case has to replace an old
# one, or be moved behind an opt-in like FHS_RUN_AS_WRITTEN_FULLSIZE), and any single GPU test over 60 s fails the run.
GPU_TEST_CAP = 440
GPU_SUITE_BUDGET_S = 450
GPU_TEST_LIMIT_S = 60


def pytest_collection_modifyitems(config, items):
    n_gpu = sum(1 for it in items if it.get_closest_marker("gpu"))
    if n_gpu > GPU_TEST_CAP:
        raise pytest.UsageError("%d GPU tests collected, the cap is %d (tests/conftest.py: the suite's budget is %d s of the "
                                "driver's 900 s step)" % (n_gpu, GPU_TEST_CAP, GPU_SUITE_BUDGET_S))


@pytest.hookimpl(hookwrapper=True)
def pytest_runtest_call(item):
    import time
    t0 = time.perf_counter()
    yield
    dt = time.perf_counter() - t0
    if item.get_closest_marker("gpu") and dt > GPU_TEST_LIMIT_S and not item.get_closest_marker("slow"):
        pytest.fail("%s took %.0f s: a single GPU test may take %d s at most (suite budget %d s)" % (
            item.nodeid, dt, GPU_TEST_LIMIT_S, GPU_SUITE_BUDGET_S), pytrace=False)
