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


def free_port():
    """An ephemeral rendezvous port on the loopback interface, as a string for MASTER_PORT / --master-port.  Every
    multi-process test asks for its own: a fixed port is the next flake on a shared box."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return str(s.getsockname()[1])


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


# The driver runs `pytest tests -x -q -m gpu` under a 900 s step limit.  Budget: 450 s.  The NUMBER of GPU tests is
# capped here (a new case has to replace an old one, or be moved behind an opt-in like FHS_RUN_AS_WRITTEN_FULLSIZE); a
# single GPU test over 60 s is REPORTED in the terminal summary (it used to fail the run: under `-x` a slow fresh box
# then cost every test behind it -- the same way round 5's harness flake erased the parity evidence).
GPU_TEST_CAP = 440
GPU_SUITE_BUDGET_S = 450
GPU_TEST_LIMIT_S = 60

# `-x` stops at the first failure, so the ORDER decides what evidence a failure can cost: parity
# of the kernels first, then the char / string layer against the oracle and the golden vectors, then noise, splits,
# multi-rank, the compiled C host and the CLI, and the subprocess harness tests around bench.py LAST.
GPU_ORDER = ["test_gpu_pbs", "test_gpu_kat", "test_gpu_wide_parity", "test_gpu_ops", "test_gpu_fullsize",
             "test_gpu_fft_mode", "test_gpu_rotation_sharing", "test_gpu_noise", "test_gpu_margins", "test_gpu_skew",
             "test_gpu_split_long", "test_gpu_parallel", "test_gpu_c_host", "test_cli", "test_gpu_bench_contract"]
_SLOW = []


def _gpu_rank(item):
    name = os.path.splitext(os.path.basename(str(item.fspath)))[0]
    if name in GPU_ORDER:
        return GPU_ORDER.index(name)
    return len(GPU_ORDER) - 2          # unknown GPU files: before the CLI and the bench harness


def pytest_collection_modifyitems(config, items):
    n_gpu = sum(1 for it in items if it.get_closest_marker("gpu"))
    if n_gpu > GPU_TEST_CAP:
        raise pytest.UsageError("%d GPU tests collected, the cap is %d (tests/conftest.py: the suite's budget is %d s of the "
                                "driver's 900 s step)" % (n_gpu, GPU_TEST_CAP, GPU_SUITE_BUDGET_S))
    gpu = [it for it in items if it.get_closest_marker("gpu")]
    if gpu:                              # stable sort: the order inside a file stays; CPU tests keep theirs, in front
        cpu = [it for it in items if not it.get_closest_marker("gpu")]
        items[:] = cpu + sorted(gpu, key=_gpu_rank)


@pytest.hookimpl(hookwrapper=True)
def pytest_runtest_call(item):
    import time
    t0 = time.perf_counter()
    yield
    dt = time.perf_counter() - t0
    if item.get_closest_marker("gpu") and dt > GPU_TEST_LIMIT_S and not item.get_closest_marker("slow"):
        _SLOW.append((item.nodeid, dt))


def pytest_terminal_summary(terminalreporter):
    for nodeid, dt in _SLOW:
        terminalreporter.write_line("SLOW GPU TEST: %s took %.0f s (limit %d s, suite budget %d s)" % (
            nodeid, dt, GPU_TEST_LIMIT_S, GPU_SUITE_BUDGET_S))
