import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # Multi-process GPU tests (tests/test_dist_gpu.py) take their children from a fork server that is started HERE, before any
    # test has touched the GPU: a process that has initialised HIP must never fork+exec a new interpreter on this pool, while
    # forking from the (GPU-free) server involves no exec at all.
    import multiprocessing as mp
    try:
        ctx = mp.get_context("forkserver")
        from multiprocessing import forkserver
        forkserver.ensure_running()
    except (ValueError, RuntimeError, OSError):
        pass


@pytest.fixture(scope="session")
def ftx_lib():
    from fusiontransformer_amd import _lib
    return _lib.load()


def pytest_sessionstart(session):
    # the oracle runs on the CPU: use the cores this process really has (cgroup quota), not the host's
    import torch
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (AttributeError, OSError, ValueError):
        pass
    torch.set_num_threads(max(1, min(n, 16)))
