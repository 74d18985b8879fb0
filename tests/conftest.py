import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # Multi-process GPU tests (tests/test_sharded_gpu.py) start their ranks from multiprocessing's fork server.  It has to
    # exist BEFORE this process initialises HIP: a process that holds the GPU must not fork+exec on the GPU boxes, and the
    # server (started here, by one exec from a still GPU-free pytest) forks every later child from its own clean image.
    import multiprocessing as mp
    from multiprocessing import forkserver
    try:
        mp.set_forkserver_preload([])
        forkserver.ensure_running()
    except Exception as exc:   # the CPU suite does not need it
        print(f"[conftest] fork server not started: {exc}")


def golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


@pytest.fixture(scope="session")
def golden_loader():
    return golden
