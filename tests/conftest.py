import os
import sys
import time

if os.path.exists("/dev/kfd"):
    # GPU box: let the HIP runtime name a queue error / fault in the normal run's log (it prints its abort reason only at
    # log level >= 1).  Must be in the environment before libamdhip64 is loaded, i.e. before `import torch`.
    os.environ.setdefault("AMD_LOG_LEVEL", "1")

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


_PROGRESS = None


def pytest_sessionstart(session):
    """GPU runs keep a crash trail that a later run never overwrites (the log of the one aborted run was lost):
    gpurun_out/crash/<time>_<pid>.txt gets one flushed line per test START, so after an abort from a runtime thread the file
    names the test that was running; faulthandler (pytest enables it: Python stacks of all threads on SIGSEGV / SIGABRT) goes
    to stderr, which tools/gpu_suite.sh redirects into a log with the same unique stem."""
    global _PROGRESS
    if not torch.cuda.is_available():
        return
    d = os.path.join(REPO, "gpurun_out", "crash")
    try:
        os.makedirs(d, exist_ok=True)
        _PROGRESS = open(os.path.join(d, time.strftime("%Y%m%d_%H%M%S") + f"_{os.getpid()}.txt"), "w")
        _PROGRESS.write(f"AMD_LOG_LEVEL={os.environ.get('AMD_LOG_LEVEL')} torch={torch.__version__} argv={sys.argv}\n")
        _PROGRESS.flush()
    except OSError:
        _PROGRESS = None


def pytest_runtest_logstart(nodeid, location):
    if _PROGRESS is not None:
        _PROGRESS.write(f"{time.strftime('%H:%M:%S')} START {nodeid}\n")
        _PROGRESS.flush()
        os.fsync(_PROGRESS.fileno())


def pytest_sessionfinish(session, exitstatus):
    if _PROGRESS is not None:
        _PROGRESS.write(f"{time.strftime('%H:%M:%S')} SESSION FINISHED exit={exitstatus}\n")
        _PROGRESS.close()


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    return load


def rel_err(a, b):
    """(max-abs, rms) of a-b, both relative to the std of b."""
    a = torch.as_tensor(a, dtype=torch.float64).reshape(-1)
    b = torch.as_tensor(b, dtype=torch.float64).reshape(-1)
    s = float(b.std()) + 1e-30
    d = a - b
    return float(d.abs().max()) / s, float(d.square().mean().sqrt()) / s
