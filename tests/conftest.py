import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

GOLDEN = ROOT / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # Mutation runs (dev, tools/r05/build_mutants.py): MEDTOK_TEST_LIB=<path> binds a VARIANT BUILD of the library for this test
    # session -- to record which tests catch a re-introduced bug.  Never set in the driver's runs; the product reads no environment.
    lib = os.environ.get("MEDTOK_TEST_LIB")
    if lib:
        from medtok_amd import _lib
        _lib.use_library(lib)


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(GOLDEN / f"{name}.npz", allow_pickle=False)
    return load


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    return torch.device("cuda:0")


def run_with_retry(make_cmd, timeout=420, **kw):
    """subprocess.run for the multi-process launches of the GPU tests (torch.distributed.run / bench.py --gpus 2, two ranks over gloo
    on this box's one GPU): a legitimate run takes seconds, so a launch that has not returned after `timeout` seconds is a stuck
    rendezvous of the environment (seen once: one box, one test, its whole 900-second limit) and is tried ONCE more on a fresh port.
    make_cmd(port) -> argv.  Product-side, medtok_amd.distributed gives gloo a 300-second timeout of its own."""
    import socket
    import subprocess
    last = None
    for attempt in range(2):
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
        try:
            return subprocess.run(make_cmd(port), capture_output=True, text=True, timeout=timeout, **kw)
        except subprocess.TimeoutExpired as exc:
            last = exc
    raise last
