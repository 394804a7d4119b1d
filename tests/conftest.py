import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _torch_cuda_first():
    """torch refuses to initialise its HIP context after another library has touched the runtime in-process: whatever order
    the GPU tests run in, torch (when a GPU is there) goes first."""
    try:
        import torch
        if torch.cuda.is_available():
            torch.zeros(1, device="cuda:0")
    except Exception:
        pass
    yield


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure), built on demand from oracle/msfl_oracle.c."""
    from oracle import oracle as orc
    orc.build()
    orc.lib()
    return orc


@pytest.fixture(scope="session")
def gpu():
    """A handle on cuda:0.  Fails loudly when the HIP extension or the GPU is missing."""
    from msf_loam_amd import capi
    try:
        # initialise torch's HIP context first: tests that share device buffers with torch need it,
        # and torch refuses to initialise after another library has touched the runtime in-process
        import torch
        if torch.cuda.is_available():
            torch.zeros(1, device="cuda:0")
    except Exception:
        pass
    h = capi.Handle(0)
    yield h
    h.close()
