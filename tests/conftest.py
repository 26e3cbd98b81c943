import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _have_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture(scope="session")
def ctx():
    """A ccm context on GPU 0.  GPU tests fail (not skip) when the library is missing.
    torch is initialised FIRST where a GPU is present: a few tests use torch tensors as device buffers (as bench.py does), and
    torch's bundled HIP runtime refuses to come up ("No HIP GPUs are available") once libccm_hot.so has initialised the
    system one in the same process; in the other order both share torch's copy."""
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except ImportError:
        pass
    from motioncheck_ccm_slam_amd import _lib
    return _lib.default_context(0)


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle_py
    oracle_py.lib()
    return oracle_py
