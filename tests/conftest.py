import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure only)."""
    from oracle import oracle as O
    O.lib()
    return O


@pytest.fixture(scope="session")
def gpu64():
    """A double-precision context sized for the small parity images."""
    import sift_amd
    ctx = sift_amd.SiftGPU(max_width=640, max_height=480, max_batch=4, precision=sift_amd.SG_F64, keep_prequant=True)
    yield ctx
    ctx.close()


@pytest.fixture(scope="session")
def gpu32():
    import sift_amd
    ctx = sift_amd.SiftGPU(max_width=640, max_height=480, max_batch=4, precision=sift_amd.SG_F32, keep_prequant=True)
    yield ctx
    ctx.close()
