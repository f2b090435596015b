import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def has_gpu() -> bool:
    try:
        from algoplonk_amd import _lib
        return _lib.device_count() > 0
    except Exception:
        return False


@pytest.fixture(scope="session")
def gpu():
    from algoplonk_amd import _lib
    n = _lib.device_count()
    if n == 0:
        pytest.fail("no HIP device visible: the gpu-marked tests must run on the GPU box (libapk has no CPU fallback)")
    return 0
