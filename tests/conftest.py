import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """The multi-process tests (several ranks sharing the box's one GPU, subprocesses of bench.py) run LAST: on a freshly
    booted box the first HIP initialisation pages the runtime in, and three processes doing that at once have been seen to
    take minutes (round 3: the first such test sat silent for 240 s on a cold box and took 5.6 s on a warm one)."""
    late = [it for it in items if "test_gpu_comm" in it.nodeid]
    if late:
        items[:] = [it for it in items if "test_gpu_comm" not in it.nodeid] + late


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
    # warm the device up in this process before any test spawns ranks: runtime initialisation, code-object load, one tiny proof
    import __graft_entry__
    __graft_entry__.smoke()
    return 0
