import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _native_built():
    """The native pieces are built in-tree before any test touches them: liburf_hip.so (hipcc
    cross-compiles without a GPU) and the oracles (test infrastructure)."""
    from urban_road_filter_amd import build as b
    b.build(verbose=False)   # a no-op when the library is newer than its sources
    import oracles
    oracles.ensure_built()


def gpu_available():
    try:
        import ctypes
        n = ctypes.c_int(0)
        hip = ctypes.CDLL("libamdhip64.so")
        return hip.hipGetDeviceCount(ctypes.byref(n)) == 0 and n.value > 0
    except Exception:
        return False
