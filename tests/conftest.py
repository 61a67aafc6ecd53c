import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` through gpurun)")
    config.addinivalue_line("markers", "sanitize: CPU-only ASan / UBSan leg of the host C-ABI (tests/test_sanitize_host.py)")


def _have_gpu() -> bool:
    try:
        import ctypes as C
        from cova_amd import _lib as L
        n = C.c_int(0)
        return L.lib().covahip_device_count(C.byref(n)) == 0 and n.value > 0
    except Exception:
        return False


@pytest.fixture(scope="session")
def ctx():
    from cova_amd.elements import Context
    if not _have_gpu():
        pytest.skip("no HIP device")
    c = Context(0)
    yield c
    c.close()


@pytest.fixture(scope="session")
def weights_flat():
    from cova_amd import weights as W
    return W.random_init(1234)
