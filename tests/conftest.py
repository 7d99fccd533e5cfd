import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


@pytest.fixture(scope="session")
def hs():
    import hsamd

    return hsamd.load()


def have_gpu():
    try:
        import hsamd

        L = hsamd.load()._lib.lib()
        return L.hs_device_info(None, 0, None, None) > 0
    except Exception:
        return False
