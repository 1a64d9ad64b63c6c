import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def have_gpu():
    try:
        import sift_amd
        return sift_amd.device_count() > 0
    except Exception:
        return False


@pytest.fixture(scope="session")
def hip():
    """The HIP library; GPU tests fail (not skip) if it cannot be loaded."""
    import sift_amd
    L = sift_amd.lib()
    assert L.sift_amd_device_count() > 0, "no GPU visible to libsift_amd.so"
    return L
