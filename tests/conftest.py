import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_lib():
    import oracle
    return oracle.lib()


@pytest.fixture(scope="session")
def hip():
    """The HIP path. Fails loudly (never skips) when marked gpu and the device is missing."""
    from exmc_amd import _lib
    L = _lib.load()
    if L.exmc_hip_device_count() <= 0:
        pytest.fail("no HIP device: the -m gpu tests must run on the MI355X box")
    return L
