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


# Development aid (tools/prebuild_plugins.sh): EXMC_PREBUILD_PLUGINS=1 python -m pytest tests -m gpu
# on a box WITHOUT a GPU builds every plug-in library the GPU tests will ask for (so that they
# travel to the GPU box ready-made) and skips each test at the point where it would open the device.
if os.environ.get("EXMC_PREBUILD_PLUGINS") == "1":
    from exmc_amd import sampler as _sampler

    def _no_device(*_a, **_k):
        pytest.skip("plug-in prebuilt; no device on this box")
    _sampler.compile = _no_device


@pytest.fixture(scope="session")
def hip():
    """The HIP path. Fails loudly (never skips) when marked gpu and the device is missing."""
    from exmc_amd import _lib
    L = _lib.load()
    if os.environ.get("EXMC_PREBUILD_PLUGINS") == "1":
        return L
    if L.exmc_hip_device_count() <= 0:
        pytest.fail("no HIP device: the -m gpu tests must run on the MI355X box")
    return L
