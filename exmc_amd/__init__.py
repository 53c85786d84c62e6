"""exmc_amd — MI355X-native NUTS inner loop behind eXMC's sampler API.

Package layout: csrc/ (HIP kernels + the C ABI of libexmc_hip.so), sampler.py (host mirror of
Exmc.NUTS.Sampler), models.py (model kinds of the BASELINE configs), build.py (hipcc driver).
If torch is going to be used in the same process it must load its HIP runtime first, so it is
imported here before libexmc_hip.so whenever it is installed.
"""
try:  # plumbing only: device memory, streams, torch.distributed (RCCL)
    import torch  # noqa: F401
except Exception:  # pragma: no cover
    torch = None

from . import _lib, models, sampler  # noqa: E402,F401
from ._lib import ExmcHipError  # noqa: E402,F401

__all__ = ["models", "sampler", "ExmcHipError"]
