"""Diagnostics.ess_bulk on the device (diagnostics.ex:60-72, 186-219): rank-normalised ESS of every
(dim, chain) series of a device trace, against the checker bit for bit (deterministic log in the
probit; libm mode within 1e-9)."""
import numpy as np
import pytest
import torch

import oracle as O
from exmc_amd import _lib, models, sampler

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("S,D,Cn", [(1000, 3, 5), (257, 2, 9), (3, 1, 2), (64, 2, 3), (9000, 1, 3)])
def test_ess_bulk_kernel_bit_exact(hip, S, D, Cn):
    comp = sampler.compile(models.eight_schools())
    rng = np.random.default_rng(S * 7 + D)
    x = np.zeros((S, D, Cn))
    rho = rng.uniform(0.0, 0.9, size=(D, Cn))
    e = rng.standard_t(3, size=(S, D, Cn))          # heavy tails: where bulk and plain ESS differ
    x[0] = e[0]
    for i in range(1, S):
        x[i] = rho * x[i - 1] + e[i]
    if S >= 64:
        x[5:9, 0, 0] = x[4, 0, 0]                    # ties -> average ranks
        x[:, D - 1, Cn - 1] = np.round(x[:, D - 1, Cn - 1])   # many ties
    dev = torch.device("cuda:0")
    xd = torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    out = torch.empty((D, Cn), dtype=torch.float64, device=dev)
    _lib.check(hip.exmc_hip_ess_bulk(comp.h, xd.data_ptr(), S, D, Cn, out.data_ptr()))
    torch.cuda.synchronize()
    out = out.cpu().numpy()
    L = O.lib()
    for dim in range(D):
        for c in range(Cn):
            series = np.ascontiguousarray(x[:, dim, c])
            assert L.exo_ess_bulk_mode(O.dptr(series), S, 1) == out[dim, c], (dim, c)
            assert abs(L.exo_ess_bulk(O.dptr(series), S) - out[dim, c]) <= 1e-9 * S
