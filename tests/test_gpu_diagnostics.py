"""Exmc.Diagnostics on the device (lib/exmc/diagnostics.ex:42-52, 80-115, 123-167): ESS of every
(dim, chain) series and split R-hat per dimension of a device-resident trace [draw][dim][chain],
bit for bit against the checker's restatement (which is pinned by the reference's closed forms in
tests/test_golden_reference.py / test_oracle_sampler.py)."""
import numpy as np
import pytest
import torch

import oracle as O
from exmc_amd import _lib, models, sampler

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("S,D,Cn", [(60, 3, 5), (101, 2, 300), (8, 1, 2), (250, 10, 64), (5000, 2, 70)])
def test_ess_and_rhat_kernels_bit_exact(hip, S, D, Cn):
    comp = sampler.compile(models.eight_schools())
    rng = np.random.default_rng(S + D)
    # AR(1) series with chain-dependent correlation and a small chain-dependent offset
    x = np.zeros((S, D, Cn))
    rho = rng.uniform(0.0, 0.95, size=(D, Cn))
    e = rng.normal(size=(S, D, Cn))
    x[0] = e[0]
    for i in range(1, S):
        x[i] = rho * x[i - 1] + e[i]
    x += 0.2 * rng.normal(size=(1, D, Cn))
    dev = torch.device("cuda:0")
    xd = torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    ess = torch.empty((D, Cn), dtype=torch.float64, device=dev)
    rhat = torch.empty((D,), dtype=torch.float64, device=dev)
    _lib.check(hip.exmc_hip_ess(comp.h, xd.data_ptr(), S, D, Cn, ess.data_ptr()))
    _lib.check(hip.exmc_hip_rhat(comp.h, xd.data_ptr(), S, D, Cn, rhat.data_ptr()))
    torch.cuda.synchronize()
    ess, rhat = ess.cpu().numpy(), rhat.cpu().numpy()
    L = O.lib()
    for dim in range(D):
        chains = np.ascontiguousarray(x[:, dim, :].T)          # [C][S]
        assert L.exo_rhat(O.dptr(chains), Cn, S) == rhat[dim], dim
        for c in range(min(Cn, 40)):
            assert L.exo_ess(O.dptr(np.ascontiguousarray(chains[c])), S) == ess[dim, c], (dim, c)
    if S >= 4:
        assert np.all(rhat > 0.9) and np.all(ess > 0)
