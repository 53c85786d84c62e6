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


@pytest.mark.parametrize("S", [4, 7, 64, 333])
def test_ess_and_rhat_kernels_on_degenerate_series_bit_exact(hip, S):
    """The same kernels on series a sampler can really produce when it is stuck or broken: constant series
    (variance 0: diagnostics.ex's guards), two-valued series, a single outlier of 1e300, infinities, a NaN in the
    middle, signed zeros, denormal scales -- the checker's bits, NaN for NaN. (An Erlang float holds neither NaN nor
    an infinity -- the reference's Enum code would have raised before reaching them -- so for those entries this is the
    two restatements agreeing with each other, not with the reference.)"""
    comp = sampler.compile(models.eight_schools())
    rng = np.random.default_rng(900 + S)
    D, Cn = 4, 12
    x = rng.normal(size=(S, D, Cn))
    x[:, 0, 0] = 3.25                                    # constant
    x[:, 0, 1] = np.where(np.arange(S) % 2 == 0, 1.0, -1.0)   # perfectly anti-correlated
    x[:, 0, 2] = np.where(np.arange(S) < S // 2, 0.0, 1.0)    # a step: the split halves disagree
    x[S // 2, 0, 3] = 1e300                              # one outlier
    x[:, 0, 4] *= 1e-310                                 # denormal scale
    x[:, 0, 5] = np.where(np.arange(S) % 3 == 0, 0.0, -0.0)   # signed zeros only
    x[S // 3, 1, 0] = np.inf
    x[S // 3, 1, 1] = -np.inf
    x[S // 2, 1, 2] = np.nan
    x[0, 1, 3] = np.nan                                  # a NaN first
    x[S - 1, 1, 4] = np.nan                              # a NaN last
    x[:, 2, :] = 7.0                                     # a whole dimension constant across chains
    x[:, 3, :] = np.arange(Cn)[None, :]                  # constant within chains, different between them
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
        chains = np.ascontiguousarray(x[:, dim, :].T)
        want = L.exo_rhat(O.dptr(chains), Cn, S)
        assert want == rhat[dim] or (np.isnan(want) and np.isnan(rhat[dim])), (dim, want, rhat[dim])
        for c in range(Cn):
            w = L.exo_ess(O.dptr(np.ascontiguousarray(chains[c])), S)
            assert w == ess[dim, c] or (np.isnan(w) and np.isnan(ess[dim, c])), (dim, c, w, ess[dim, c])
