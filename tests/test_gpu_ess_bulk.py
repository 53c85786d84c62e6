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


@pytest.mark.parametrize("S", [4, 5, 64, 1000, 1024, 1025, 4096, 4097])
def test_ranks_by_sorting_equal_ranks_by_counting(hip, monkeypatch, S):
    """Round 5: the ranks come from a bitonic sort of the series in LDS (rank_scores_sort_kernel) where
    they used to be counted (S comparisons per element). Ranks are integers: both kernels, and the
    checker, must give the same bits -- on ties, a constant series, signed zeros, infinities, a series
    that holds a NaN (ranked by the counting rule), at sizes on both sides of every power of two and of
    the sort's capacity (4096, above it the counting kernel runs)."""
    comp = sampler.compile(models.eight_schools())
    rng = np.random.default_rng(S)
    D, Cn = 2, 6
    x = rng.normal(size=(S, D, Cn))
    x[:, 0, 1] = 3.25                                   # constant
    x[:, 0, 2] = np.round(x[:, 0, 2] * 2.0) / 2.0       # heavy ties
    x[::3, 0, 3] = 0.0
    x[1::3, 0, 3] = -0.0                                # -0.0 == +0.0: one tie group
    x[0, 0, 4] = np.inf
    x[S - 1, 0, 4] = -np.inf
    if S > 4:
        x[2, 0, 4] = np.inf
    x[S // 2, 1, 0] = np.nan                            # a NaN: every comparison with it is false
    dev = torch.device("cuda:0")
    xd = torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    outs = []
    for mode in (None, "0"):
        if mode is None:
            monkeypatch.delenv("EXMC_HIP_RANK_SORT", raising=False)
        else:
            monkeypatch.setenv("EXMC_HIP_RANK_SORT", mode)
        out = torch.empty((D, Cn), dtype=torch.float64, device=dev)
        _lib.check(hip.exmc_hip_ess_bulk(comp.h, xd.data_ptr(), S, D, Cn, out.data_ptr()))
        torch.cuda.synchronize()
        outs.append(out.cpu().numpy())
    assert np.array_equal(outs[0], outs[1], equal_nan=True)
    L = O.lib()
    for dim in range(D):
        for c in range(Cn):
            series = np.ascontiguousarray(x[:, dim, c])
            if np.isnan(series).any():
                continue    # outside the reference's domain (a BEAM float is never NaN): only sort == count, above
            want = L.exo_ess_bulk_mode(O.dptr(series), S, 1)
            got = outs[0][dim, c]
            assert want == got or (np.isnan(want) and np.isnan(got)), (S, dim, c, want, got)
