"""The hand-written kinds evaluate their models in a fast window (short divisions, the main paths of exp /
log, sticky domain flags decided per pass: exmc_device.hpp with_fast_div, exmc_models.hpp) and again exactly
when a wavefront leaves it. tests/test_gpu_parity.py compares them with the checker on moderate positions and
on the clamp edges; here the positions are chosen to LEAVE the window -- NaN, infinities, 1e308, denormals,
+-705 / +-745, zeros -- in three chains of every four, so that a wavefront holds chains that need the exact
form next to chains that do not (one chain per wavefront at 64 lanes: whole waves of either kind), for every
lane layout of every kind. Bit for bit against the checker, NaN for NaN."""
import ctypes as C

import numpy as np
import pytest

import oracle as O
from exmc_amd import _lib, models, sampler

pytestmark = pytest.mark.gpu

SPECIALS = [np.nan, np.inf, -np.inf, 1e308, -1e308, 1e5, -1e5, 705.0, -705.0, 745.2, -745.2, 5e-324, -5e-324,
            2.2250738585072014e-308, 1e-310, 0.0, -0.0, 199.99999, -199.99999, 200.0, -200.0, 200.00001,
            -200.00001, 38.0, -38.0, 1e-200, -1e-200]


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _kinds():
    import test_golden_traces as TG
    return [("simple", models.simple, [1]),
            ("eight_schools", models.eight_schools, [1, 2, 4, 8, 16]),
            ("sv", lambda: models.sv(TG.GOLD["sv_returns"]), [32, 64]),
            ("logistic", models.logistic, [4, 8, 16]),
            ("radon", models.radon, [32, 64])]


@pytest.mark.parametrize("name,factory,lane_list", _kinds(), ids=lambda x: x if isinstance(x, str) else "")
def test_positions_that_leave_the_window_bit_exact(hip, name, factory, lane_list):
    spec = factory()
    comp = sampler.compile(spec)
    om = O.model_for(spec)
    rng = np.random.default_rng(29)
    n = 384
    q0 = spec.to_unconstrained(spec.default_init)
    q = np.ascontiguousarray(q0[None, :] + rng.normal(size=(n, spec.d)) * 0.4)
    for c in range(n):
        if c % 4 == 0:
            continue
        for i in rng.choice(spec.d, size=min(int(rng.integers(1, 4)), spec.d), replace=False):
            q[c, i] = SPECIALS[int(rng.integers(len(SPECIALS)))]
    for lanes in lane_list:
        lp = np.full(n, 7.0)
        g = np.full((n, spec.d), 7.0)
        _lib.check(hip.exmc_hip_logp_grad_host(comp.h, _dp(q), n, lanes, _dp(lp), _dp(g)))
        cfg = O.Cfg(1, lanes)
        n_nonfinite = 0
        for c in range(n):
            olp, og = om.logp_grad(q[c], cfg)
            assert olp == lp[c] or (np.isnan(olp) and np.isnan(lp[c])), (name, lanes, c, olp, lp[c], q[c])
            assert np.array_equal(og, g[c], equal_nan=True), (name, lanes, c, q[c], og, g[c])
            n_nonfinite += not (np.isfinite(olp) and np.all(np.isfinite(og)))
        assert 0 < n_nonfinite < n, (name, lanes, n_nonfinite)
