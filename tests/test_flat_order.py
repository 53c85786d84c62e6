"""The RNG-consuming steps draw in the reference's flat order (PointMap.build sorts the free-RV ids
as strings, lib/exmc/point_map.ex:30-60; init_position sampler.ex:339-349 and sample_momentum_fast
sampler.ex:393-403 fill that vector front to back), whatever the kernels' compute layout is."""
import ctypes as C

import numpy as np
import pytest

import oracle as O
from exmc_amd import models


def sv_returns(seed=42, T=100):
    rng = np.random.default_rng(seed)
    s = np.cumsum(rng.normal(0, 0.15, T))
    return np.exp(s) * rng.standard_t(10.0, T)


def specs():
    return [models.eight_schools(), models.simple(), models.sv(sv_returns()), models.logistic(),
            models.radon()]


@pytest.mark.parametrize("spec", specs(), ids=lambda s: s.name)
def test_flat_order_is_the_string_sort_of_the_ids(spec):
    order = spec.flat_order()
    names = [spec.var_names[i] for i in order]
    assert names == sorted(spec.var_names)          # Enum.sort_by(& &1.id) on binaries
    assert sorted(order) == list(range(spec.d))


def test_known_orders():
    sv = models.sv(sv_returns())
    names = [sv.var_names[i] for i in sv.flat_order()]
    assert names[:5] == ["nu", "s_1", "s_10", "s_100", "s_11"] and names[-1] == "sigma"
    lg = models.logistic()
    names = [lg.var_names[i] for i in lg.flat_order()]
    assert names[:4] == ["alpha", "beta_1", "beta_10", "beta_11"] and names[-1] == "beta_9"
    for spec in (models.eight_schools(), models.simple()):
        assert spec.flat_order() == list(range(spec.d))   # kernel order is sorted already


@pytest.mark.parametrize("spec", [models.sv(sv_returns()), models.logistic()], ids=lambda s: s.name)
def test_checker_defaults_equal_the_spec(spec):
    """The kinds whose names the kind fixes get the sorted order without being told."""
    m = O.Model(spec.kind, spec.d, spec.data)
    assert m.flat_order() == spec.flat_order()


def test_checker_rejects_a_non_permutation():
    m = O.model_for(models.radon())
    bad = list(range(m.d))
    bad[3] = 4
    with pytest.raises(ValueError):
        m.set_flat_order(bad)


@pytest.mark.parametrize("spec", specs(), ids=lambda s: s.name)
def test_random_init_lands_in_flat_order(spec):
    """With no init values the position is 0.1 * normal_s per flat entry: variate r must land on
    kernel dimension flat_order[r] (for sv: the first variate on nu)."""
    m = O.model_for(spec)
    seed = 42
    r = O.Rng()
    O.lib().exo_rng_seed(C.byref(r), seed)
    z = np.array([O.lib().exo_rng_normal(C.byref(r), 1) for _ in range(spec.d)])
    t, _ = O.sample_tuned(m, 1.0e-3, np.ones(spec.d), None, num_samples=1, max_tree_depth=1,
                          seed=seed, cfg=O.Cfg(1, 1))
    # one tiny-step transition barely moves q: recover the init from the draw
    q1 = t["draws"][0]
    order = spec.flat_order()
    expect = np.zeros(spec.d)
    expect[order] = 0.1 * z
    assert np.allclose(q1, expect, atol=5e-2 * (1 + np.abs(expect)))
    if spec.name == "sv":
        assert order[0] == spec.var_names.index("nu")


@pytest.mark.parametrize("spec", [models.sv(sv_returns()), models.logistic(), models.radon()],
                         ids=lambda s: s.name)
def test_momentum_lands_in_flat_order(spec):
    """One transition of one short leapfrog: the accepted move is dq = +-eps * M^-1 * (p + h g),
    so dq / (eps M^-1) shows the momentum p_i = z_rank(i) / sqrt(M^-1_i) dimension by dimension."""
    m = O.model_for(spec)
    d = spec.d
    order = spec.flat_order()
    q0 = spec.to_unconstrained(spec.default_init)
    im = 0.5 + np.arange(d) / d
    eps = 1.0e-4
    _, g0 = m.logp_grad(q0)
    for seed in range(7, 40):
        t, _ = O.sample_tuned(m, eps, im, q0, num_samples=1, max_tree_depth=1, seed=seed,
                              cfg=O.Cfg(0, 1))
        dq = t["draws"][0] - q0
        if np.any(dq != 0.0):
            break
    else:
        pytest.fail("no accepted move in 33 seeds")
    r = O.Rng()
    O.lib().exo_rng_seed(C.byref(r), seed)
    z = np.array([O.lib().exo_rng_normal(C.byref(r), 0) for _ in range(d)])
    p = np.zeros(d)
    p[order] = z
    p = p / np.sqrt(im)
    seen = dq / (eps * im)
    sign = 1.0 if np.sum(np.abs(seen - p)) < np.sum(np.abs(seen + p)) else -1.0
    tol = 1e-6 + eps * np.abs(g0)          # the half-step gradient kick, h * g
    assert np.all(np.abs(sign * seen - p) <= tol * 1.01 + 1e-7 * np.abs(p))
    # the kernel-order assignment is a different vector
    assert np.max(np.abs(sign * seen - z / np.sqrt(im))) > 0.1
