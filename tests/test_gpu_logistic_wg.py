"""The workgroup form of logistic's sampling kernel (exmc_nuts.hpp nuts_kernel_wg: eight wavefronts
around one LDS image of the design matrix) against the checker, on the shapes the
full-size protocol does not reach: observation counts that are not a multiple of sixteen (a partly
filled last step, `live` masks), fewer than sixteen observations (one partial step), exactly 512 (the
image's capacity), chain counts that leave wavefronts and lane groups of a workgroup empty, several
workgroups, predictors outside [-200, 200] (the per-step fallback to the general exp / log / division),
and more observations than the image holds (the library must take the one-wave form). EXMC_HIP_NUTS_WG
forces the form; both forms and the dispatcher's own choice must give the checker's bits."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

pytestmark = pytest.mark.gpu

import oracle as O  # noqa: E402
from exmc_amd import models, sampler  # noqa: E402

KEYS = ("draws", "logp", "tree_depth", "n_steps", "divergent", "accept_prob", "energy")


def _run(spec, n_chains, n_draws, eps, seed, init=None, inv_mass=None):
    comp = sampler.compile(spec)
    try:
        d = spec.d
        im = np.ones(d) if inv_mass is None else inv_mass
        tuning = dict(epsilon=eps, inv_mass=im, chol_cov=None)
        opts = dict(num_warmup=0, num_samples=n_draws, seed=seed, lanes_per_chain=16, max_tree_depth=6)
        _, _, extra = sampler.sample_compiled_tuned(comp, tuning, init if init is not None else spec.default_init, opts,
                                                    num_chains=n_chains)
        return extra["raw"], extra["total_leapfrogs"]
    finally:
        comp.close()


def _oracle(spec, n_chains, n_draws, eps, seed, init=None, inv_mass=None):
    om = O.model_for(spec)
    im = np.ones(spec.d) if inv_mass is None else inv_mass
    q0 = spec.to_unconstrained(init if init is not None else spec.default_init)
    out = {k: [] for k in KEYS}
    for c in range(n_chains):
        t, _ = O.sample_tuned(om, eps, im, q0, num_samples=n_draws, max_tree_depth=6, seed=seed + 7919 * c,
                              cfg=O.Cfg(1, 16))
        for k in KEYS:
            out[k].append(t[k])
    return {k: np.stack(v) for k, v in out.items()}


@pytest.mark.parametrize("n_obs,n_chains", [(500, 37), (37, 5), (7, 3), (512, 70), (129, 33), (16, 4)])
def test_workgroup_form_equals_the_checker(monkeypatch, n_obs, n_chains):
    X, y = models.logistic_data(seed=100 + n_obs, n=n_obs, k=20)
    spec = models.logistic(X, y)
    want = _oracle(spec, n_chains, 12, 0.2, 9)
    for form in ("1", "0", None):
        if form is None:
            monkeypatch.delenv("EXMC_HIP_NUTS_WG", raising=False)
        else:
            monkeypatch.setenv("EXMC_HIP_NUTS_WG", form)
        got, _ = _run(spec, n_chains, 12, 0.2, 9)
        for k in KEYS:
            assert np.array_equal(want[k], got[k], equal_nan=True), (form, n_obs, n_chains, k)


def test_predictors_outside_the_short_forms_domain(monkeypatch):
    """Coefficients of +-30 against features of size ~3 put most linear predictors beyond +-200 in some
    steps and inside in others: the per-step choice between the short and the general forms must not
    show in any bit (clipped probabilities, zero residuals, log of the clip bounds)."""
    rng = np.random.default_rng(4)
    X = rng.normal(size=(200, 20)) * 3.0
    y = (rng.uniform(size=200) < 0.5).astype(np.float64)
    spec = models.logistic(X, y)
    init = {nm: float(v) for nm, v in zip(spec.var_names, rng.normal(size=21) * 30.0)}
    want = _oracle(spec, 9, 6, 0.01, 3, init=init)
    for form in ("1", "0"):
        monkeypatch.setenv("EXMC_HIP_NUTS_WG", form)
        got, _ = _run(spec, 9, 6, 0.01, 3, init=init)
        for k in KEYS:
            assert np.array_equal(want[k], got[k], equal_nan=True), (form, k)


def test_more_observations_than_the_image_holds(monkeypatch):
    """600 observations do not fit the 512-row image: the workgroup form is not taken even when forced,
    and the one-wave form streams the rows from L2 as before."""
    X, y = models.logistic_data(seed=77, n=600, k=20)
    spec = models.logistic(X, y)
    want = _oracle(spec, 6, 8, 0.15, 2)
    monkeypatch.setenv("EXMC_HIP_NUTS_WG", "1")
    got, _ = _run(spec, 6, 8, 0.15, 2)
    for k in KEYS:
        assert np.array_equal(want[k], got[k], equal_nan=True), k
