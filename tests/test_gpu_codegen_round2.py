"""GPU parity for the generator coverage of round 2 (vector free RVs: GaussianRandomWalk and
MvNormal, a Custom-distribution closure, meas_obs; tests/gen_models.py::walk_ir): the plug-in HIP
library through the C ABI against the CPU checker running the same generated text -- log-density and
gradient on 200 points, the sample/3 path (warmup + draws) and a batch of chains, bit for bit; the
vector entries come back as one array per rv in the trace."""
import ctypes as C

import numpy as np
import pytest

import gen_checker as GC
import gen_models as GM
import oracle as O
from exmc_amd import codegen as cg, sampler

pytestmark = pytest.mark.gpu


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


@pytest.fixture(scope="module")
def walk(hip):
    spec = cg.compile_ir(GM.walk_ir(), name="gen_walk", default_init=GM.WALK_INIT)
    assert spec.gen.lanes == 1 and spec.d == 12
    return spec, sampler.compile(spec), GC.model(spec.gen, 1)


def test_logp_grad_bit_exact(walk):
    spec, comp, om = walk
    rng = np.random.default_rng(5)
    n = 200
    q = np.ascontiguousarray(rng.normal(size=(n, spec.d)) * 1.1)
    q[0] = spec.to_unconstrained(spec.default_init)
    q[1, :] = 250.0
    q[2, :] = -250.0
    q[3, :] = 0.0
    lp, g = np.zeros(n), np.zeros((n, spec.d))
    comp.check(comp.L.exmc_hip_logp_grad_host(comp.h, _dp(q), n, 1, _dp(lp), _dp(g)))
    for i in range(n):
        lpo, go = om.logp_grad(q[i], O.Cfg(1, 1))
        assert (lp[i] == lpo) or (np.isnan(lp[i]) and np.isnan(lpo)), i
        assert np.array_equal(g[i], go, equal_nan=True), i


def test_sample_and_chains_bit_exact(walk):
    spec, comp, om = walk
    q0 = spec.to_unconstrained(spec.default_init)
    trace, stats = sampler.sample(spec, spec.default_init, dict(num_warmup=120, num_samples=60, seed=4))
    t, st = O.sample(om, q0, num_warmup=120, num_samples=60, seed=4, cfg=O.Cfg(1, 1))
    assert st.step_size == stats["step_size"]
    assert np.array_equal(t["draws"], stats["raw"]["draws"][0])
    assert np.array_equal(t["tree_depth"], stats["raw"]["tree_depth"][0])
    assert trace["w"].shape == (60, 6) and trace["m"].shape == (60, 3) and np.all(trace["sigma"] > 0)
    assert np.array_equal(trace["w"][:, 5], trace["w[5]"])
    opts = dict(num_warmup=120, num_samples=25, seed=4)
    tuning = sampler.warmup(comp, spec.default_init, opts)
    _, _, extra = sampler.sample_compiled_tuned(comp, tuning, spec.default_init, opts, num_chains=70)
    tc, _ = O.sample_chains(om, 70, init_q=q0, num_warmup=120, num_samples=25, seed=4, n_threads=8,
                            cfg=O.Cfg(1, 1))
    assert np.array_equal(tc["draws"], extra["raw"]["draws"])
    assert np.array_equal(tc["n_steps"], extra["raw"]["n_steps"])


@pytest.fixture(scope="module")
def simplex(hip):
    spec = cg.compile_ir(GM.simplex_ir(), name="gen_simplex", default_init=GM.SIMPLEX_INIT)
    assert spec.gen.lanes == 1 and spec.d == 7
    return spec, sampler.compile(spec), GC.model(spec.gen, 1)


def test_simplex_model_bit_exact(simplex):
    """Dirichlet behind :stick_breaking, Gamma / Beta / Weibull / Uniform01 free rvs, Poisson /
    Weibull / Gamma / Dirichlet observations (tests/gen_models.py::simplex_ir)."""
    spec, comp, om = simplex
    rng = np.random.default_rng(8)
    n = 200
    q = np.ascontiguousarray(rng.normal(size=(n, spec.d)) * 1.2)
    q[0] = spec.to_unconstrained(spec.default_init)
    q[1, :] = 40.0
    q[2, :] = -40.0
    q[3, :] = 0.0
    lp, g = np.zeros(n), np.zeros((n, spec.d))
    comp.check(comp.L.exmc_hip_logp_grad_host(comp.h, _dp(q), n, 1, _dp(lp), _dp(g)))
    for i in range(n):
        lpo, go = om.logp_grad(q[i], O.Cfg(1, 1))
        assert (lp[i] == lpo) or (np.isnan(lp[i]) and np.isnan(lpo)), i
        assert np.array_equal(g[i], go, equal_nan=True), i
    q0 = spec.to_unconstrained(spec.default_init)
    trace, stats = sampler.sample(spec, spec.default_init, dict(num_warmup=150, num_samples=80, seed=6))
    t, st = O.sample(om, q0, num_warmup=150, num_samples=80, seed=6, cfg=O.Cfg(1, 1))
    assert st.step_size == stats["step_size"]
    assert np.array_equal(t["draws"], stats["raw"]["draws"][0])
    assert np.array_equal(t["n_steps"], stats["raw"]["n_steps"][0])
    th = trace["theta"]
    assert th.shape == (80, 4) and np.allclose(th.sum(axis=1), 1.0) and np.all(th > 0)
    assert np.all(trace["rate"] > 0) and np.all((trace["p"] > 0) & (trace["p"] < 1))


@pytest.fixture(scope="module")
def survival(hip):
    spec = cg.compile_ir(GM.survival_ir(), name="gen_survival", default_init=GM.SURVIVAL_INIT)
    assert spec.gen.lanes == 1 and spec.d == 6
    return spec, sampler.compile(spec), GC.model(spec.gen, 1)


def test_obs_meta_model_bit_exact(survival):
    """Builder.obs meta (censored right / left / interval, weight, mask, reduce :mean and
    :logsumexp), an observation of a :log-transformed rv and a Mixture likelihood
    (tests/gen_models.py::survival_ir; values against scipy in tests/test_codegen_obs_meta.py)."""
    spec, comp, om = survival
    rng = np.random.default_rng(9)
    n = 200
    q = np.ascontiguousarray(rng.normal(size=(n, spec.d)) * 1.5)
    q[0] = spec.to_unconstrained(spec.default_init)
    q[1, :] = 30.0
    q[2, :] = -30.0
    q[3, :] = 0.0
    lp, g = np.zeros(n), np.zeros((n, spec.d))
    comp.check(comp.L.exmc_hip_logp_grad_host(comp.h, _dp(q), n, 1, _dp(lp), _dp(g)))
    for i in range(n):
        lpo, go = om.logp_grad(q[i], O.Cfg(1, 1))
        assert (lp[i] == lpo) or (np.isnan(lp[i]) and np.isnan(lpo)), i
        assert np.array_equal(g[i], go, equal_nan=True), i
    q0 = spec.to_unconstrained(spec.default_init)
    trace, stats = sampler.sample(spec, spec.default_init, dict(num_warmup=150, num_samples=80, seed=3))
    t, st = O.sample(om, q0, num_warmup=150, num_samples=80, seed=3, cfg=O.Cfg(1, 1))
    assert st.step_size == stats["step_size"]
    assert np.array_equal(t["draws"], stats["raw"]["draws"][0])
    assert np.array_equal(t["n_steps"], stats["raw"]["n_steps"][0])
    assert np.all(trace["k"] > 0) and np.all(trace["s"] > 0)
    # the two mixture means stay on their own sides (priors at -1 and 2, weights 0.35 / 0.65)
    assert trace["m1"].mean() < trace["m2"].mean()
    opts = dict(num_warmup=100, num_samples=20, seed=12)
    tuning = sampler.warmup(comp, spec.default_init, opts)
    _, _, extra = sampler.sample_compiled_tuned(comp, tuning, spec.default_init, opts, num_chains=96)
    tc, _ = O.sample_chains(om, 96, init_q=q0, num_warmup=100, num_samples=20, seed=12, n_threads=8,
                            cfg=O.Cfg(1, 1))
    assert np.array_equal(tc["draws"], extra["raw"]["draws"])
    assert np.array_equal(tc["n_steps"], extra["raw"]["n_steps"])
