"""GPU parity for generated models (exmc_amd/codegen.py, codegen_vec.py): the plug-in HIP library,
through the C ABI, against the CPU oracle running the same generated text compiled with gcc
(tests/gen_checker.py). Bit for bit, as for the hand-written kinds, in both layouts: one lane per
chain and plates across 16 lanes."""
import ctypes as C

import numpy as np
import pytest

import gen_checker as GC
import gen_models as GM
import oracle as O
from exmc_amd import codegen as cg, models, sampler

pytestmark = pytest.mark.gpu


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _compiled(ir, init, name):
    spec = cg.compile_ir(ir, name=name, default_init=init)
    assert spec.gen.lanes == 16
    return spec, sampler.compile(spec), {1: GC.model(spec.gen, 1), 16: GC.model(spec.gen, 16)}


@pytest.fixture(scope="module")
def simple(hip):
    return _compiled(GM.simple_ir(), {"mu": 2.0, "sigma": 1.0}, "gen_simple")


@pytest.fixture(scope="module")
def schools(hip):
    init = {n: 0.0 for n in ["mu"] + ["theta_%d" % j for j in range(8)]}
    init["tau"] = 1.0
    return _compiled(GM.eight_schools_ir(), init, "gen_eight_schools")


@pytest.fixture(scope="module")
def zoo(hip):
    return _compiled(GM.zoo_ir(), GM.ZOO_INIT, "gen_zoo")


@pytest.mark.parametrize("lanes", [1, 16])
@pytest.mark.parametrize("which", ["simple", "schools", "zoo"])
def test_generated_logp_grad_bit_exact(which, lanes, request):
    spec, comp, oms = request.getfixturevalue(which)
    assert comp.default_lanes == 16
    rng = np.random.default_rng(11)
    n = 193
    q = np.ascontiguousarray(rng.normal(size=(n, spec.d)) * 1.2)
    q[0] = spec.to_unconstrained(spec.default_init)
    q[1, :] = 250.0      # beyond every clamp
    q[2, :] = -250.0
    q[3, :] = 0.0
    lp = np.zeros(n)
    g = np.zeros((n, spec.d))
    comp.check(comp.L.exmc_hip_logp_grad_host(comp.h, _dp(q), n, lanes, _dp(lp), _dp(g)))
    cfg = O.Cfg(1, lanes)
    for c in range(n):
        olp, og = oms[lanes].logp_grad(q[c], cfg)
        assert olp == lp[c] or (np.isnan(olp) and np.isnan(lp[c])), (which, c, olp, lp[c])
        assert np.array_equal(og, g[c], equal_nan=True), (which, c, og, g[c])


@pytest.mark.parametrize("lanes", [1, 16])
@pytest.mark.parametrize("which", ["simple", "schools", "zoo"])
def test_generated_sample_bit_exact(which, lanes, request):
    """Sampler.sample/3 end to end (warmup adaptation + sampling) on the generated kernels."""
    spec, comp, oms = request.getfixturevalue(which)
    opts = dict(num_warmup=150, num_samples=120, seed=17, lanes_per_chain=lanes)
    trace, stats = sampler.sample_compiled(comp, spec.default_init, opts)
    q0 = spec.to_unconstrained(spec.default_init)
    t, st = O.sample(oms[lanes], init_q=q0, num_warmup=150, num_samples=120, seed=17,
                     cfg=O.Cfg(1, lanes))
    assert stats["step_size"] == st.step_size
    raw = stats["raw"]
    assert np.array_equal(raw["tree_depth"][0], t["tree_depth"])
    assert np.array_equal(raw["n_steps"][0], t["n_steps"])
    assert np.array_equal(raw["draws"][0], t["draws"])
    assert np.array_equal(raw["energy"][0], t["energy"])
    # the constrained trace (Transform.apply + reconstruct_ncp)
    x = spec.constrain(t["draws"])
    for i, name in enumerate(spec.var_names):
        assert np.array_equal(np.asarray(trace[name]), x[:, i])


def test_generated_eight_schools_chains_bit_exact_and_same_posterior(schools, hip):
    """The default (16-lane) layout over a batch of chains; same posterior as the hand-written
    kind."""
    spec, comp, oms = schools
    opts = dict(num_warmup=200, num_samples=200, seed=42, init_values=spec.default_init)
    traces, stats = sampler.sample_chains_compiled(comp, 96, opts)
    q0 = spec.to_unconstrained(spec.default_init)
    t, st = O.sample_chains(oms[16], 96, init_q=q0, num_warmup=200, num_samples=200, seed=42,
                            cfg=O.Cfg(1, 16), n_threads=8)
    raw = stats[0]["extra"]["raw"]
    assert stats[0]["step_size"] == st.step_size
    assert np.array_equal(raw["n_steps"], t["n_steps"])
    assert np.array_equal(raw["draws"], t["draws"])
    hs = models.eight_schools()
    hcomp = sampler.compile(hs)
    htr, hst = sampler.sample_chains_compiled(hcomp, 96, dict(num_warmup=200, num_samples=200, seed=43,
                                                              init_values=hs.default_init))
    hraw = hst[0]["extra"]["raw"]["draws"]
    xg = spec.constrain(raw["draws"]).reshape(-1, 10)
    hx = hs.constrain(hraw).reshape(-1, 10)
    h_theta0 = hx[:, 0] + hx[:, 1] * hx[:, 2]
    assert abs(xg[:, 0].mean() - hx[:, 0].mean()) < 0.25
    assert abs(np.median(xg[:, 1]) - np.median(hx[:, 1])) < 0.3
    assert abs(xg[:, 2].mean() - h_theta0.mean()) < 0.35


def test_generated_eight_schools_at_the_bench_size(schools, hip):
    """4096 chains x 1000 draws on the generated 16-lane kernels (what bench.py --model
    gen_eight_schools times): chains from the first, a middle and the last wavefront equal the
    checker's on every output; the counters are the trace's own sums."""
    spec, comp, oms = schools
    n_chains, n_draws = 4096, 1000
    opts = dict(num_warmup=1000, num_samples=n_draws, seed=42)
    tuning = sampler.warmup(comp, spec.default_init, opts)
    _, _, extra = sampler.sample_compiled_tuned(comp, tuning, spec.default_init, opts,
                                                num_chains=n_chains)
    raw = extra["raw"]
    q0 = spec.to_unconstrained(spec.default_init)
    for lo in (0, 2050, n_chains - 2):
        t, st = O.sample_chains(oms[16], n_chains, init_q=q0, num_warmup=1000, num_samples=n_draws,
                                seed=42, chain_lo=lo, chain_hi=lo + 2, cfg=O.Cfg(1, 16))
        assert st.step_size == tuning["epsilon"]
        for k in ("draws", "logp", "tree_depth", "n_steps", "divergent", "accept_prob", "energy"):
            assert np.array_equal(t[k], raw[k][lo:lo + 2]), (k, lo)
    assert extra["total_leapfrogs"] == int(raw["n_steps"].astype(np.int64).sum())


def test_plugin_refuses_other_kinds_and_wrong_data(simple):
    spec, comp, _ = simple
    h = C.c_void_p()
    es = models.eight_schools()
    rc = comp.L.exmc_hip_model_create(es.kind, es.d, _dp(es.data), int(es.data.size), 0, C.byref(h))
    assert rc != 0
    bad = np.zeros(spec.data.size + 1)
    rc = comp.L.exmc_hip_model_create(cg.CUSTOM, spec.d, _dp(bad), int(bad.size), 0, C.byref(h))
    assert rc != 0
