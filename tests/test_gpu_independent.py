"""sample_chains(ir, n, vectorized: false) -- sample_chains_parallel, sampler.ex:992-1000, 1139-1176:
chain i is Sampler.sample/3 with seed base + 7919 i, its own adaptation and then its draws. On the
GPU all chains are ONE launch (exmc_nuts.hpp indep_kernel: every lane group a chain from its first
warmup transition to its last draw). Oracle: the checker's sample(seed = base + 7919 i), chain by
chain -- every per-draw output, each chain's step size, inverse mass and warmup divergences, bit
for bit."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

pytestmark = pytest.mark.gpu

import oracle as O  # noqa: E402
import test_golden_traces as TG  # noqa: E402
from exmc_amd import models, sampler  # noqa: E402


def _check(spec, n_chains, lanes, num_warmup, num_samples, seed, init_values=None, chain_lo=0, chain_hi=None):
    comp = sampler.compile(spec)
    try:
        opts = dict(num_warmup=num_warmup, num_samples=num_samples, seed=seed, lanes_per_chain=lanes,
                    vectorized=False, init_values=init_values or {})
        traces, stats = sampler.sample_chains_independent_compiled(comp, n_chains, opts, chain_lo=chain_lo,
                                                                    chain_hi=chain_hi)
    finally:
        comp.close()
    hi = n_chains if chain_hi is None else chain_hi
    assert len(traces) == hi - chain_lo
    om = O.model_for(spec)
    q0 = None if not init_values else spec.to_unconstrained(init_values)
    raw = stats[0]["extra"]["raw"]
    eps = set()
    for k, c in enumerate(range(chain_lo, hi)):
        t, st = O.sample(om, init_q=q0, num_warmup=num_warmup, num_samples=num_samples, seed=seed + 7919 * c,
                         cfg=O.Cfg(1, lanes))
        assert stats[k]["step_size"] == st.step_size, (c, stats[k]["step_size"], st.step_size)
        assert np.array_equal(stats[k]["inv_mass_diag"], np.array(st.inv_mass[:spec.d])), c
        assert stats[k]["divergences"] == st.divergences, c
        assert np.array_equal(raw["draws"][k], t["draws"]), c
        for key in ("tree_depth", "n_steps", "divergent"):
            assert np.array_equal(raw[key][k], t[key]), (c, key)
        for key in ("logp", "accept_prob", "energy"):
            assert np.array_equal(raw[key][k], t[key]), (c, key)
        eps.add(st.step_size)
    return eps, stats


def test_eight_schools_64_chains_adapt_independently():
    """64 chains x (300 + 200) at 16 lanes per chain: four chains per wavefront adapt side by side
    (their step sizes, window moments and tree depths differ) in lock step."""
    eps, stats = _check(models.eight_schools(), 64, 16, 300, 200, 42)
    assert len(eps) == 64                       # every chain tuned its own step size
    assert stats[0]["extra"]["warmup_leapfrogs"] > 0


def test_sv_8_chains_adapt_independently():
    """sv (d = 102, one chain per wavefront) 8 x (200 + 50)."""
    eps, _ = _check(models.sv(TG.GOLD["sv_returns"]), 8, 64, 200, 50, 7)
    assert len(eps) == 8


def test_shard_of_a_chain_range_and_explicit_init():
    """chains [3, 9) of 12 keep their seeds (seed + 7919 i whatever the shard); explicit init values;
    a warmup without windows (num_warmup 60: init buffer only, sampler.ex:559-575)."""
    spec = models.eight_schools()
    _check(spec, 12, 16, 60, 40, 11, init_values=spec.default_init, chain_lo=3, chain_hi=9)


def test_api_route_and_one_chain_equals_sample():
    """sampler.sample_chains(ir, n, {"vectorized": False}) routes here, and so does n = 1 whatever the
    flag (sampler.ex:993: vectorized defaults to num_chains > 1): one chain = sample/3."""
    spec = models.eight_schools()
    opts = dict(num_warmup=120, num_samples=60, seed=5, lanes_per_chain=16)
    tr, st = sampler.sample_chains(spec, 3, dict(opts, vectorized=False))
    assert len(tr) == 3 and len({s["step_size"] for s in st}) == 3
    one_t, one_s = sampler.sample_chains(spec, 1, opts)
    t1, s1 = sampler.sample(spec, None, opts)
    assert one_s[0]["step_size"] == s1["step_size"]
    for name in one_t[0]:
        assert np.array_equal(one_t[0][name], t1[name]), name
    assert np.array_equal(one_t[0]["mu"], tr[0]["mu"])   # chain 0 of three is that same chain
    # dense_mass / warm_start are forwarded to each chain as the reference does (sampler.ex:1146-1153):
    # chain i = sample/3 with seed + 7919 i, not a refusal
    dn_t, dn_s = sampler.sample_chains(spec, 2, dict(opts, vectorized=False, dense_mass=True))
    for i in range(2):
        ti, si = sampler.sample(spec, None, dict(opts, dense_mass=True, seed=5 + 7919 * i))
        assert dn_s[i]["step_size"] == si["step_size"] and np.array_equal(dn_s[i]["chol_cov"], si["chol_cov"])
        assert np.array_equal(dn_t[i]["tau"], ti["tau"])
    ws = dict(inv_mass_diag=s1["inv_mass_diag"], step_size=s1["step_size"])
    ws_t, ws_s = sampler.sample_chains(spec, 1, dict(opts, warm_start=ws, devices=[0]))
    tw, sw = sampler.sample(spec, None, dict(opts, warm_start=ws))
    assert ws_s[0]["step_size"] == sw["step_size"] and np.array_equal(ws_t[0]["mu"], tw["mu"])


def test_logistic_and_radon_chains_adapt_independently():
    """the other two lane layouts: logistic (16 lanes, four chains per wavefront, rows from L2 in this
    kernel) and radon (64 lanes, observations in LDS), small protocols."""
    _check(models.logistic(), 6, 16, 80, 30, 3)
    _check(models.radon(), 3, 64, 80, 30, 3)


def test_generated_models_adapt_independently():
    """A plug-in carries the kernel too (its part 7): eight schools written as Builder nodes, one lane
    per chain (64 chains of a wavefront adapt side by side) and in the 16-lane plate layout, against
    the generated text compiled for the host."""
    import gen_checker as GC
    from exmc_amd import codegen
    gen = codegen.generate(codegen.eight_schools_ir())
    init = {n: 0.0 for n in ["mu"] + ["theta_%d" % j for j in range(8)]}
    init["tau"] = 1.0
    spec = codegen.compile_ir(codegen.eight_schools_ir(), default_init=init)
    for lanes, n_chains in ((1, 70), (16, 6)):
        comp = sampler.compile(spec)
        try:
            opts = dict(num_warmup=100, num_samples=40, seed=21, lanes_per_chain=lanes, vectorized=False)
            _, stats = sampler.sample_chains_independent_compiled(comp, n_chains, opts)
        finally:
            comp.close()
        om = GC.model(gen, lanes)
        raw = stats[0]["extra"]["raw"]
        for c in range(n_chains):
            t, st = O.sample(om, num_warmup=100, num_samples=40, seed=21 + 7919 * c, cfg=O.Cfg(1, lanes))
            assert stats[c]["step_size"] == st.step_size, (lanes, c)
            assert np.array_equal(stats[c]["inv_mass_diag"], np.array(st.inv_mass[:spec.d])), (lanes, c)
            for key in ("draws", "tree_depth", "n_steps", "divergent", "logp", "accept_prob", "energy"):
                assert np.array_equal(raw[key][c], t[key]), (lanes, c, key)


@pytest.mark.parametrize("name,lanes,n_chains,spread", [("eight_schools", 16, 10, 40.0), ("eight_schools", 16, 10, 6.0),
                                                         ("logistic", 16, 5, 6.0), ("radon", 64, 3, 6.0)])
def test_chains_adapt_independently_from_a_hostile_start(name, lanes, n_chains, spread):
    """All chains start at the same point far from the mode (explicit init values, 6 or 40 units away on the
    unconstrained scale) and differ by their seeds only: each runs its own step-size search (dozens of halvings),
    its own first trees that diverge at the first leaf, its own walk back -- in lock step with its neighbours in
    the wavefront, which are somewhere else in that story. Every chain against the checker's sample/3."""
    spec = {"eight_schools": models.eight_schools, "logistic": models.logistic, "radon": models.radon}[name]()
    rng = np.random.default_rng(43)
    q_far = spec.to_unconstrained(spec.default_init) + rng.normal(size=spec.d) * spread
    init = {n: float(np.exp(q_far[i])) if spec.transforms.get(n) == "log" else float(q_far[i])
            for i, n in enumerate(spec.var_names)}
    # (no claim that the step sizes differ: a chain that never gets an acceptance follows dual averaging down the
    # same deterministic path as its neighbours, and two of these four cases end with one common step size)
    _check(spec, n_chains, lanes, 60, 20, 13, init_values=init)
