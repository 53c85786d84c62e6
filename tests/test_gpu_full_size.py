"""BASELINE.json's full single-GPU configuration (eight_schools, 4096 chains x 1000 draws after
the shared 1000-iteration warmup) through the C ABI, checked with properties that do not need the
CPU checker to run 4 million transitions: a sample of chains bit for bit against the checker, shard
independence, run-to-run determinism of the whole trace, convergence diagnostics. Plus the edge
shapes of the entry points (no draws, depth 1, one chain, ragged last wavefront)."""
import hashlib

import numpy as np
import pytest

import oracle as O
from exmc_amd import distributed, models, sampler

pytestmark = pytest.mark.gpu

KEYS = ("tree_depth", "n_steps", "divergent", "draws", "logp", "accept_prob", "energy")


def _digest(raw):
    h = hashlib.sha256()
    for k in KEYS:
        h.update(np.ascontiguousarray(raw[k]).tobytes())
    return h.hexdigest()


def test_eight_schools_4096_chains_1000_draws(hip):
    spec = models.eight_schools()
    comp = sampler.compile(spec)
    om = O.eight_schools()
    n_chains, n_draws = 4096, 1000
    opts = dict(num_warmup=1000, num_samples=n_draws, seed=42, lanes_per_chain=16)
    tuning = sampler.warmup(comp, spec.default_init, opts)
    _, _, extra = sampler.sample_compiled_tuned(comp, tuning, spec.default_init, opts,
                                                num_chains=n_chains)
    raw = extra["raw"]
    assert raw["draws"].shape == (n_chains, n_draws, spec.d)

    # (1) chains from the first, a middle and the last wavefront equal the checker's, every output
    q0 = spec.to_unconstrained(spec.default_init)
    for lo in (0, 2050, n_chains - 3):
        hi = lo + 3
        t, st = O.sample_chains(om, n_chains, init_q=q0, num_warmup=1000, num_samples=n_draws,
                                seed=42, chain_lo=lo, chain_hi=hi, cfg=O.Cfg(1, 16))
        assert st.step_size == tuning["epsilon"]
        for k in KEYS:
            assert np.array_equal(t[k], raw[k][lo:hi]), (k, lo)

    # (2) a shard run on its own reproduces the same rows (seed + 7919 i is shard independent)
    _, _, ex2 = sampler.sample_compiled_tuned(comp, tuning, spec.default_init, opts,
                                              num_chains=n_chains, chain_lo=1021, chain_hi=1031)
    for k in KEYS:
        assert np.array_equal(ex2["raw"][k], raw[k][1021:1031]), k

    # (3) the counters the bench divides by are the trace's own sums
    assert extra["total_leapfrogs"] == int(raw["n_steps"].astype(np.int64).sum())
    assert extra["total_divergences"] == int(raw["divergent"].astype(np.int64).sum())

    # (4) the whole 4096 x 1000 trace is reproducible bit for bit
    _, _, ex3 = sampler.sample_compiled_tuned(comp, tuning, spec.default_init, opts,
                                              num_chains=n_chains)
    assert _digest(ex3["raw"]) == _digest(raw)

    # (5) it is a sample of the posterior: split R-hat, acceptance, divergences, depth bound
    import torch
    draws = torch.from_numpy(np.ascontiguousarray(raw["draws"].transpose(1, 2, 0)))   # [S][d][C]
    rhat = distributed.split_rhat(draws)
    assert tuple(rhat.shape) == (spec.d,) and float(rhat.max()) < 1.01
    assert 0.75 < float(raw["accept_prob"].mean()) < 0.95
    assert float(raw["divergent"].mean()) < 0.005
    assert int(raw["tree_depth"].max()) <= 10 and int(raw["tree_depth"].min()) >= 1
    assert np.all(raw["n_steps"] <= (1 << raw["tree_depth"]) - 1)
    assert np.isfinite(raw["draws"]).all() and np.isfinite(raw["energy"]).all()


@pytest.mark.parametrize("n_chains,lanes", [(1, 16), (5, 16), (3, 1), (65, 1), (9, 8)])
def test_ragged_chain_counts(hip, n_chains, lanes):
    """Chain counts that leave the last wavefront partly empty (or nearly empty)."""
    spec = models.eight_schools()
    comp = sampler.compile(spec)
    om = O.eight_schools()
    opts = dict(num_warmup=0, num_samples=15, seed=5, lanes_per_chain=lanes)
    tuning = dict(epsilon=0.4, inv_mass=np.ones(spec.d), chol_cov=None, warmup_divergences=0)
    _, _, extra = sampler.sample_compiled_tuned(comp, tuning, spec.default_init, opts,
                                                num_chains=n_chains)
    assert extra["raw"]["draws"].shape == (n_chains, 15, spec.d)
    q0 = spec.to_unconstrained(spec.default_init)
    for c in sorted({0, n_chains // 2, n_chains - 1}):
        t, _ = O.sample_tuned(om, 0.4, np.ones(spec.d), q0, num_samples=15, seed=5 + 7919 * c,
                              cfg=O.Cfg(1, lanes))
        for k in KEYS:
            assert np.array_equal(t[k], extra["raw"][k][c]), (k, c)


def test_no_draws_and_depth_one(hip):
    spec = models.eight_schools()
    comp = sampler.compile(spec)
    om = O.eight_schools()
    tuning = dict(epsilon=0.3, inv_mass=np.ones(spec.d), chol_cov=None, warmup_divergences=0)
    # zero draws: empty traces, no leapfrogs, no error
    _, _, ex0 = sampler.sample_compiled_tuned(comp, tuning, spec.default_init,
                                              dict(num_warmup=0, num_samples=0, seed=1), num_chains=7)
    assert ex0["raw"]["draws"].shape == (7, 0, spec.d) and ex0["total_leapfrogs"] == 0
    # max_tree_depth 1: every transition is one doubling (n_steps == 1)
    opts = dict(num_warmup=0, num_samples=25, seed=2, max_tree_depth=1, lanes_per_chain=16)
    _, _, ex1 = sampler.sample_compiled_tuned(comp, tuning, spec.default_init, opts, num_chains=6)
    assert np.all(ex1["raw"]["n_steps"] == 1) and np.all(ex1["raw"]["tree_depth"] == 1)
    t, _ = O.sample_tuned(om, 0.3, np.ones(spec.d), spec.to_unconstrained(spec.default_init),
                          num_samples=25, max_tree_depth=1, seed=2, cfg=O.Cfg(1, 16))
    for k in KEYS:
        assert np.array_equal(t[k], ex1["raw"][k][0]), k


@pytest.mark.parametrize("name,lanes,n_chains,n_draws", [("eight_schools", 16, 2048, 150),
                                                        ("logistic", 16, 96, 40),
                                                        ("radon", 64, 48, 40), ("sv", 64, 12, 12)])
def test_every_chain_of_a_batch_bit_exact(hip, name, lanes, n_chains, n_draws):
    """Every chain of a batch against the checker (not a sample of chains): hundreds of thousands of
    transitions exercise the rare paths of the lock-step walk — subtrees ending early on a U-turn
    or a divergence while their wave neighbours continue, several ziggurat words of one momentum
    draw needing the long way, trees of different depth in one wavefront."""
    import bench
    spec = bench.make_spec(name)[0]
    comp = sampler.compile(spec)
    om = O.model_for(spec)
    opts = dict(num_warmup=1000, num_samples=n_draws, seed=123, lanes_per_chain=lanes)
    tuning = sampler.warmup(comp, spec.default_init, opts)
    _, _, extra = sampler.sample_compiled_tuned(comp, tuning, spec.default_init, opts,
                                                num_chains=n_chains)
    t, st = O.sample_chains(om, n_chains, init_q=spec.to_unconstrained(spec.default_init),
                            num_warmup=1000, num_samples=n_draws, seed=123, n_threads=8,
                            cfg=O.Cfg(1, lanes))
    assert st.step_size == tuning["epsilon"]
    for k in KEYS:
        assert np.array_equal(t[k], extra["raw"][k]), k
    assert extra["total_leapfrogs"] == st.total_leapfrogs
    if name == "eight_schools":
        ns = extra["raw"]["n_steps"]
        assert (ns != (1 << extra["raw"]["tree_depth"]) - 1).sum() > 50   # early-ended subtrees occurred


def _bench_models():
    import bench
    return [("sv", lambda: bench.make_spec("sv")[0], 2048, 64, 64),
            ("logistic", lambda: bench.make_spec("logistic")[0], 8192, 16, 64),
            ("radon", lambda: bench.make_spec("radon")[0], 1024, 64, 64)]


@pytest.mark.parametrize("name,factory,n_chains,lanes,warm_lanes", _bench_models(),
                         ids=lambda x: x if isinstance(x, str) else "")
def test_other_baseline_configs_at_full_size(hip, name, factory, n_chains, lanes, warm_lanes):
    """The other BASELINE.json configurations at the sizes and layouts `bench.py --model <name>`
    runs (sv 2048 x 1000 at 64 lanes, logistic 8192 x 1000 at 16 lanes after its 64-lane warmup,
    radon 1024 x 1000 at 64 lanes): chains from the first, a middle and the last wavefront against
    the checker bit for bit, shard independence, the counters, run-to-run determinism."""
    spec = factory()
    comp = sampler.compile(spec)
    om = O.model_for(spec)
    n_draws = 1000
    opts = dict(num_warmup=1000, num_samples=n_draws, seed=42, lanes_per_chain=lanes)
    tuning = sampler.warmup(comp, spec.default_init, dict(opts, warmup_lanes=warm_lanes))
    q0 = spec.to_unconstrained(spec.default_init)
    st = O.warmup(om, q0, num_warmup=1000, seed=42, cfg=O.Cfg(1, warm_lanes))
    assert st.step_size == tuning["epsilon"]
    assert np.array_equal(np.array(st.inv_mass[:spec.d]), tuning["inv_mass"])
    _, _, extra = sampler.sample_compiled_tuned(comp, tuning, spec.default_init, opts, num_chains=n_chains)
    raw = extra["raw"]
    assert raw["draws"].shape == (n_chains, n_draws, spec.d)
    for c in (0, n_chains // 2 + 1, n_chains - 1):
        t, _ = O.sample_tuned(om, tuning["epsilon"], tuning["inv_mass"], q0, num_samples=n_draws,
                              seed=42 + 7919 * c, cfg=O.Cfg(1, lanes))
        for k in ("tree_depth", "n_steps", "divergent", "draws", "accept_prob", "energy"):
            assert np.array_equal(t[k], raw[k][c]), (name, c, k)
    lo = n_chains // 3
    _, _, ex2 = sampler.sample_compiled_tuned(comp, tuning, spec.default_init, opts,
                                              num_chains=n_chains, chain_lo=lo, chain_hi=lo + 5)
    for k in KEYS:
        assert np.array_equal(ex2["raw"][k], raw[k][lo:lo + 5]), (name, k)
    assert extra["total_leapfrogs"] == int(raw["n_steps"].astype(np.int64).sum())
    assert extra["total_divergences"] == int(raw["divergent"].astype(np.int64).sum())
    d1 = _digest(raw)
    del raw, extra
    _, _, ex3 = sampler.sample_compiled_tuned(comp, tuning, spec.default_init, opts, num_chains=n_chains)
    assert _digest(ex3["raw"]) == d1
    r = ex3["raw"]
    assert int(r["tree_depth"].max()) <= 10 and np.all(r["n_steps"] <= (1 << r["tree_depth"]) - 1)
    assert np.isfinite(r["draws"]).all() and float(r["divergent"].mean()) < 0.05
