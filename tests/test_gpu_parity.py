"""GPU parity: the HIP path (through the C ABI) vs the CPU checker, bit for bit.

The checker runs in deterministic-math mode with the same lane layout G as the kernel
(oracle/exmc_oracle.h). Integer outputs (tree depth, n_steps, divergence flags) AND floating
outputs (positions, log-prob, accept stat, energy) are required to be identical bits; the
tolerance against libm mode (the reference's own arithmetic) is asserted separately in
tests/test_oracle_sampler.py.
"""
import ctypes as C

import numpy as np
import pytest

import oracle as O
from exmc_amd import _lib, models, sampler

pytestmark = pytest.mark.gpu


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


@pytest.fixture(scope="module")
def es(hip):
    spec = models.eight_schools()
    return spec, sampler.compile(spec), O.eight_schools()


def _rand_q(rng, n, d, scale=1.0):
    return np.ascontiguousarray(rng.normal(size=(n, d)) * scale)


@pytest.mark.parametrize("lanes", [1, 2, 4, 8, 16])
def test_logp_grad_bit_exact(es, hip, lanes):
    spec, comp, om = es
    rng = np.random.default_rng(7)
    C_ = 257
    q = _rand_q(rng, C_, spec.d, 1.5)
    q[0] = 0.0
    q[1, 1] = 250.0   # outside the :log clamp (transform.ex:17-29)
    q[2, 1] = -250.0
    lp = np.zeros(C_)
    g = np.zeros((C_, spec.d))
    _lib.check(hip.exmc_hip_logp_grad_host(comp.h, _dp(q), C_, lanes, _dp(lp), _dp(g)))
    cfg = O.Cfg(1, lanes)
    for c in range(C_):
        olp, og = om.logp_grad(q[c], cfg)
        assert olp == lp[c] or (np.isnan(olp) and np.isnan(lp[c])), (c, olp, lp[c])
        assert np.array_equal(og, g[c]), (c, og, g[c])


def _other_models():
    import test_golden_traces as TG
    return [("sv", lambda: models.sv(TG.GOLD["sv_returns"]), [32, 64]),
            ("logistic", models.logistic, [4, 8, 16]),
            ("radon", models.radon, [32, 64])]


@pytest.mark.parametrize("name,factory,lane_list", _other_models(), ids=lambda x: x if isinstance(x, str) else "")
def test_model_logp_grad_bit_exact(hip, name, factory, lane_list):
    """vag_fn (compiler.ex:131-141) for the other BASELINE configs: sv (d=102), logistic (d=21,
    N=500 dense X@beta), radon (d=90, 919 observations in 85 counties)."""
    spec = factory()
    comp = sampler.compile(spec)
    om = O.model_for(spec)
    rng = np.random.default_rng(3)
    C_ = 37
    q = _rand_q(rng, C_, spec.d, 0.4)
    q[0] = spec.to_unconstrained(spec.default_init)
    for lanes in lane_list:
        lp = np.zeros(C_)
        g = np.zeros((C_, spec.d))
        _lib.check(hip.exmc_hip_logp_grad_host(comp.h, _dp(q), C_, lanes, _dp(lp), _dp(g)))
        cfg = O.Cfg(1, lanes)
        for c in range(C_):
            olp, og = om.logp_grad(q[c], cfg)
            assert olp == lp[c], (name, lanes, c, olp, lp[c])
            assert np.array_equal(og, g[c]), (name, lanes, c)


def _skewed_radon():
    """County sizes like the real radon survey (one county with 116 observations, one with 105,
    many with 1-3), the large ones LAST in file order: their owner lanes sit in the second dimension
    slot of the 64-lane layout and add up to 116 cells of the wavefront's strip."""
    rng = np.random.default_rng(17)
    J, N = 85, 919
    sizes = np.concatenate([rng.integers(1, 6, size=J - 6), [14, 25, 46, 52, 105, 116]])
    sizes[:J - 6] += 1
    while sizes.sum() > N:
        sizes[int(np.argmax(sizes[:J - 6]))] -= 1
    while sizes.sum() < N:
        sizes[int(rng.integers(0, J - 6))] += 1
    start = np.concatenate([[0], np.cumsum(sizes)])
    u = rng.normal(0.0, 0.5, size=J)
    county = np.repeat(np.arange(J), sizes)
    floor = (rng.uniform(size=N) < 0.2).astype(float)
    y = 1.4 + 0.7 * u[county] - 0.7 * floor + 0.7 * rng.normal(size=N)
    return u, start, floor, y


@pytest.mark.parametrize("sort_counties", [False, True])
def test_radon_chunked_counties_bit_exact(hip, sort_counties):
    """Radon with 64 lanes per chain on survey-like county sizes: the observations are spread over
    the lanes whatever their county and a county's owner lane adds its cells in index order (the
    test keeps its name from the chunked walk of rounds 1-2); in file order the large counties are
    owned by the second dimension slot. logp / gradient and whole transitions."""
    spec = models.radon(_skewed_radon(), sort_counties=sort_counties)
    comp = sampler.compile(spec)
    om = O.model_for(spec)
    rng = np.random.default_rng(5)
    C_ = 21
    q = _rand_q(rng, C_, spec.d, 0.4)
    lp = np.zeros(C_)
    g = np.zeros((C_, spec.d))
    _lib.check(hip.exmc_hip_logp_grad_host(comp.h, _dp(q), C_, 64, _dp(lp), _dp(g)))
    cfg = O.Cfg(1, 64)
    for c in range(C_):
        olp, og = om.logp_grad(q[c], cfg)
        assert olp == lp[c], (c, olp, lp[c])
        assert np.array_equal(og, g[c]), c
    # 32 lanes per chain keep whole counties: a different (documented) sum order, same value to rounding
    lp32 = np.zeros(C_)
    _lib.check(hip.exmc_hip_logp_grad_host(comp.h, _dp(q), C_, 32, _dp(lp32), _dp(g)))
    assert np.allclose(lp32, lp, rtol=1e-12, atol=0)
    opts = dict(num_warmup=40, num_samples=30, seed=3, lanes_per_chain=64)
    tuning = sampler.warmup(comp, spec.default_init, opts)
    st = O.warmup(om, spec.to_unconstrained(spec.default_init), num_warmup=40, seed=3, cfg=O.Cfg(1, 64))
    assert tuning["epsilon"] == st.step_size
    assert np.array_equal(tuning["inv_mass"], np.array(st.inv_mass[:spec.d]))


def _radon_shape(sizes, seed):
    rng = np.random.default_rng(seed)
    sizes = np.asarray(sizes, int)
    J, N = len(sizes), int(sizes.sum())
    assert J == 85
    start = np.concatenate([[0], np.cumsum(sizes)])
    u = rng.normal(0.0, 0.5, size=J)
    county = np.repeat(np.arange(J), sizes)
    floor = (rng.uniform(size=N) < 0.2).astype(float)
    y = 1.4 + 0.7 * u[county] - 0.7 * floor + 0.7 * rng.normal(size=N)
    return u, start, floor, y


_RADON_SHAPES = {
    # every slot of the 64-lane layout full to the brim: 16 x 64 observations, no partly filled slot
    "n1024": [12] * 84 + [16],
    # fewer observations than lanes: no full slot at all, only the partly filled one
    "n40": [1] * 40 + [0] * 45,
    # exactly one full slot, an empty partly filled one; empty counties between full ones
    "n64": [0, 8, 0, 16, 0, 24, 0, 16] + [0] * 77,
    # counties of exactly one, two and five whole batches of eight, and 8 k + 7
    "batches": [40, 16, 8, 15, 23, 7, 1] + [3] * 78,
    # one county holds nearly everything (its owner lane walks 100 whole batches of eight)
    "one_big": [800] + [1] * 84,
}


@pytest.mark.parametrize("shape", sorted(_RADON_SHAPES))
@pytest.mark.parametrize("sort_counties", [False, True])
def test_radon_observation_and_county_shapes_bit_exact(hip, shape, sort_counties):
    """The 64-lane radon layout's addressing at its edges (round 5: observations fetched by slot from
    zero-padded copies, the slots every lane fills behind a scalar test, county sums in whole batches of
    eight cells plus up to seven more): no full slot, only full slots, the capacity, empty counties,
    counties of exactly k batches, one county with 800 observations. logp / gradient and a short run."""
    sizes = _RADON_SHAPES[shape]
    assert min(sizes) >= 0 and sum(sizes) <= 1024
    spec = models.radon(_radon_shape(sizes, 11), sort_counties=sort_counties)
    comp = sampler.compile(spec)
    om = O.model_for(spec)
    rng = np.random.default_rng(7)
    C_ = 9
    q = _rand_q(rng, C_, spec.d, 0.4)
    lp = np.zeros(C_)
    g = np.zeros((C_, spec.d))
    _lib.check(hip.exmc_hip_logp_grad_host(comp.h, _dp(q), C_, 64, _dp(lp), _dp(g)))
    cfg = O.Cfg(1, 64)
    for c in range(C_):
        olp, og = om.logp_grad(q[c], cfg)
        assert olp == lp[c], (shape, c, olp, lp[c])
        assert np.array_equal(og, g[c]), (shape, c)
    opts = dict(num_warmup=30, num_samples=12, seed=3, lanes_per_chain=64)
    tuning = sampler.warmup(comp, spec.default_init, opts)
    st = O.warmup(om, spec.to_unconstrained(spec.default_init), num_warmup=30, seed=3, cfg=cfg)
    assert tuning["epsilon"] == st.step_size
    assert np.array_equal(tuning["inv_mass"], np.array(st.inv_mass[:spec.d]))


@pytest.mark.parametrize("lanes", [1, 8, 16])
@pytest.mark.parametrize("eps", [0.3, -0.3])
def test_multi_step_bit_exact(es, hip, lanes, eps):
    spec, comp, om = es
    rng = np.random.default_rng(11)
    C_, n = 70, 37
    d = spec.d
    q = _rand_q(rng, C_, d, 0.5)
    p = _rand_q(rng, C_, d)
    im = np.ascontiguousarray(rng.uniform(0.5, 2.0, size=d))
    cfg = O.Cfg(1, lanes)
    g = np.array([om.logp_grad(q[c], cfg)[1] for c in range(C_)])
    aq = np.zeros((C_, n, d)); ap = np.zeros((C_, n, d)); ag = np.zeros((C_, n, d))
    alp = np.zeros((C_, n))
    _lib.check(hip.exmc_hip_multi_step_host(comp.h, _dp(q), _dp(p), _dp(g), eps, _dp(im), n, C_,
                                            lanes, _dp(aq), _dp(ap), _dp(alp), _dp(ag)))
    for c in range(0, C_, 7):
        oq, op, olp, og = om.multi_step(q[c], p[c], g[c], eps, im, n, cfg)
        assert np.array_equal(oq, aq[c]) and np.array_equal(op, ap[c])
        assert np.array_equal(olp, alp[c]) and np.array_equal(og, ag[c])


def _oracle_transitions(om, q, logp, g, rngs, n_draws, eps, im, max_depth, cfg):
    """Apply n_draws NUTS transitions per chain with the checker (sampler.ex:854-925)."""
    L = O.lib()
    C_, d = q.shape
    out = dict(draws=np.zeros((C_, n_draws, d)), logp=np.zeros((C_, n_draws)),
               tree_depth=np.zeros((C_, n_draws), np.int32), n_steps=np.zeros((C_, n_draws), np.int32),
               divergent=np.zeros((C_, n_draws), np.int32), accept_prob=np.zeros((C_, n_draws)),
               energy=np.zeros((C_, n_draws)))
    for c in range(C_):
        r = O.Rng(int(rngs[c, 0]), int(rngs[c, 1]))
        qc, gc, lpc = q[c].copy(), g[c].copy(), float(logp[c])
        for s in range(n_draws):
            p = np.array([L.exo_rng_normal(C.byref(r), cfg.math_mode) / np.sqrt(im[i])
                          for i in range(d)])
            ke = L.exo_kinetic_energy(_dp(p), _dp(im), d, cfg)
            jlp0 = lpc - ke
            qo, go, res = om.tree_build(qc, p, lpc, gc, eps, im, max_depth, r, jlp0, cfg)
            L.exo_rng_uniform(C.byref(r))
            qc, gc, lpc = qo, go, res.logp
            out["draws"][c, s] = qc
            out["logp"][c, s] = lpc
            out["tree_depth"][c, s] = res.depth
            out["n_steps"][c, s] = res.n_steps
            out["divergent"][c, s] = res.divergent
            out["accept_prob"][c, s] = res.accept_sum / res.n_steps if res.n_steps else 0.0
            out["energy"][c, s] = -jlp0
        rngs[c, 0], rngs[c, 1] = r.a, r.b
        q[c], g[c], logp[c] = qc, gc, lpc
    return out


@pytest.mark.parametrize("lanes,eps,max_depth", [(1, 0.45, 10), (16, 0.45, 10), (8, 0.2, 6),
                                                 (16, 1.6, 10), (4, 0.02, 5), (2, 0.45, 3)])
def test_transitions_bit_exact(es, hip, lanes, eps, max_depth):
    """Whole transitions from explicit state: every per-draw output identical to the checker,
    including divergent trees (eps 1.6) and depth-capped trees (eps 0.02, depth 5)."""
    spec, comp, om = es
    rng = np.random.default_rng(5 + lanes)
    C_, n_draws, d = 33, 12, spec.d
    cfg = O.Cfg(1, lanes)
    q = _rand_q(rng, C_, d, 0.7)
    im = np.ascontiguousarray(rng.uniform(0.3, 3.0, size=d))
    g = np.zeros((C_, d)); logp = np.zeros(C_)
    for c in range(C_):
        logp[c], g[c] = om.logp_grad(q[c], cfg)
    rngs = np.zeros((C_, 2), dtype=np.uint64)
    for c in range(C_):
        r = O.Rng()
        O.lib().exo_rng_seed(C.byref(r), 1000 + c)
        rngs[c] = (r.a, r.b)
    hq, hg, hl, hr = q.copy(), g.copy(), logp.copy(), rngs.copy()
    t, tr = sampler._host_trace(C_, n_draws, d)
    _lib.check(hip.exmc_hip_transitions_host(comp.h, _dp(hq), _dp(hl), _dp(hg),
                                             hr.ctypes.data_as(C.POINTER(C.c_uint64)), C_, n_draws,
                                             eps, _dp(im), max_depth, lanes, tr))
    o = _oracle_transitions(om, q, logp, g, rngs, n_draws, eps, im, max_depth, cfg)
    for k in ("tree_depth", "n_steps", "divergent"):
        assert np.array_equal(o[k], t[k]), k
    for k in ("draws", "logp", "accept_prob", "energy"):
        assert np.array_equal(o[k], t[k]), k
    assert np.array_equal(hq, q) and np.array_equal(hg, g) and np.array_equal(hl, logp)
    assert np.array_equal(hr, rngs)
    if eps > 1.0:
        assert t["divergent"].sum() > 0


@pytest.mark.parametrize("lanes", [1, 16])
def test_warmup_and_chains_bit_exact(es, hip, lanes):
    """sample_chains vectorized semantics (sampler.ex:1020-1136): shared warmup on chain 0, then
    chains seeded seed + 7919*i. Tuning and every draw identical to the checker."""
    spec, comp, om = es
    opts = dict(num_warmup=150, num_samples=60, seed=42, lanes_per_chain=lanes)
    tuning = sampler.warmup(comp, spec.default_init, opts)
    q0 = spec.to_unconstrained(spec.default_init)
    cfg = O.Cfg(1, lanes)
    n_chains = 24
    t, st = O.sample_chains(om, n_chains, init_q=q0, num_warmup=150, num_samples=60, seed=42, cfg=cfg)
    assert st.step_size == tuning["epsilon"]
    assert np.array_equal(np.array(st.inv_mass[:spec.d]), tuning["inv_mass"])
    traces, stats, extra = sampler.sample_compiled_tuned(comp, tuning, spec.default_init, opts,
                                                         num_chains=n_chains)
    raw = extra["raw"]
    for k in ("tree_depth", "n_steps", "divergent", "draws", "logp", "accept_prob", "energy"):
        assert np.array_equal(t[k], raw[k]), k
    assert extra["total_leapfrogs"] == int(t["n_steps"].sum()) == st.total_leapfrogs
    # a shard [8, 16) reproduces the same chains (seed + 7919*i is shard independent)
    _, _, ex2 = sampler.sample_compiled_tuned(comp, tuning, spec.default_init, opts,
                                              num_chains=n_chains, chain_lo=8, chain_hi=16)
    assert np.array_equal(ex2["raw"]["draws"], raw["draws"][8:16])


@pytest.mark.parametrize("num_warmup", [0, 2, 60, 150, 320])
def test_device_warmup_equals_host_driven_and_oracle(es, hip, num_warmup, monkeypatch):
    """run_warmup (sampler.ex:537-762) in one kernel == one launch per transition with the
    adaptation scalars on the host == the checker; covers the short schedules (no Phase II when
    adapt_end <= init_buffer, sampler.ex:580-582)."""
    spec, comp, om = es
    opts = dict(num_warmup=num_warmup, num_samples=1, seed=9, lanes_per_chain=16)
    dev = sampler.warmup(comp, spec.default_init, opts)      # two-wave pipeline (the default)
    monkeypatch.setenv("EXMC_HIP_WARMUP_PIPE", "0")
    one = sampler.warmup(comp, spec.default_init, opts)      # one wave does both jobs
    monkeypatch.delenv("EXMC_HIP_WARMUP_PIPE")
    monkeypatch.setenv("EXMC_HIP_HOST_WARMUP", "1")
    host = sampler.warmup(comp, spec.default_init, opts)
    monkeypatch.delenv("EXMC_HIP_HOST_WARMUP")
    st = O.warmup(om, spec.to_unconstrained(spec.default_init), num_warmup=num_warmup, seed=9,
                  cfg=O.Cfg(1, 16))
    assert dev["epsilon"] == one["epsilon"] == host["epsilon"] == st.step_size
    assert np.array_equal(dev["inv_mass"], one["inv_mass"])
    assert dev["warmup_divergences"] == one["warmup_divergences"]
    assert np.array_equal(dev["inv_mass"], host["inv_mass"])
    assert np.array_equal(dev["inv_mass"], np.array(st.inv_mass[:spec.d]))
    assert dev["warmup_divergences"] == host["warmup_divergences"] == st.divergences


def test_random_init_bit_exact(es, hip):
    """init_values = %{} => 0.1 * normal_s per dimension (sampler.ex:339-349)."""
    spec, comp, om = es
    opts = dict(num_warmup=40, num_samples=20, seed=3, lanes_per_chain=16)
    tuning = sampler.warmup(comp, None, opts)
    t, st = O.sample_chains(om, 5, init_q=None, num_warmup=40, num_samples=20, seed=3,
                            cfg=O.Cfg(1, 16))
    assert st.step_size == tuning["epsilon"]
    _, _, extra = sampler.sample_compiled_tuned(comp, tuning, None, opts, num_chains=5)
    assert np.array_equal(t["draws"], extra["raw"]["draws"])
    assert np.array_equal(t["tree_depth"], extra["raw"]["tree_depth"])


def _permuted_models():
    import bench
    return [("sv", lambda: bench.make_spec("sv")[0], 64), ("sv32", lambda: bench.make_spec("sv")[0], 32),
            ("radon", lambda: bench.make_spec("radon")[0], 64),
            ("logistic", lambda: bench.make_spec("logistic")[0], 16),
            ("logistic8", lambda: bench.make_spec("logistic")[0], 8),
            ("logistic_mfma", lambda: bench.make_spec("logistic")[0], 4)]


@pytest.mark.parametrize("name,factory,lanes", _permuted_models(),
                         ids=lambda x: x if isinstance(x, str) else "")
def test_random_init_and_momentum_in_flat_order_bit_exact(hip, name, factory, lanes):
    """Models whose kernel layout is not the sorted-id order: the 0.1 * normal_s init
    (sampler.ex:339-349), the step-size search and every momentum draw (sampler.ex:393-403) consume
    the stream in PointMap's flat order (point_map.ex:30-60) on both sides."""
    spec = factory()
    assert spec.flat_order() != list(range(spec.d))
    comp = sampler.compile(spec)
    om = O.model_for(spec)
    opts = dict(num_warmup=30, num_samples=6, seed=5, lanes_per_chain=lanes, max_tree_depth=5)
    tuning = sampler.warmup(comp, None, opts)
    t, st = O.sample_chains(om, 5, init_q=None, num_warmup=30, num_samples=6, seed=5,
                            max_tree_depth=5, cfg=O.Cfg(1, lanes))
    assert st.step_size == tuning["epsilon"]
    assert np.array_equal(np.array(st.inv_mass[:spec.d]), tuning["inv_mass"])
    _, _, extra = sampler.sample_compiled_tuned(comp, tuning, None, opts, num_chains=5)
    for k in ("tree_depth", "n_steps", "divergent", "draws", "energy"):
        assert np.array_equal(t[k], extra["raw"][k]), k


def test_single_chain_sample_bit_exact(hip):
    """Sampler.sample/3 on the d=2 plumbing config (BASELINE configs[0])."""
    spec = models.simple()
    trace, stats = sampler.sample(spec, spec.default_init,
                                  dict(num_warmup=200, num_samples=100, seed=0))
    om = O.simple()
    t, st = O.sample(om, spec.to_unconstrained(spec.default_init), num_warmup=200, num_samples=100,
                     seed=0, cfg=O.Cfg(1, 1))
    assert st.step_size == stats["step_size"]
    assert np.array_equal(t["draws"], stats["raw"]["draws"][0])
    assert np.array_equal(t["tree_depth"], stats["raw"]["tree_depth"][0])
    assert st.divergences == stats["divergences"]
    assert abs(trace["mu"].mean() - 2.15) < 0.3


@pytest.mark.parametrize("num_warmup", [0, 30, 50, 400])
def test_warm_start_bit_exact(hip, num_warmup):
    """opts[:warm_start] (sampler.ex:167-197): the previous run's inv_mass_diag and step_size, a short
    warmup of min(num_warmup, 50) iterations without the initial step-size search, then sampling."""
    spec = models.eight_schools()
    first, st1 = sampler.sample(spec, spec.default_init, dict(num_warmup=300, num_samples=20, seed=1))
    ws = dict(inv_mass_diag=st1["inv_mass_diag"], step_size=st1["step_size"])
    _, st2 = sampler.sample(spec, spec.default_init,
                            dict(num_warmup=num_warmup, num_samples=60, seed=5, warm_start=ws))
    om = O.eight_schools()
    t, st = O.sample_warm(om, ws["step_size"], ws["inv_mass_diag"], spec.to_unconstrained(spec.default_init),
                          num_warmup=num_warmup, num_samples=60, seed=5, cfg=O.Cfg(1, 16))
    assert st.step_size == st2["step_size"]
    assert np.array_equal(np.array(st.inv_mass[:spec.d]), st2["inv_mass_diag"])
    assert np.array_equal(t["draws"], st2["raw"]["draws"][0])
    assert np.array_equal(t["n_steps"], st2["raw"]["n_steps"][0])
    assert st.divergences == st2["divergences"]
    if num_warmup == 0:
        assert st2["step_size"] == ws["step_size"]


def _bench_models():
    import bench
    return [("sv", lambda: bench.make_spec("sv")[0], 64, 6, 12),
            ("radon", lambda: bench.make_spec("radon")[0], 64, 6, 25),
            ("logistic", lambda: bench.make_spec("logistic")[0], 16, 37, 25),
            ("logistic_mfma", lambda: bench.make_spec("logistic")[0], 4, 37, 25)]


@pytest.mark.parametrize("name,factory,lanes,n_chains,n_draws", _bench_models(),
                         ids=lambda x: x if isinstance(x, str) else "")
def test_bench_protocol_other_models_bit_exact(hip, name, factory, lanes, n_chains, n_draws):
    """The bench protocol on the other BASELINE configs at full depth: the shared 1000-iteration
    warmup (sampler.ex:126-257; sv reaches tree depth 8-9, beyond the LDS-resident stack levels)
    and max_tree_depth 10 sampling of more chain groups than one wavefront holds. Tuning and every
    per-draw output identical to the checker."""
    spec = factory()
    comp = sampler.compile(spec)
    om = O.model_for(spec)
    opts = dict(num_warmup=1000, num_samples=n_draws, seed=42, lanes_per_chain=lanes)
    tuning = sampler.warmup(comp, spec.default_init, opts)
    q0 = spec.to_unconstrained(spec.default_init)
    t, st = O.sample_chains(om, n_chains, init_q=q0, num_warmup=1000, num_samples=n_draws, seed=42,
                            n_threads=8, cfg=O.Cfg(1, lanes))
    assert st.step_size == tuning["epsilon"]
    assert np.array_equal(np.array(st.inv_mass[:spec.d]), tuning["inv_mass"])
    _, _, extra = sampler.sample_compiled_tuned(comp, tuning, spec.default_init, opts,
                                                num_chains=n_chains)
    raw = extra["raw"]
    for k in ("tree_depth", "n_steps", "divergent", "draws", "logp", "accept_prob", "energy"):
        assert np.array_equal(t[k], raw[k]), k
    assert extra["total_leapfrogs"] == st.total_leapfrogs
    if name == "sv":
        assert t["tree_depth"].max() >= 7   # the global spill levels were exercised


def test_logistic_warmup_layout_differs_from_sampling_layout(hip):
    """logistic: the shared one-chain warmup runs with the chain spread over a whole wavefront
    (exmc_hip_model_default_warmup_lanes = 64: 8 observations per lane instead of 32), sampling
    with 16 lanes per chain. The tuning is layout-independent data; each phase equals the checker
    in its own layout, bit for bit."""
    spec = models.logistic()
    comp = sampler.compile(spec)
    assert comp.default_lanes == 16 and comp.default_warmup_lanes == 64
    om = O.model_for(spec)
    q0 = spec.to_unconstrained(spec.default_init)
    opts = dict(num_warmup=300, num_samples=30, seed=42)
    tuning = sampler.warmup(comp, spec.default_init, opts)                  # library defaults: 64 lanes
    st = O.warmup(om, q0, num_warmup=300, seed=42, cfg=O.Cfg(1, 64))
    assert st.step_size == tuning["epsilon"]
    assert np.array_equal(np.array(st.inv_mass[:spec.d]), tuning["inv_mass"])
    t16 = sampler.warmup(comp, spec.default_init, dict(opts, lanes_per_chain=16))   # a named layout is kept
    st16 = O.warmup(om, q0, num_warmup=300, seed=42, cfg=O.Cfg(1, 16))
    assert st16.step_size == t16["epsilon"]
    assert abs(st16.step_size - st.step_size) < 0.5 * st.step_size          # same sampler, other roundings
    _, _, extra = sampler.sample_compiled_tuned(comp, tuning, spec.default_init, opts, num_chains=20)
    raw = extra["raw"]
    for c in (0, 7, 19):
        t, _ = O.sample_tuned(om, tuning["epsilon"], tuning["inv_mass"], q0, num_samples=30,
                              seed=42 + 7919 * c, cfg=O.Cfg(1, 16))
        for k in ("draws", "tree_depth", "n_steps", "divergent", "energy", "accept_prob"):
            assert np.array_equal(t[k], raw[k][c]), (c, k)


@pytest.mark.parametrize("chunk", [1, 7, 64])
def test_sample_stream_equals_sample(hip, chunk):
    """sample_stream/4 (sampler.ex:1186-1277): the messages ("exmc_sample", i, point, stat) for
    i = 1..n then ("exmc_done", n), produced chunk by chunk from the resident chain, carry exactly
    the draws and statistics of sample/3 under the same seed, whatever the chunk size."""
    spec = models.eight_schools()
    opts = dict(num_warmup=120, num_samples=45, seed=8, stream_chunk=chunk)
    trace, stats = sampler.sample(spec, spec.default_init, opts)
    msgs = []
    assert sampler.sample_stream(spec, msgs.append, spec.default_init, opts) == "ok"
    assert msgs[-1] == ("exmc_done", 45) and len(msgs) == 46
    for i, (tag, idx, point, stat) in enumerate(msgs[:-1]):
        assert tag == "exmc_sample" and idx == i + 1
        assert all(point[k] == float(trace[k][i]) for k in trace)
        assert stat == stats["sample_stats"][i]


def _all_models():
    return [("eight_schools", models.eight_schools, [16]),
            ("sv", lambda: __import__("bench").make_spec("sv")[0], [64]),
            ("logistic", models.logistic, [16]),
            ("radon", models.radon, [64])]


@pytest.mark.parametrize("name,factory,lane_list", _all_models(), ids=lambda x: x if isinstance(x, str) else "")
def test_logp_grad_extreme_operands_bit_exact(hip, name, factory, lane_list):
    """The kernels divide without the hardware's range instructions only while every watched
    operand is inside 2^+-380 (exmc_device.hpp: Div); outside it a wavefront re-evaluates with the
    ordinary division. Positions that push scales, residuals and prior arguments to zero, to the
    clamp of the log transform, to 1e+-150 and beyond must give the checker's bits either way
    (NaN where the checker gives NaN)."""
    spec = factory()
    comp = sampler.compile(spec)
    om = O.model_for(spec)
    rng = np.random.default_rng(17)
    d = spec.d
    rows = [np.zeros(d), np.full(d, 1e-300), np.full(d, -1e-300), np.full(d, 1e-160),
            np.full(d, 345.0), np.full(d, -345.0), np.full(d, 1e150), np.full(d, -1e150),
            np.full(d, 199.999), np.full(d, -199.999), np.full(d, 200.0), np.full(d, -250.0)]
    for scale in (1e-200, 1e-8, 30.0, 150.0, 400.0, 1e100):
        rows.append(rng.normal(size=d) * scale)
    base = rng.normal(size=d) * 0.3
    for i in range(0, d, max(1, d // 12)):      # one extreme coordinate in an ordinary position
        for v in (0.0, 1e-310, 700.0, -700.0, 1e308):
            r = base.copy()
            r[i] = v
            rows.append(r)
    for i in range(max(0, d - 5), d):           # the scale / hyper-parameters sit at the end
        for v in (0.0, 150.0, -150.0, 260.0, -260.0):
            r = base.copy()
            r[i] = v
            rows.append(r)
    q = np.ascontiguousarray(np.array(rows))
    C_ = q.shape[0]
    for lanes in lane_list:
        lp = np.zeros(C_)
        g = np.zeros((C_, d))
        _lib.check(hip.exmc_hip_logp_grad_host(comp.h, _dp(q), C_, lanes, _dp(lp), _dp(g)))
        cfg = O.Cfg(1, lanes)
        for c in range(C_):
            olp, og = om.logp_grad(q[c], cfg)
            assert np.array_equal(np.array([olp]), np.array([lp[c]]), equal_nan=True), (name, c, olp, lp[c])
            assert np.array_equal(og, g[c], equal_nan=True), (name, c, og, g[c])


def test_wave_pair_sampling_kernel_equals_default(es, hip, monkeypatch):
    """EXMC_HIP_NUTS_PIPE=1: every workgroup of nuts_kernel is a tree wave + an integrator wave
    (the warmup's pipeline applied to the sampling phase; opt-in, slower on a full chip)."""
    spec, comp, _ = es
    opts = dict(num_warmup=120, num_samples=150, seed=5, lanes_per_chain=16)
    tuning = sampler.warmup(comp, spec.default_init, opts)
    _, _, a = sampler.sample_compiled_tuned(comp, tuning, spec.default_init, opts, num_chains=37)
    monkeypatch.setenv("EXMC_HIP_NUTS_PIPE", "1")
    _, _, b = sampler.sample_compiled_tuned(comp, tuning, spec.default_init, opts, num_chains=37)
    monkeypatch.delenv("EXMC_HIP_NUTS_PIPE")
    for k in ("draws", "logp", "tree_depth", "n_steps", "divergent", "accept_prob", "energy"):
        assert np.array_equal(a["raw"][k], b["raw"][k]), k
    assert a["total_leapfrogs"] == b["total_leapfrogs"]


@pytest.mark.parametrize("pipe", ["1", "0"])
def test_warmup_replicas_race_to_the_same_result(es, hip, monkeypatch, pipe):
    """EXMC_HIP_WARMUP_REPLICAS: the warmup launch is N workgroups running the same deterministic
    chain, the first to finish publishes; 1, the default 32 and 200 replicas give the same tuning
    and leave the same chain state behind (the next sampling run is identical)."""
    spec, comp, _ = es
    opts = dict(num_warmup=130, num_samples=40, seed=21, lanes_per_chain=16)
    monkeypatch.setenv("EXMC_HIP_WARMUP_PIPE", pipe)
    out = []
    for reps in ("1", "32", "200"):
        monkeypatch.setenv("EXMC_HIP_WARMUP_REPLICAS", reps)
        trace, stats = sampler.sample_compiled(comp, spec.default_init, opts)
        out.append((stats["step_size"], stats["inv_mass_diag"].copy(), stats["raw"]["draws"].copy(),
                    stats["divergences"]))
    for o in out[1:]:
        assert o[0] == out[0][0] and np.array_equal(o[1], out[0][1])
        assert np.array_equal(o[2], out[0][2]) and o[3] == out[0][3]


@pytest.mark.parametrize("name,lanes", [("eight_schools", 1), ("eight_schools", 16), ("simple", 1)])
def test_dense_mass_bit_exact(hip, name, lanes):
    """opts[:dense_mass] on the one-lane-per-chain layouts and on eight_schools' row layout (16
    lanes, one dimension per lane: matrix rows in registers, products by v_fmac_f64_dpp, the same
    fma chains in the same order): dense Welford windows of base
    max(25, 10 d), covariance shrinkage and Cholesky on the device, momentum p = L^-T z, M^-1 p by
    the dense product, U-turn through v = M^-1 rho -- tuning (step size, covariance, factor) and
    every per-draw output identical to the checker's restatement (tests/test_dense_mass_oracle.py
    pins that restatement; the reference raises in this mode for d >= 2)."""
    spec = models.eight_schools() if name == "eight_schools" else models.simple()
    om = O.eight_schools() if name == "eight_schools" else O.simple()
    comp = sampler.compile(spec)
    opts = dict(num_warmup=600, num_samples=40, seed=13, lanes_per_chain=lanes, dense_mass=True)
    tuning = sampler.warmup(comp, spec.default_init, opts)
    q0 = spec.to_unconstrained(spec.default_init)
    st, cov, chol = O.warmup_dense(om, q0, num_warmup=600, seed=13, cfg=O.Cfg(1, lanes))
    assert st.step_size == tuning["epsilon"]
    assert np.array_equal(cov, tuning["cov"]) and np.array_equal(chol, tuning["chol_cov"])
    assert np.array_equal(np.diag(cov), tuning["inv_mass_diag"])
    assert np.abs(cov - np.diag(np.diag(cov))).max() > 0          # it is a dense matrix
    _, _, extra = sampler.sample_compiled_tuned(comp, tuning, spec.default_init, opts, num_chains=5)
    raw = extra["raw"]
    for c in range(5):
        t, _ = O.sample_tuned_dense(om, st.step_size, cov, chol, q0, num_samples=40, seed=13 + 7919 * c,
                                    cfg=O.Cfg(1, lanes))
        for k in ("draws", "tree_depth", "n_steps", "divergent", "energy", "accept_prob"):
            assert np.array_equal(t[k], raw[k][c]), (c, k)
    # Sampler.sample/3 with dense_mass: the warmup chain goes on sampling under the dense mass
    tr1, st1 = sampler.sample(spec, spec.default_init, dict(opts, num_warmup=300, num_samples=25))
    o_st, o_cov, o_chol = O.warmup_dense(om, q0, num_warmup=300, seed=13, cfg=O.Cfg(1, lanes))
    assert st1["step_size"] == o_st.step_size and np.array_equal(st1["cov"], o_cov)
    assert np.array_equal(st1["chol_cov"], o_chol)
    # back to a diagonal tuning on the same handle: the dense mass is no longer in force
    diag = sampler.warmup(comp, spec.default_init, dict(num_warmup=100, seed=13, lanes_per_chain=lanes))
    _, _, e2 = sampler.sample_compiled_tuned(comp, diag, spec.default_init,
                                             dict(num_samples=10, seed=13, lanes_per_chain=lanes), num_chains=2)
    t2, _ = O.sample_tuned(om, diag["epsilon"], diag["inv_mass"], q0, num_samples=10, seed=13, cfg=O.Cfg(1, lanes))
    assert np.array_equal(t2["draws"], e2["raw"]["draws"][0])
    # the other multi-lane layouts refuse the mode
    if name == "eight_schools":
        with pytest.raises(Exception):
            sampler.warmup(comp, spec.default_init, dict(opts, lanes_per_chain=8))


@pytest.mark.parametrize("name,lanes,W", [("logistic", 16, 500), ("radon", 64, 400), ("sv", 64, 300)])
def test_dense_mass_lane_layouts_bit_exact(hip, name, lanes, W):
    """opts[:dense_mass] for the kinds that live in a lane layout (d = 21, 90, 102 over 16 / 64
    lanes, exmc_models.hpp LaneDenseModel): the covariance and its factor belong to the reference's
    FLAT vector (sampler.ex:682-705 feeds Welford the flat q; sv and logistic have a kernel order
    that is not the flat order), every contraction runs in ascending flat index. Dense Welford
    window + Cholesky on the device, p = L^-T z, M^-1 p, U-turn through M^-1 rho: tuning and every
    per-draw output equal the checker's."""
    import test_golden_traces as TG
    spec = {"sv": lambda: models.sv(TG.GOLD["sv_returns"]), "logistic": models.logistic, "radon": models.radon}[name]()
    comp = sampler.compile(spec)
    om = O.model_for(spec)
    assert comp.default_dense_lanes == lanes
    opts = dict(num_warmup=W, num_samples=25, seed=13, dense_mass=True)      # the layout is the library's choice
    tuning = sampler.warmup(comp, spec.default_init, opts)
    q0 = spec.to_unconstrained(spec.default_init)
    st, cov, chol = O.warmup_dense(om, q0, num_warmup=W, seed=13, cfg=O.Cfg(1, lanes))
    assert st.step_size == tuning["epsilon"]
    assert np.array_equal(cov, tuning["cov"]) and np.array_equal(chol, tuning["chol_cov"])
    flat = np.asarray(om.flat_order())                                         # flat entry -> kernel dimension
    assert np.array_equal(np.diag(cov), tuning["inv_mass_diag"][flat])
    assert np.abs(cov - np.diag(np.diag(cov))).max() > 0
    _, _, extra = sampler.sample_compiled_tuned(comp, tuning, spec.default_init, opts, num_chains=6)
    raw = extra["raw"]
    for c in range(6):
        t, _ = O.sample_tuned_dense(om, st.step_size, cov, chol, q0, num_samples=25, seed=13 + 7919 * c,
                                    cfg=O.Cfg(1, lanes))
        for k in ("draws", "tree_depth", "n_steps", "divergent", "energy", "accept_prob"):
            assert np.array_equal(t[k], raw[k][c]), (name, c, k)
    # sample/3: the warmup chain goes on sampling under the dense mass
    tr1, st1 = sampler.sample(spec, spec.default_init, dict(opts, num_samples=15))
    assert st1["step_size"] == st.step_size and np.array_equal(st1["cov"], cov)
    assert np.array_equal(st1["chol_cov"], chol) and np.all(np.isfinite(st1["raw"]["draws"]))
    # a diagonal tuning on the same handle afterwards: the dense mass is no longer in force
    diag = sampler.warmup(comp, spec.default_init, dict(num_warmup=60, seed=13, lanes_per_chain=lanes))
    _, _, e2 = sampler.sample_compiled_tuned(comp, diag, spec.default_init,
                                             dict(num_samples=8, seed=13, lanes_per_chain=lanes), num_chains=2)
    t2, _ = O.sample_tuned(om, diag["epsilon"], diag["inv_mass"], q0, num_samples=8, seed=13, cfg=O.Cfg(1, lanes))
    assert np.array_equal(t2["draws"], e2["raw"]["draws"][0])


def test_chain_migration_bit_exact(hip, monkeypatch, capfd):
    """sv's sampling kernel (one chain per wave, two waves per SIMD) lets a chain move, between
    transitions, to a SIMD that has run empty (exmc_nuts.hpp "chain migration"): the chain's state
    (q, g, logp, generator) crosses through the state arrays, so every output is the same with the
    mode forced on, forced off, and in the checker -- and chains do move."""
    import test_golden_traces as TG
    spec = models.sv(TG.GOLD["sv_returns"])
    comp = sampler.compile(spec)
    om = O.model_for(spec)
    opts = dict(num_warmup=150, num_samples=50, seed=5)
    tuning = sampler.warmup(comp, spec.default_init, opts)
    res = {}
    monkeypatch.setenv("EXMC_HIP_MIGRATE_STATS", "1")
    for mig in ("0", "1"):
        monkeypatch.setenv("EXMC_HIP_MIGRATE", mig)
        _, _, extra = sampler.sample_compiled_tuned(comp, tuning, spec.default_init, opts, num_chains=2048)
        res[mig] = extra["raw"]
    err = capfd.readouterr().err
    moved = [int(ln.split("moved")[1].split(",")[0]) for ln in err.splitlines() if "[exmc migrate]" in ln]
    assert moved and moved[-1] > 0 and "chains left 0" in err, err
    for k in ("draws", "n_steps", "tree_depth", "energy", "accept_prob", "divergent", "logp"):
        assert np.array_equal(res["0"][k], res["1"][k]), k
    q0 = spec.to_unconstrained(spec.default_init)
    for c in (0, 7, 1024, 2047):
        t, _ = O.sample_tuned(om, tuning["epsilon"], tuning["inv_mass"], q0, num_samples=50, seed=5 + 7919 * c,
                              cfg=O.Cfg(1, 64))
        assert np.array_equal(t["draws"], res["1"]["draws"][c]), c


@pytest.mark.parametrize("name", ["eight_schools", "sv"])
def test_sample_stream_push_equals_sample(hip, name):
    """sample_stream/4 with the per-draw notification of sampler.ex:1240-1270 as ONE launch
    (exmc_hip_stream_start / _finish): the kernel writes each finished draw into page-locked host
    memory and publishes the count with a system-scope release; the receiver is handed every draw
    while the launch is still running. Same messages as the chunked form and as sample/3; for sv
    (milliseconds per transition) the poll loop must have seen the count grow in several steps."""
    import test_golden_traces as TG
    spec = models.eight_schools() if name == "eight_schools" else models.sv(TG.GOLD["sv_returns"])
    n = 60 if name == "sv" else 200
    opts = dict(num_warmup=120, num_samples=n, seed=8, stream_push=True)
    comp = sampler.compile(spec)
    trace, stats = sampler.sample_compiled(comp, spec.default_init, dict(opts, stream_push=False))
    msgs = []
    assert sampler.sample_stream(comp, msgs.append, spec.default_init, opts) == "ok"
    assert msgs[-1] == ("exmc_done", n) and len(msgs) == n + 1
    for i, (tag, idx, point, stat) in enumerate(msgs[:-1]):
        assert tag == "exmc_sample" and idx == i + 1
        assert all(point[k] == float(trace[k][i]) for k in point)
        assert stat == stats["sample_stats"][i]
    counts = comp.last_stream_counts
    assert counts[-1] == n and all(a < b for a, b in zip(counts, counts[1:]))
    if name == "sv":
        assert len(counts) >= 5, counts            # delivered while running, not at the end
    # the resident chain goes on: a second push run continues where the first ended
    more = []
    sampler_opts = dict(opts, num_samples=10)
    L = comp.L
    view, prog = _lib.Trace(), C.POINTER(C.c_int32)()
    comp.check(L.exmc_hip_stream_start(comp.h, 10, C.byref(view), C.byref(prog)))
    with pytest.raises(Exception):                  # one run in flight at a time
        comp.check(L.exmc_hip_stream_start(comp.h, 10, C.byref(view), C.byref(prog)))
    # ... and nothing else on the handle either: the launch owns its stream, events, counters and
    # trace until stream_finish (a poller may be a thread of the caller's own)
    q1 = np.zeros((1, spec.d))
    lp1, g1 = np.zeros(1), np.zeros((1, spec.d))
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))   # noqa: E731
    assert L.exmc_hip_logp_grad_host(comp.h, dp(q1), 1, 0, dp(lp1), dp(g1)) != 0
    assert b"in flight" in L.exmc_hip_last_error()
    with pytest.raises(Exception):
        sampler.warmup(comp, spec.default_init, dict(opts, num_warmup=5))
    div = C.c_int32()
    comp.check(L.exmc_hip_stream_finish(comp.h, C.byref(div)))
    assert prog[0] == 10
    assert L.exmc_hip_logp_grad_host(comp.h, dp(q1), 1, 0, dp(lp1), dp(g1)) == 0
    t2, s2 = sampler.sample_compiled(comp, spec.default_init, dict(opts, stream_push=False, num_samples=n + 10))
    got = np.ctypeslib.as_array(C.cast(view.draws, C.POINTER(C.c_double)), shape=(10 * spec.d,)).reshape(10, spec.d)
    assert np.array_equal(got, s2["raw"]["draws"][0][n:])


def test_stream_push_always_finishes_its_run(hip):
    """A receiver that raises (or a timeout) must not leave the handle with a run in flight: the
    library side is finished before sample_stream returns, and the handle works afterwards."""
    spec = models.eight_schools()
    comp = sampler.compile(spec)
    opts = dict(num_warmup=60, num_samples=40, seed=3, stream_push=True)

    class Boom(Exception):
        pass

    def bad(msg):
        if msg[0] == "exmc_sample" and msg[1] == 3:
            raise Boom()
    with pytest.raises(Boom):
        sampler.sample_stream(comp, bad, spec.default_init, opts)
    msgs = []
    assert sampler.sample_stream(comp, msgs.append, spec.default_init, opts) == "ok"
    assert len(msgs) == 41
    # a dense mass left on the handle by an earlier run does not leak into the stream
    sampler.sample_compiled(comp, spec.default_init, dict(num_warmup=150, num_samples=5, seed=3, dense_mass=True,
                                                         lanes_per_chain=16))
    again = []
    assert sampler.sample_stream(comp, again.append, spec.default_init, opts) == "ok"
    assert again == msgs


def test_dense_mass_lane_layout_edge_cases(hip):
    """The lane-layout dense mode at its corners (logistic, 16 lanes): a warmup too short for any
    window leaves the identity covariance (mass_matrix.ex:105-109 via an empty Welford) and the
    sampler then runs the dense operations on it -- the same draws as the diagonal identity mass
    up to the order of the sums, and bit-identical to the checker; a caller-supplied (cov, chol)
    pair; max_tree_depth = 1; the kernel order differing from the flat order throughout."""
    spec = models.logistic()
    comp = sampler.compile(spec)
    om = O.model_for(spec)
    d = spec.d
    q0 = spec.to_unconstrained(spec.default_init)
    opts = dict(num_warmup=40, num_samples=12, seed=21, dense_mass=True)
    tuning = sampler.warmup(comp, spec.default_init, opts)
    st, cov, chol = O.warmup_dense(om, q0, num_warmup=40, seed=21, cfg=O.Cfg(1, 16))
    assert np.array_equal(cov, np.eye(d)) and np.array_equal(tuning["cov"], np.eye(d))
    assert st.step_size == tuning["epsilon"]
    _, _, extra = sampler.sample_compiled_tuned(comp, tuning, spec.default_init, opts, num_chains=3)
    for c in range(3):
        t, _ = O.sample_tuned_dense(om, st.step_size, cov, chol, q0, num_samples=12, seed=21 + 7919 * c, cfg=O.Cfg(1, 16))
        assert np.array_equal(t["draws"], extra["raw"]["draws"][c]) and np.array_equal(t["n_steps"], extra["raw"]["n_steps"][c])
    # a caller's pair (a random SPD matrix on the flat vector), depth-1 trees
    rng = np.random.default_rng(6)
    A = rng.normal(size=(d, d)) * 0.05
    cov2 = np.ascontiguousarray(A @ A.T + 0.02 * np.eye(d))
    chol2 = np.ascontiguousarray(np.linalg.cholesky(cov2))
    tun2 = dict(epsilon=0.05, inv_mass=cov2, cov=cov2, chol_cov=chol2)
    o2 = dict(num_samples=20, seed=4, max_tree_depth=1)
    _, _, e2 = sampler.sample_compiled_tuned(comp, tun2, spec.default_init, o2, num_chains=5)
    for c in range(5):
        t, _ = O.sample_tuned_dense(om, 0.05, cov2, chol2, q0, num_samples=20, max_tree_depth=1, seed=4 + 7919 * c,
                                    cfg=O.Cfg(1, 16))
        for k in ("draws", "n_steps", "energy", "accept_prob"):
            assert np.array_equal(t[k], e2["raw"][k][c]), (c, k)
    assert e2["raw"]["n_steps"].max() <= 1


def test_resident_chains_advance_in_pieces_with_migration(hip, monkeypatch):
    """exmc_hip_chains_init / _advance (the bench's path: device-resident traces, draw offsets) for
    sv with chain migration forced on: 2048 chains advanced by 120 and then 80 draws write the same
    trace as one launch of 200 draws and as the launches with the mode off -- a chain that moved
    in the first piece starts the second from the state its host stored."""
    import torch
    import test_golden_traces as TG
    spec = models.sv(TG.GOLD["sv_returns"])
    comp = sampler.compile(spec)
    L = comp.L
    d, Cn, S = spec.d, 2048, 200
    tuning = sampler.warmup(comp, spec.default_init, dict(num_warmup=150, seed=5))
    tun = sampler._tuning_struct(tuning, d)
    sampler._apply_mass(comp, tuning)
    iq = np.ascontiguousarray(spec.to_unconstrained(spec.default_init))
    opts = sampler._c_opts(sampler._merge_opts(dict(num_warmup=0, num_samples=S, seed=5)))
    dev = torch.device("cuda:0")
    out = {}
    for name, mig, pieces in (("whole_off", "0", [(200, 0)]), ("whole_on", "1", [(200, 0)]), ("pieces_on", "1", [(120, 0), (80, 120)])):
        monkeypatch.setenv("EXMC_HIP_MIGRATE", mig)
        draws = torch.zeros((S, d, Cn), dtype=torch.float64, device=dev)
        nst = torch.zeros((S, Cn), dtype=torch.int32, device=dev)
        tr = _lib.Trace(draws.data_ptr(), None, None, nst.data_ptr(), None, None, None)
        lf, dv = C.c_int64(), C.c_int32()
        comp.check(L.exmc_hip_chains_init(comp.h, C.byref(tun), _dp(iq), Cn, 0, Cn, opts))
        total = 0
        for n, off in pieces:
            comp.check(L.exmc_hip_chains_advance(comp.h, n, off, tr, C.byref(lf), C.byref(dv)))
            total += lf.value
        torch.cuda.synchronize()
        out[name] = (draws.cpu().numpy(), nst.cpu().numpy(), total)
    for name in ("whole_on", "pieces_on"):
        assert np.array_equal(out[name][0], out["whole_off"][0]), name
        assert np.array_equal(out[name][1], out["whole_off"][1]) and out[name][2] == out["whole_off"][2], name
    assert out["whole_off"][2] == int(out["whole_off"][1].sum())


def test_radon_refuses_more_observations_than_its_lane_layout_holds(hip):
    """The radon kind's 64-lane layout gives every lane 16 observation slots: 1024 observations are
    the most it takes, and the library says so instead of walking off its strips. models.radon does
    not hand it such data: above 1024 observations it returns the generated form of the same model
    (codegen.radon_ir, a lane layout of its own), which samples bit for bit like its CPU text."""
    import gen_checker as GC
    rng = np.random.default_rng(3)
    J, N = 85, 1100
    sizes = np.full(J, N // J)
    sizes[: N - sizes.sum()] += 1
    start = np.concatenate([[0], np.cumsum(sizes)])
    data = (rng.normal(size=J), start, (rng.uniform(size=N) < 0.2).astype(float), rng.normal(size=N))
    with pytest.raises(_lib.ExmcHipError, match="1024 observations"):
        sampler.compile(models.radon(data, builtin=True))
    spec = models.radon(data)
    assert hasattr(spec, "gen") and spec.d == 90
    comp = sampler.compile(spec)
    q = 0.2 * rng.normal(size=(6, spec.d))
    lp = np.zeros(6)
    g = np.zeros((6, spec.d))
    comp.check(comp.L.exmc_hip_logp_grad_host(comp.h, sampler._dp(np.ascontiguousarray(q)), 6, 64, sampler._dp(lp), sampler._dp(g)))
    for c in range(6):
        lp_c, g_c = GC.logp_grad(spec.gen, q[c], lanes=64)
        assert lp[c] == lp_c and np.array_equal(g[c], g_c)
