"""GPU parity for the lane layout of generated models (exmc_amd/codegen_lanes.py): stochastic
volatility, radon and the logistic regression compiled from Builder node lists, through the C ABI
of their plug-in libraries, against the CPU oracle running the same generated text on virtual
lanes (tests/gen_checker.py) -- bit for bit -- and against the hand-written kinds of the oracle
(an independent restatement of the same densities) to 1e-12."""
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


_cache = {}


def _compiled(which, hip):
    if which not in _cache:
        if which in ("zoo16", "walk16"):
            ir = GM.zoo_ir() if which == "zoo16" else GM.walk_ir()
            init = GM.ZOO_INIT if which == "zoo16" else GM.WALK_INIT
            spec = cg.compile_ir(ir, name=which, default_init=init, lanes=16)
            hand = None
        else:
            ir, ncp, hand, lanes = GM.baseline_pair(which)
            spec = cg.compile_ir(ir, ncp=ncp, name="gen_" + which, default_init=hand.default_init, lanes=lanes)
        lanes = spec.gen.lanes
        _cache[which] = (spec, sampler.compile(spec), GC.model(spec.gen, lanes), hand, lanes)
    return _cache[which]


@pytest.mark.parametrize("which", ["sv", "radon", "logistic", "zoo16", "walk16"])
def test_lane_layout_logp_grad_bit_exact(which, hip):
    spec, comp, om, hand, lanes = _compiled(which, hip)
    assert comp.default_lanes == lanes
    rng = np.random.default_rng(11)
    n = 200
    scale = 0.1 if which == "sv" else 0.4
    q0 = spec.to_unconstrained(spec.default_init)
    q = np.ascontiguousarray(q0[None, :] + rng.normal(size=(n, spec.d)) * scale)
    q[0] = q0
    q[1, :] = 250.0      # beyond every clamp
    q[2, :] = -250.0
    q[3, :] = 0.0
    q[4] = q0 + rng.normal(size=spec.d) * 8.0
    lp = np.zeros(n)
    g = np.zeros((n, spec.d))
    comp.check(comp.L.exmc_hip_logp_grad_host(comp.h, _dp(q), n, lanes, _dp(lp), _dp(g)))
    cfg = O.Cfg(1, lanes)
    for c in range(n):
        olp, og = om.logp_grad(q[c], cfg)
        assert olp == lp[c] or (np.isnan(olp) and np.isnan(lp[c])), (which, c, olp, lp[c])
        assert np.array_equal(og, g[c], equal_nan=True), (which, c)
    if hand is not None:
        hm = O.model_for(hand)
        idx = GM.to_spec_order(spec.gen, hand)
        inv = np.argsort(idx)
        for c in range(5, 60):
            hlp, hg = hm.logp_grad(q[c][inv], O.Cfg(0, 1))
            assert abs(hlp - lp[c]) <= 1e-12 * max(1.0, abs(hlp)), (which, c)
            assert np.all(np.abs(hg[idx] - g[c]) <= 2e-12 * max(1.0, np.max(np.abs(hg)))), (which, c)


@pytest.mark.parametrize("which", ["sv", "radon", "logistic", "walk16"])
def test_lane_layout_sample_bit_exact(which, hip):
    """Sampler.sample/3 end to end (on-device adaptation + sampling) on the generated lane kernels."""
    spec, comp, om, _, lanes = _compiled(which, hip)
    nw, ns = (120, 60) if which == "sv" else (150, 100)
    opts = dict(num_warmup=nw, num_samples=ns, seed=17, lanes_per_chain=lanes)
    trace, stats = sampler.sample_compiled(comp, spec.default_init, opts)
    q0 = spec.to_unconstrained(spec.default_init)
    t, st = O.sample(om, init_q=q0, num_warmup=nw, num_samples=ns, seed=17, cfg=O.Cfg(1, lanes))
    assert stats["step_size"] == st.step_size
    raw = stats["raw"]
    assert np.array_equal(raw["tree_depth"][0], t["tree_depth"])
    assert np.array_equal(raw["n_steps"][0], t["n_steps"])
    assert np.array_equal(raw["divergent"][0], t["divergent"])
    assert np.array_equal(raw["draws"][0], t["draws"])
    assert np.array_equal(raw["energy"][0], t["energy"])


@pytest.mark.parametrize("which", ["sv", "logistic"])
def test_lane_layout_chain_batches_bit_exact(which, hip):
    """A batch of chains after the shared warmup (logistic: four chains per wavefront, each with
    its own LDS strip; sv: one chain per wavefront)."""
    spec, comp, om, _, lanes = _compiled(which, hip)
    nc, nw, ns = (24, 100, 30) if which == "sv" else (70, 120, 60)
    # (warmup in the sampling layout, as the checker's sample_chains does; the default one-chain
    # form of a 16-lane layout has its own test below)
    opts = dict(num_warmup=nw, num_samples=ns, seed=42, init_values=spec.default_init, warmup_lanes=lanes)
    _, stats = sampler.sample_chains_compiled(comp, nc, opts)
    q0 = spec.to_unconstrained(spec.default_init)
    t, st = O.sample_chains(om, nc, init_q=q0, num_warmup=nw, num_samples=ns, seed=42,
                            cfg=O.Cfg(1, lanes), n_threads=8)
    raw = stats[0]["extra"]["raw"]
    assert stats[0]["step_size"] == st.step_size
    for k in ("draws", "logp", "tree_depth", "n_steps", "divergent", "accept_prob", "energy"):
        assert np.array_equal(raw[k], t[k]), k


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_random_models_in_the_lane_layout_bit_exact(seed, hip):
    """Randomly drawn hierarchical models (tests/test_codegen_lanes.py _random_big_ir: one to three
    plates of random size and likelihood, random transforms, sometimes a random walk) at 16, 32 and
    64 lanes per chain: log-density, gradient and a short sample/3 equal the generated text on the
    CPU bit for bit."""
    import test_codegen_lanes as TL
    ir, rng = TL._random_big_ir(seed)
    lanes = (16, 32, 64)[seed % 3]
    gen = cg.generate(ir, lanes=lanes)
    so = cg.build_plugin(gen)
    init = None
    spec = cg.GeneratedSpec(gen, so, name="gen_random_lanes_%d" % seed, default_init=init)
    comp = sampler.compile(spec)
    om = GC.model(gen, lanes)
    n = 64
    q = np.ascontiguousarray(rng.normal(size=(n, gen.d)) * 0.7)
    q[1] = 40.0
    q[2] = -40.0
    lp, g = np.zeros(n), np.zeros((n, gen.d))
    comp.check(comp.L.exmc_hip_logp_grad_host(comp.h, _dp(q), n, lanes, _dp(lp), _dp(g)))
    cfg = O.Cfg(1, lanes)
    for c in range(n):
        olp, og = om.logp_grad(q[c], cfg)
        assert olp == lp[c] or (np.isnan(olp) and np.isnan(lp[c])), (seed, c, olp, lp[c])
        assert np.array_equal(og, g[c], equal_nan=True), (seed, c)
    q0 = np.ascontiguousarray(rng.normal(size=gen.d) * 0.1)
    opts = dict(num_warmup=60, num_samples=25, seed=5, lanes_per_chain=lanes)
    tun = sampler._lib.Tuning()
    tr, t = sampler._host_trace(1, 25, gen.d)
    dv = C.c_int32()
    comp.check(comp.L.exmc_hip_sample_host(comp.h, _dp(q0), sampler._c_opts(sampler._merge_opts(opts)), t,
                                          C.byref(tun), C.byref(dv)))
    ot, ost = O.sample(om, init_q=q0, num_warmup=60, num_samples=25, seed=5, cfg=cfg)
    assert tun.epsilon == ost.step_size
    assert np.array_equal(tr["draws"][0], ot["draws"]) and np.array_equal(tr["n_steps"][0], ot["n_steps"])


def test_generated_sv_two_waves_per_simd_bit_exact(hip, monkeypatch):
    """compile_ir(..., waves_per_simd=2) for a 64-lane model: the sampling kernel is capped at 256
    registers and runs with what the hand-written sv kind runs with -- cross-row sums through
    ds_bpermute, chain migration, the time-sliced issue priority. None of it may change a bit: 2048
    chains (two per SIMD, so all of it is active) equal the one-wave-per-SIMD build and, for a few
    chains, the checker."""
    ir, ncp, hand, lanes = GM.baseline_pair("sv")
    spec1, comp1, om, _, _ = _compiled("sv", hip)
    spec2 = cg.compile_ir(ir, ncp=ncp, name="gen_sv_wps2", default_init=hand.default_init, lanes=lanes,
                          waves_per_simd=2)
    comp2 = sampler.compile(spec2)
    opts = dict(num_warmup=100, num_samples=40, seed=9)
    tuning = sampler.warmup(comp1, spec1.default_init, opts)
    monkeypatch.setenv("EXMC_HIP_MIGRATE", "1")     # (default: launches of 100 draws and more)
    res = []
    for comp, spec in ((comp1, spec1), (comp2, spec2)):
        _, _, extra = sampler.sample_compiled_tuned(comp, tuning, spec.default_init, opts, num_chains=2048)
        res.append(extra["raw"])
    for k in ("draws", "n_steps", "tree_depth", "energy", "accept_prob", "divergent", "logp"):
        assert np.array_equal(res[0][k], res[1][k]), k
    q0 = spec1.to_unconstrained(spec1.default_init)
    for c in (0, 1500):
        t, _ = O.sample_tuned(om, tuning["epsilon"], tuning["inv_mass"], q0, num_samples=40, seed=9 + 7919 * c,
                              cfg=O.Cfg(1, lanes))
        assert np.array_equal(t["draws"], res[1]["draws"][c]), c


@pytest.mark.parametrize("which", ["logistic", "walk16"])
def test_one_chain_warmup_form_bit_exact(which, hip):
    """The shared warmup of a generated layout with fewer than 64 lanes per chain runs the chain on
    the whole wavefront by default (exmc_hip_model_default_warmup_lanes = 64: CustomSplit, the units
    of every family over the four lane groups, their sums added in group order). Its tuning equals
    the checker's in that form (gen_checker.model(..., wave_split=True)), differs from the sampling
    layout's own warmup, and the chains sampled from it (in the sampling layout) equal the checker."""
    spec, comp, om, _, lanes = _compiled(which, hip)
    assert lanes == 16 and comp.default_warmup_lanes == 64
    oms = GC.model(spec.gen, lanes, wave_split=True)
    opts = dict(num_warmup=150, num_samples=40, seed=23)
    tuning = sampler.warmup(comp, spec.default_init, opts)                       # default: the split form
    q0 = spec.to_unconstrained(spec.default_init)
    st = O.warmup(oms, q0, num_warmup=150, seed=23, cfg=O.Cfg(1, lanes))
    assert tuning["epsilon"] == st.step_size
    assert np.array_equal(tuning["inv_mass"], np.array(st.inv_mass[:spec.d]))
    plain = sampler.warmup(comp, spec.default_init, dict(opts, warmup_lanes=lanes))
    st_plain = O.warmup(om, q0, num_warmup=150, seed=23, cfg=O.Cfg(1, lanes))
    assert plain["epsilon"] == st_plain.step_size
    assert np.array_equal(plain["inv_mass"], np.array(st_plain.inv_mass[:spec.d]))
    if which == "logistic":     # (walk16's families have one slot per lane: the other groups add zeros)
        assert plain["epsilon"] != tuning["epsilon"] or not np.array_equal(plain["inv_mass"], tuning["inv_mass"])
    _, _, extra = sampler.sample_compiled_tuned(comp, tuning, spec.default_init, opts, num_chains=9)
    for c in (0, 8):
        t, _ = O.sample_tuned(om, tuning["epsilon"], tuning["inv_mass"], q0, num_samples=40, seed=23 + 7919 * c,
                              cfg=O.Cfg(1, lanes))
        assert np.array_equal(t["draws"], extra["raw"]["draws"][c]), c


def test_one_chain_warmup_form_at_32_lanes_bit_exact(hip):
    """The same with two lane groups per wavefront: sv compiled at 32 lanes per chain (four
    dimensions per lane) warms up on the whole wavefront by default and equals the checker's
    wave-split form; the tuning then drives chains in the 32-lane sampling layout."""
    ir, ncp, hand, _ = GM.baseline_pair("sv")
    spec = cg.compile_ir(ir, ncp=ncp, name="gen_sv_32", default_init=hand.default_init, lanes=32)
    comp = sampler.compile(spec)
    assert comp.default_lanes == 32 and comp.default_warmup_lanes == 64
    om, oms = GC.model(spec.gen, 32), GC.model(spec.gen, 32, wave_split=True)
    opts = dict(num_warmup=80, num_samples=20, seed=31)
    tuning = sampler.warmup(comp, spec.default_init, opts)
    q0 = spec.to_unconstrained(spec.default_init)
    st = O.warmup(oms, q0, num_warmup=80, seed=31, cfg=O.Cfg(1, 32))
    assert tuning["epsilon"] == st.step_size
    assert np.array_equal(tuning["inv_mass"], np.array(st.inv_mass[:spec.d]))
    _, _, extra = sampler.sample_compiled_tuned(comp, tuning, spec.default_init, opts, num_chains=5)
    t, _ = O.sample_tuned(om, tuning["epsilon"], tuning["inv_mass"], q0, num_samples=20, seed=31 + 7919 * 4,
                          cfg=O.Cfg(1, 32))
    assert np.array_equal(t["draws"], extra["raw"]["draws"][4])
