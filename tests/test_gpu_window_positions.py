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


@pytest.mark.parametrize("name,factory,lane_list", _kinds(), ids=lambda x: x if isinstance(x, str) else "")
def test_multi_step_through_exploding_trajectories_bit_exact(hip, name, factory, lane_list):
    """multi_step_fn (the batched leapfrog, B2) along trajectories that blow up: a step size far too large
    and momenta from 1 to 1e150, so that within a dozen steps the positions pass through every magnitude up to
    infinity and NaN -- every step of every chain against the checker."""
    spec = factory()
    comp = sampler.compile(spec)
    om = O.model_for(spec)
    rng = np.random.default_rng(31)
    C_, n, d = 48, 12, spec.d
    q0 = spec.to_unconstrained(spec.default_init)
    q = np.ascontiguousarray(q0[None, :] + rng.normal(size=(C_, d)) * 0.3)
    scale = 10.0 ** rng.choice([0, 0, 1, 2, 3, 5, 8, 20, 80, 150], size=(C_, 1))
    scale[::4] = 1.0                                     # every fourth chain stays tame
    p = np.ascontiguousarray(rng.normal(size=(C_, d)) * scale)
    im = np.ascontiguousarray(rng.uniform(0.5, 2.0, size=d))
    for lanes in lane_list:
        cfg = O.Cfg(1, lanes)
        g = np.array([om.logp_grad(q[c], cfg)[1] for c in range(C_)])
        for eps in (0.05, 1.5):
            aq = np.zeros((C_, n, d)); ap = np.zeros((C_, n, d)); ag = np.zeros((C_, n, d))
            alp = np.zeros((C_, n))
            _lib.check(hip.exmc_hip_multi_step_host(comp.h, _dp(q), _dp(p), _dp(g), eps, _dp(im), n, C_,
                                                    lanes, _dp(aq), _dp(ap), _dp(alp), _dp(ag)))
            bad = 0
            for c in range(C_):
                oq, op, olp, og = om.multi_step(q[c], p[c], g[c], eps, im, n, cfg)
                assert np.array_equal(oq, aq[c], equal_nan=True), (name, lanes, eps, c)
                assert np.array_equal(op, ap[c], equal_nan=True), (name, lanes, eps, c)
                assert np.array_equal(olp, alp[c], equal_nan=True), (name, lanes, eps, c)
                assert np.array_equal(og, ag[c], equal_nan=True), (name, lanes, eps, c)
                with np.errstate(invalid="ignore"):
                    bad += (not np.all(np.isfinite(olp))) or bool(np.nanmax(np.abs(oq)) > 1e50)
            if eps < 1.0:
                assert bad < C_                           # at the small step size not every chain blew up
        assert bad > 0                                    # (at the large one some did)


def _sample_kinds():
    import test_golden_traces as TG
    return [("simple", models.simple, 1), ("eight_schools", models.eight_schools, 16),
            ("sv", lambda: models.sv(TG.GOLD["sv_returns"]), 64), ("logistic", models.logistic, 16),
            ("radon", models.radon, 64)]


@pytest.mark.parametrize("spread", [6.0, 40.0])
@pytest.mark.parametrize("name,factory,lanes", _sample_kinds(), ids=lambda x: x if isinstance(x, str) else "")
def test_sample_from_a_hostile_start_bit_exact(hip, name, factory, lanes, spread):
    """Sampler.sample/3 (step-size search, warmup, draws) started far from the mode -- every coordinate a few
    (6) or many (40) units away on the unconstrained scale, scales from e^-100 to e^+100: the first trees
    diverge at their first leaf or run into the clamps, the step-size search halves dozens of times, and the
    adaptation has to walk back. The whole run against the checker, bit for bit."""
    spec = factory()
    comp = sampler.compile(spec)
    om = O.model_for(spec)
    rng = np.random.default_rng(37)
    q_far = spec.to_unconstrained(spec.default_init) + rng.normal(size=spec.d) * spread
    init = {n: float(np.exp(q_far[i])) if spec.transforms.get(n) == "log" else float(q_far[i])
            for i, n in enumerate(spec.var_names)}
    q0 = spec.to_unconstrained(init)
    assert np.all(np.isfinite(q0))
    nw, ns = (40, 12) if name == "sv" else (60, 25)
    opts = dict(num_warmup=nw, num_samples=ns, seed=5, lanes_per_chain=lanes, max_tree_depth=8)
    _, stats = sampler.sample_compiled(comp, init, opts)
    t, st = O.sample(om, init_q=q0, num_warmup=nw, num_samples=ns, seed=5, cfg=O.Cfg(1, lanes), max_tree_depth=8)
    assert stats["step_size"] == st.step_size or (np.isnan(stats["step_size"]) and np.isnan(st.step_size))
    raw = stats["raw"]
    for k in ("tree_depth", "n_steps", "divergent", "draws", "energy"):
        assert np.array_equal(raw[k][0], t[k], equal_nan=True), (name, spread, k)


@pytest.mark.parametrize("spread", [6.0, 40.0])
@pytest.mark.parametrize("name,lanes", [("eight_schools", 16), ("eight_schools", 1), ("simple", 1)])
def test_dense_mass_warmup_from_a_hostile_start_bit_exact(hip, name, lanes, spread):
    """opts[:dense_mass] from a start far from the mode: the dense Welford windows collect a chain that is still
    walking in (near-singular covariances before the shrinkage), the Cholesky factor and the step size must be
    the checker's bits all the same, and so must draws under that mass."""
    spec = models.eight_schools() if name == "eight_schools" else models.simple()
    om = O.eight_schools() if name == "eight_schools" else O.simple()
    comp = sampler.compile(spec)
    rng = np.random.default_rng(47)
    q_far = spec.to_unconstrained(spec.default_init) + rng.normal(size=spec.d) * spread
    init = {n: float(np.exp(q_far[i])) if spec.transforms.get(n) == "log" else float(q_far[i])
            for i, n in enumerate(spec.var_names)}
    q0 = spec.to_unconstrained(init)
    opts = dict(num_warmup=320, num_samples=20, seed=19, lanes_per_chain=lanes, dense_mass=True)
    tuning = sampler.warmup(comp, init, opts)
    st, cov, chol = O.warmup_dense(om, q0, num_warmup=320, seed=19, cfg=O.Cfg(1, lanes))
    assert st.step_size == tuning["epsilon"] or (np.isnan(st.step_size) and np.isnan(tuning["epsilon"]))
    assert np.array_equal(cov, tuning["cov"], equal_nan=True)
    assert np.array_equal(chol, tuning["chol_cov"], equal_nan=True)
    _, _, extra = sampler.sample_compiled_tuned(comp, tuning, init, opts, num_chains=3)
    for c in range(3):
        t, _ = O.sample_tuned_dense(om, st.step_size, cov, chol, q0, num_samples=20, seed=19 + 7919 * c,
                                    cfg=O.Cfg(1, lanes))
        for k in ("draws", "tree_depth", "n_steps", "divergent", "energy", "accept_prob"):
            assert np.array_equal(t[k], extra["raw"][k][c], equal_nan=True), (name, lanes, spread, c, k)


@pytest.mark.parametrize("name,factory,lanes", _sample_kinds(), ids=lambda x: x if isinstance(x, str) else "")
def test_chain_batches_from_a_hostile_start_bit_exact(hip, name, factory, lanes):
    """sample_chains (vectorized: the shared warmup of chain 0, then the batch seeded seed + 7919 i) from a start 6
    units from the mode on the unconstrained scale, with a SHORT warmup -- the batch starts its draws with a step size
    and a mass that are not yet right for where the chains are, so the chains of a wavefront diverge, turn and run
    to the depth cap side by side. Tuning and every per-draw output against the checker."""
    spec = factory()
    comp = sampler.compile(spec)
    om = O.model_for(spec)
    rng = np.random.default_rng(53)
    q_far = spec.to_unconstrained(spec.default_init) + rng.normal(size=spec.d) * 6.0
    init = {n: float(np.exp(q_far[i])) if spec.transforms.get(n) == "log" else float(q_far[i])
            for i, n in enumerate(spec.var_names)}
    q0 = spec.to_unconstrained(init)
    nc, nw, ns = (6, 25, 8) if name == "sv" else (21, 30, 15)
    opts = dict(num_warmup=nw, num_samples=ns, seed=61, lanes_per_chain=lanes, warmup_lanes=lanes, max_tree_depth=7)
    tuning = sampler.warmup(comp, init, opts)
    t, st = O.sample_chains(om, nc, init_q=q0, num_warmup=nw, num_samples=ns, seed=61, cfg=O.Cfg(1, lanes),
                            max_tree_depth=7, n_threads=8)
    assert st.step_size == tuning["epsilon"]
    assert np.array_equal(np.array(st.inv_mass[:spec.d]), tuning["inv_mass"])
    _, _, extra = sampler.sample_compiled_tuned(comp, tuning, init, opts, num_chains=nc)
    raw = extra["raw"]
    for k in ("tree_depth", "n_steps", "divergent", "draws", "logp", "accept_prob", "energy"):
        assert np.array_equal(t[k], raw[k], equal_nan=True), (name, k)
