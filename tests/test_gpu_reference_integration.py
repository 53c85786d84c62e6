"""The reference's own end-to-end tests (test/integration_test.exs), model by model, on the GPU: the Builder
model as IR nodes -> the reference's rewrite passes (default transforms, non-centred rewrite) -> generated HIP ->
plug-in -> exmc_amd.sampler.sample / sample_chains with the test's options and seed. The acceptance bands are
the reference's literals (path:line at each test). The random stream is the checker-verified restatement of OTP's
exsss, so these are not the reference's draws -- the bands are what the reference itself asserts of ANY correct
sampler run."""
import numpy as np
import pytest

from exmc_amd import codegen as cg, diagnostics, sampler

pytestmark = pytest.mark.gpu


def _spec(ir, name, ncp=True):
    return cg.compile_ir(ir, ncp=ncp, name=name, rewrite_passes=True)


def _normal_normal(prior_sigma, obs, lik_sigma=1.0):
    ir = cg.IR()
    ir.rv("mu", "normal", dict(mu=0.0, sigma=prior_sigma))
    ir.rv("x", "normal", dict(mu="mu", sigma=lik_sigma))
    ir.obs("x_obs", "x", obs)
    return ir


def _summary(x):
    """Diagnostics.summary (diagnostics.ex:14-34): mean, divisor-n std, linear quantiles."""
    x = np.asarray(x, dtype=np.float64)
    q = np.quantile(x, [0.05, 0.25, 0.5, 0.75, 0.95], method="linear")
    return dict(mean=x.mean(), std=np.sqrt(((x - x.mean()) ** 2).mean()), q=q)


def _rhat(comp, chains):
    ch = np.stack([np.asarray(c, dtype=np.float64) for c in chains])          # [C][S]
    tr = np.repeat(ch[:, :, None], comp.d, axis=2)
    return diagnostics.rhat(comp, tr)[0]


def test_conjugate_normal_normal(hip):
    """integration_test.exs:13-33: mu ~ N(0, 10), x ~ N(mu, 1), x = 5: mean 4.95 +- 0.5, std sqrt(0.99) +- 0.5."""
    spec = _spec(_normal_normal(10.0, 5.0), "ri_nn")
    trace, stats = sampler.sample(spec, {}, dict(num_warmup=300, num_samples=500, seed=42))
    s = _summary(trace["mu"])
    assert abs(s["mean"] - 4.95) <= 0.5 and abs(s["std"] - np.sqrt(0.99)) <= 0.5
    assert stats["divergences"] < 20


def test_multi_chain_rhat_and_ess(hip):
    """integration_test.exs:35-56: two chains, R-hat 1.0 +- 0.2, ESS of the combined draws > 50."""
    spec = _spec(_normal_normal(5.0, 3.0), "ri_mc")
    comp = sampler.compile(spec)
    try:
        traces, _ = sampler.sample_chains(comp, 2, dict(num_warmup=200, num_samples=300, seed=7))
        chains = [t["mu"] for t in traces]
        assert abs(_rhat(comp, chains) - 1.0) <= 0.2
        combined = np.concatenate(chains)
        ess = diagnostics.ess(comp, np.repeat(combined[None, :, None], comp.d, axis=2))[0, 0]
        assert ess > 50
    finally:
        comp.close()


def test_gamma_and_exponential_priors_respect_their_support(hip):
    """integration_test.exs:58-76 (Gamma(2, 1): all draws positive, mean 2.0 +- 1.0) and :78-93 (Exponential(2):
    positive, mean 0.5 +- 0.3)."""
    ir = cg.IR().rv("alpha", "gamma", dict(alpha=2.0, beta=1.0))
    trace, _ = sampler.sample(_spec(ir, "ri_gamma"), {}, dict(num_warmup=200, num_samples=200, seed=99))
    assert np.all(trace["alpha"] > 0.0) and abs(trace["alpha"].mean() - 2.0) <= 1.0
    ir = cg.IR().rv("rate", "exponential", {"lambda": 2.0})
    trace, _ = sampler.sample(_spec(ir, "ri_exp"), {}, dict(num_warmup=200, num_samples=300, seed=77))
    assert np.all(trace["rate"] > 0.0) and abs(trace["rate"].mean() - 0.5) <= 0.3


def test_hierarchical_posterior_shift(hip):
    """integration_test.exs:95-126: parent_mu ~ N(0, 5), child ~ N(parent_mu, 2), child = 4: mean 3.45 +- 1.0,
    ordered quantiles, ESS > 30, divergences < 50."""
    ir = cg.IR()
    ir.rv("parent_mu", "normal", dict(mu=0.0, sigma=5.0))
    ir.rv("child", "normal", dict(mu="parent_mu", sigma=2.0))
    ir.obs("child_obs", "child", 4.0)
    spec = _spec(ir, "ri_hier")
    comp = sampler.compile(spec)
    try:
        trace, stats = sampler.sample(comp, {}, dict(num_warmup=300, num_samples=400, seed=55))
        s = _summary(trace["parent_mu"])
        assert abs(s["mean"] - 3.45) <= 1.0 and np.all(np.diff(s["q"]) > 0)
        assert diagnostics.ess(comp, stats["raw"]["draws"])[0, 0] > 30
        assert stats["divergences"] < 50
    finally:
        comp.close()


def test_sample_stats_consistency(hip):
    """integration_test.exs:159-198: lengths, tree depth and acceptance bounds, divergence count."""
    ir = cg.IR().rv("x", "normal", dict(mu=0.0, sigma=1.0))
    _, stats = sampler.sample(_spec(ir, "ri_stats"), {}, dict(num_warmup=100, num_samples=100, max_tree_depth=10, seed=33))
    ss = list(stats["sample_stats"])
    assert len(ss) == 100
    assert all(0 <= s["tree_depth"] <= 10 and s["n_steps"] >= 1 and 0.0 <= s["accept_prob"] <= 1.0 for s in ss)
    assert all(isinstance(s["divergent"], bool) and isinstance(s["n_steps"], int) for s in ss)
    assert stats["divergences"] >= sum(s["divergent"] for s in ss)


def _three_obs(vector):
    ir = cg.IR().rv("mu", "normal", dict(mu=0.0, sigma=10.0))
    if vector:
        ir.rv("x", "normal", dict(mu="mu", sigma=1.0))
        ir.obs("x_obs", "x", [4.0, 3.8, 4.2])
    else:
        for k, v in enumerate((4.0, 3.8, 4.2), start=1):
            ir.rv("x%d" % k, "normal", dict(mu="mu", sigma=1.0))
            ir.obs("x%d_obs" % k, "x%d" % k, v)
    return ir


def test_more_observations_narrow_the_posterior_and_vector_obs_equal_scalar_obs(hip):
    """integration_test.exs:200-231 (std with three observations < std with one) and :611-646 (a vector obs node
    gives the posterior of three scalar ones: means within 0.5, stds within 0.3)."""
    t1, _ = sampler.sample(_spec(_normal_normal(10.0, 4.0), "ri_one"), {}, dict(num_warmup=200, num_samples=300, seed=42))
    t3, _ = sampler.sample(_spec(_three_obs(False), "ri_three"), {}, dict(num_warmup=200, num_samples=300, seed=42))
    assert _summary(t3["mu"])["std"] < _summary(t1["mu"])["std"]
    ts, _ = sampler.sample(_spec(_three_obs(False), "ri_three"), {}, dict(num_warmup=300, num_samples=500, seed=42))
    tv, _ = sampler.sample(_spec(_three_obs(True), "ri_vec"), {}, dict(num_warmup=300, num_samples=500, seed=42))
    a, b = _summary(ts["mu"]), _summary(tv["mu"])
    assert abs(a["mean"] - b["mean"]) <= 0.5 and abs(a["std"] - b["std"]) <= 0.3
    assert abs(b["mean"] - 3.99) <= 0.5                     # the analytic value the test quotes


def test_vector_obs_with_five_observations(hip):
    """integration_test.exs:648-669: mu ~ N(0, 5), x ~ N(mu, 1), x = [1..5]: mean 3.0 +- 0.5, std < 1, divergences < 50."""
    ir = cg.IR().rv("mu", "normal", dict(mu=0.0, sigma=5.0))
    ir.rv("x", "normal", dict(mu="mu", sigma=1.0))
    ir.obs("x_obs", "x", [1.0, 2.0, 3.0, 4.0, 5.0])
    trace, stats = sampler.sample(_spec(ir, "ri_vec5"), {}, dict(num_warmup=300, num_samples=500, seed=42))
    s = _summary(trace["mu"])
    assert abs(s["mean"] - 3.0) <= 0.5 and s["std"] < 1.0 and stats["divergences"] < 50


def test_beta_and_student_t_priors(hip):
    """integration_test.exs:233-259 (Beta(2, 5) from p = 0.2: draws in (0, 1), mean 2/7 +- 0.15) and :261-281
    (StudentT(4, 3, 1): mean 3.0 +- 1.5)."""
    ir = cg.IR().rv("p", "beta", dict(alpha=2.0, beta=5.0))
    trace, _ = sampler.sample(_spec(ir, "ri_beta"), {"p": 0.2}, dict(num_warmup=300, num_samples=400, seed=88))
    assert np.all((trace["p"] > 0.0) & (trace["p"] < 1.0)) and abs(trace["p"].mean() - 2.0 / 7.0) <= 0.15
    ir = cg.IR().rv("x", "student_t", dict(df=4.0, loc=3.0, scale=1.0))
    trace, _ = sampler.sample(_spec(ir, "ri_t"), {}, dict(num_warmup=300, num_samples=400, seed=66))
    assert abs(trace["x"].mean() - 3.0) <= 1.5


def _scale_model():
    ir = cg.IR().rv("sigma", "exponential", {"lambda": 1.0})
    ir.rv("child", "normal", dict(mu=0.0, sigma="sigma"))
    ir.obs("child_obs", "child", 2.0)
    return ir


def test_constrained_parent_and_init_values_for_every_chain(hip):
    """integration_test.exs:283-309 (sigma ~ Exp(1), child ~ N(0, sigma), child = 2, from sigma = 2: positive draws,
    0.5 < mean < 10) and :739-773 (two chains from the same init values: positive draws, R-hat 1.0 +- 0.3, a
    positive step size in every chain's stats)."""
    spec = _spec(_scale_model(), "ri_scale")
    trace, _ = sampler.sample(spec, {"sigma": 2.0}, dict(num_warmup=500, num_samples=500, seed=44))
    assert np.all(trace["sigma"] > 0.0) and 0.5 < trace["sigma"].mean() < 10.0
    comp = sampler.compile(spec)
    try:
        traces, stats = sampler.sample_chains(comp, 2, dict(num_warmup=300, num_samples=300, seed=44,
                                                            init_values={"sigma": 2.0}))
        assert len(traces) == 2 and all(np.all(t["sigma"] > 0.0) for t in traces)
        assert abs(_rhat(comp, [t["sigma"] for t in traces]) - 1.0) <= 0.3
        assert all(isinstance(s["step_size"], float) and s["step_size"] > 0.0 for s in stats)
    finally:
        comp.close()


def test_five_parameter_hierarchical_model(hip):
    """integration_test.exs:376-444: five free parameters, the group means non-centred by the rewrite; the
    constrained scales positive, 1 < mean(alpha) < 9, 2 < mean(beta) < 14, a summary with positive spread for all."""
    ir = cg.IR()
    ir.rv("mu_global", "normal", dict(mu=0.0, sigma=10.0))
    ir.rv("sigma_global", "exponential", {"lambda": 1.0})
    ir.rv("alpha", "normal", dict(mu="mu_global", sigma="sigma_global"))
    ir.rv("beta", "normal", dict(mu="mu_global", sigma="sigma_global"))
    ir.rv("sigma_obs", "exponential", {"lambda": 2.0})
    for k, (tgt, v) in enumerate((("alpha", 4.0), ("alpha", 5.0), ("beta", 8.0)), start=1):
        ir.rv("y%d" % k, "normal", dict(mu=tgt, sigma="sigma_obs"))
        ir.obs("y%d_obs" % k, "y%d" % k, v)
    spec = _spec(ir, "ri_five")
    assert sorted(spec.gen.ncp_info) == ["alpha", "beta"]
    init = dict(mu_global=5.0, sigma_global=2.0, alpha=4.5, beta=8.0, sigma_obs=1.0)
    trace, _ = sampler.sample(spec, init, dict(num_warmup=500, num_samples=500, seed=42))
    assert len(trace) == 5
    assert np.all(trace["sigma_global"] > 0.0) and np.all(trace["sigma_obs"] > 0.0)
    assert 1.0 < trace["alpha"].mean() < 9.0 and 2.0 < trace["beta"].mean() < 14.0
    assert all(np.isfinite(v.mean()) and v.std() > 0.0 for v in trace.values())


def test_non_centred_rewrite_and_reconstruction(hip):
    """integration_test.exs:446-490: alpha ~ N(mu, sigma) becomes N(0, 1) with ncp_info[alpha] = {mu, sigma}, the trace
    carries alpha = mu + sigma z again: 0 < mean(alpha) < 8, sigma positive, divergences < 100; :492-513: the compiled
    log-density at (mu, log sigma, z) = (1, 0, 0.5) is a negative number above -100."""
    ir = cg.IR()
    ir.rv("mu", "normal", dict(mu=0.0, sigma=5.0))
    ir.rv("sigma", "exponential", {"lambda": 1.0})
    ir.rv("alpha", "normal", dict(mu="mu", sigma="sigma"))
    ir.rv("y", "normal", dict(mu="alpha", sigma=1.0))
    ir.obs("y_obs", "y", 3.0)
    spec = _spec(ir, "ri_ncp")
    assert spec.gen.ncp_info == {"alpha": {"mu": "mu", "sigma": "sigma"}}
    trace, stats = sampler.sample(spec, dict(mu=3.0, sigma=1.0, alpha=3.0), dict(num_warmup=400, num_samples=400, seed=42))
    assert 0.0 < trace["alpha"].mean() < 8.0 and np.all(trace["sigma"] > 0.0) and stats["divergences"] < 100
    ir = cg.IR()
    ir.rv("mu", "normal", dict(mu=0.0, sigma=5.0))
    ir.rv("sigma", "exponential", {"lambda": 1.0})
    ir.rv("x", "normal", dict(mu="mu", sigma="sigma"))
    spec = _spec(ir, "ri_ncp2")
    comp = sampler.compile(spec)
    try:
        import ctypes as C
        q = np.array([1.0, 0.0, 0.5])                       # flat order mu, sigma, x
        assert spec.var_names == ["mu", "sigma", "x"]
        lp, g = np.zeros(1), np.zeros(3)
        dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))   # noqa: E731
        comp.check(comp.L.exmc_hip_logp_grad_host(comp.h, dp(q), 1, 0, dp(lp), dp(g)))
        assert -100.0 < lp[0] < 0.0
    finally:
        comp.close()


def test_vectorized_chains_share_the_warmup(hip):
    """integration_test.exs:775-810: four chains, every mean 4.95 +- 1.0, ONE step size, R-hat 1.0 +- 0.2,
    fewer than 50 divergences in total."""
    spec = _spec(_normal_normal(10.0, 5.0), "ri_nn")
    comp = sampler.compile(spec)
    try:
        traces, stats = sampler.sample_chains(comp, 4, dict(num_warmup=300, num_samples=300, seed=42))
        assert len(traces) == 4 and len(stats) == 4
        assert all(abs(t["mu"].mean() - 4.95) <= 1.0 for t in traces)
        assert len({s["step_size"] for s in stats}) == 1
        assert abs(_rhat(comp, [t["mu"] for t in traces]) - 1.0) <= 0.2
        assert sum(s["divergences"] for s in stats) < 50
    finally:
        comp.close()


# ---- test/nuts_test.exs, the "Sampler" block (tests 14-19, 22) ----
def _std_normal(name="x"):
    return cg.IR().rv(name, "normal", dict(mu=0.0, sigma=1.0))


def test_nuts_standard_normal_moments(hip):
    """nuts_test.exs:306-321 (500 + 500, seed 42: |mean| < 0.3, |var - 1| < 1, divergences <= 15) and :554-569 (the
    same model at 300 + 300 -- "speculative pre-computation" is the identity the device path has no need for)."""
    spec = _spec(_std_normal("mu"), "rn_std")
    for nw, ns in ((500, 500), (300, 300)):
        trace, stats = sampler.sample(spec, {}, dict(num_warmup=nw, num_samples=ns, seed=42))
        assert abs(trace["mu"].mean()) < 0.3 and abs(trace["mu"].var() - 1.0) < 1.0 and stats["divergences"] <= 15


def test_nuts_two_parameter_prior_means(hip):
    """nuts_test.exs:323-357: mu1 ~ N(2, 0.5), mu2 ~ N(-1, 0.5): means within 0.2, variances 0.25 +- 0.2."""
    ir = cg.IR().rv("mu1", "normal", dict(mu=2.0, sigma=0.5))
    ir.rv("mu2", "normal", dict(mu=-1.0, sigma=0.5))
    trace, stats = sampler.sample(_spec(ir, "rn_two"), {}, dict(num_warmup=300, num_samples=300, seed=123))
    assert abs(trace["mu1"].mean() - 2.0) < 0.2 and abs(trace["mu2"].mean() + 1.0) < 0.2
    assert abs(trace["mu1"].var() - 0.25) < 0.2 and abs(trace["mu2"].var() - 0.25) < 0.2
    assert stats["divergences"] <= 15


def test_nuts_support_divergences_reproducibility_and_stats(hip):
    """nuts_test.exs:359-371 (Exponential(1): every draw positive), :373-382 (standard Normal: < 20 divergences),
    :384-393 (the same seed gives the same trace -- here to the bit) and :396-412 (the stats map)."""
    trace, _ = sampler.sample(_spec(cg.IR().rv("rate", "exponential", {"lambda": 1.0}), "rn_exp"), {},
                              dict(num_warmup=200, num_samples=200, seed=77))
    assert trace["rate"].min() > 0.0
    spec = _spec(_std_normal(), "rn_x")
    _, stats = sampler.sample(spec, {}, dict(num_warmup=200, num_samples=200, seed=11))
    assert stats["divergences"] < 20
    t1, _ = sampler.sample(spec, {}, dict(num_warmup=200, num_samples=100, seed=999))
    t2, _ = sampler.sample(spec, {}, dict(num_warmup=200, num_samples=100, seed=999))
    assert np.array_equal(t1["x"], t2["x"])
    _, st = sampler.sample(spec, {}, dict(num_warmup=50, num_samples=50, seed=0))
    assert isinstance(st["step_size"], float) and st["inv_mass_diag"].shape == (1,)
    assert isinstance(st["divergences"], int) and st["num_warmup"] == 50 and st["num_samples"] == 50


def test_momentum_variance_matches_the_mass_matrix(hip):
    """nuts_test.exs:92-115: p = z / sqrt(inv_mass) has variance 1 / inv_mass (0.15): the kinetic energy the kernels
    report for the momentum they draw. energy = -joint_logp_0 = KE - logp(q) at the START of a transition
    (sampler.ex:905); with a tuning given (inv_mass [0.25, 4.0]) and the chain's previous position known, KE follows,
    and E[KE] = d / 2 whatever the mass -- the check that p was scaled by 1 / sqrt(inv_mass)."""
    ir = cg.IR().rv("a", "normal", dict(mu=0.0, sigma=2.0))
    ir.rv("b", "normal", dict(mu=0.0, sigma=0.5))
    spec = _spec(ir, "rn_mass")
    comp = sampler.compile(spec)
    try:
        tuning = dict(epsilon=0.3, inv_mass=np.array([4.0, 0.25]))      # = the variances of a and b
        _, _, extra = sampler.sample_compiled_tuned(comp, tuning, {"a": 0.0, "b": 0.0}, dict(num_samples=2500, seed=123),
                                                    num_chains=2)
        raw = extra["raw"]
        logp = raw["logp"]                                             # log-density AFTER each transition
        prev = np.concatenate([np.full((2, 1), np.nan), logp[:, :-1]], axis=1)
        ke = raw["energy"] + prev                                      # KE_0 = energy + logp(q_0)
        ke = ke[:, 1:].ravel()
        assert np.all(ke >= 0.0)
        assert abs(ke.mean() - 1.0) <= 0.15                            # d / 2 with d = 2
    finally:
        comp.close()


def test_custom_distribution_with_the_sampler(hip):
    """custom_dist_test.exs:190-212: x ~ the closure -(0.5 z^2 + log sigma) with mu 0, sigma 1: 300 draws after 200 warmup
    iterations, seed 42: mean 0.0 +- 0.5, a positive step size."""
    def normal(o, x, p):
        z = o.div(o.sub(x, p["mu"]), p["sigma"])
        return o.neg(o.add(o.mul(o.f32(0.5), o.mul(z, z)), o.log(p["sigma"])))
    ir = cg.IR().rv("x", "custom", dict(logpdf=normal, mu=0.0, sigma=1.0))
    trace, stats = sampler.sample(cg.compile_ir(ir, ncp=False, name="rc_custom"), {}, dict(num_samples=300, seed=42, num_warmup=200))
    assert trace["x"].shape == (300,) and abs(trace["x"].mean()) <= 0.5 and stats["step_size"] > 0.0
