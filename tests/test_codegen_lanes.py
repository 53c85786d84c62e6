"""The lane layout of generated models (exmc_amd/codegen_lanes.py) on the CPU:
stochastic volatility, radon and the logistic regression built from Builder node lists
(codegen.sv_ir / radon_ir / logistic_ir) against the HAND-WRITTEN models of the oracle
(oracle/exmc_oracle.c logp_sv / logp_radon / logp_logistic, reference arithmetic: libm, left to
right) -- an independent restatement of the same densities -- and against central differences.
The generated text runs on virtual lanes (tests/gen_checker.py), as it does on the GPU."""
import numpy as np
import pytest

import gen_checker as GC
import gen_models as GM
import oracle as O
from exmc_amd import codegen as cg, codegen_lanes as cl


def _gen(which, lanes=None):
    ir, ncp, spec, dl = GM.baseline_pair(which)
    return cg.generate(ir, ncp=ncp, lanes=lanes or dl), spec


@pytest.mark.parametrize("which,lanes", [("sv", 64), ("sv", 32), ("radon", 64), ("logistic", 16),
                                         ("logistic", 64)])
def test_generated_equals_the_handwritten_oracle_model(which, lanes):
    gen, spec = _gen(which, lanes)
    assert gen.d == spec.d and sorted(gen.var_names) == sorted(spec.var_names)
    assert gen.var_names == sorted(gen.var_names)           # PointMap order (point_map.ex:37)
    m = O.model_for(spec)
    idx = GM.to_spec_order(gen, spec)
    rng = np.random.default_rng(5)
    q0 = spec.to_unconstrained(spec.default_init)
    scale = 0.1 if which == "sv" else 0.3
    # logistic: the reference's Bernoulli term differentiates y log p + (1 - y) log(1 - p) through
    # p (y / p - (1 - y) / (1 - p), as Nx's AD does); the hand-written kind uses the closed form
    # y - p. Where p saturates the two differ by the cancellation in 1 - p, so its points stay
    # at moderate linear predictors
    spread = 1 if which == "logistic" else 4
    for t in range(200):
        qs = q0 + scale * rng.normal(size=spec.d) * (1.0 + (t % spread))
        lp_o, g_o = m.logp_grad(qs, O.Cfg(0, 1))
        lp_g, g_g = GC.logp_grad(gen, qs[idx], lanes=lanes)
        assert abs(lp_g - lp_o) <= 1e-12 * max(1.0, abs(lp_o)), (which, t, lp_g, lp_o)
        # an entry of the gradient is a sum of per-term adjoints that cancel (sv's d/d log nu: a hundred
        # terms of the size of the largest entry), so the bound is relative to the gradient's scale
        go = g_o[idx]
        assert np.all(np.abs(g_g - go) <= 1e-12 * max(1.0, np.max(np.abs(go)))), (which, t)


@pytest.mark.parametrize("which", ["sv", "radon", "logistic"])
def test_generated_gradient_against_central_differences(which):
    gen, spec = _gen(which)
    lanes = gen.lanes
    idx = GM.to_spec_order(gen, spec)
    q = (spec.to_unconstrained(spec.default_init) + 0.05 * np.random.default_rng(2).normal(size=spec.d))[idx]
    _, g = GC.logp_grad(gen, q, lanes=lanes)
    for i in list(range(0, gen.d, 7)) + [gen.d - 1]:
        h = 1e-6
        qp, qm = q.copy(), q.copy()
        qp[i] += h
        qm[i] -= h
        fd = (GC.logp_grad(gen, qp, lanes=lanes)[0] - GC.logp_grad(gen, qm, lanes=lanes)[0]) / (2 * h)
        assert abs(fd - g[i]) <= 2e-5 * max(1.0, abs(g[i])), (which, i, fd, g[i])


def test_structure_found_in_the_graphs():
    """What the hand-written kernels exploit is found from the node lists: sv = the 100 StudentT
    terms + the 99 random-walk steps (s_1's prior has another shape and stays uniform, like the two
    hyper-priors); radon = 88 Normal priors (85 intercepts + 3 hyper-parameters, their constants
    per unit) + 919 observations gathering their county's intercept; logistic = 21 priors + 500
    observations whose 21 inputs are all shared (no gather: 21 reduced adjoints + the density)."""
    sv, _ = _gen("sv")
    assert sv.lane_layout["family_sizes"] == [100, 99] and sv.lane_layout["n_scalar_units"] == 3
    assert sv.lane_layout["gather_width"] == [3, 3]          # s_t: its own term, step t, step t + 1
    rd, _ = _gen("radon")
    assert rd.lane_layout["family_sizes"] == [88, 919] and rd.lane_layout["n_scalar_units"] == 2
    lg, _ = _gen("logistic")
    # the 500 observations as two families, by WHICH of y and 1 - y is zero: each half of
    # y * log p + (1 - y) * log(1 - p) then evaluates one logarithm and one reciprocal (0 * log of a
    # clipped probability is provably finite and folded away)
    sizes = lg.lane_layout["family_sizes"]
    assert sizes[0] == 21 and sum(sizes[1:]) == 500 and len(sizes) == 3 and lg.lane_layout["n_reduced"] == 22
    text = lg.lane_layout["text"]
    for fam in ("/* family 1", "/* family 2"):
        body = text[text.index(fam):]
        body = body[:body.index("\n  }\n") if "\n  }\n" in body else len(body)]
        assert body.count("EXMC_GENL_LOG(") == 1 and body.count("1.0 / ") == 2 and body.count("EXMC_GENL_EXP(") == 1
    # radon's observation loop multiplies the floor indicator with a plain variable: not worth a loop
    assert rd.lane_layout["family_sizes"] == [88, 919]
    assert lg.lane_layout["dpl"] == 2 and sv.lane_layout["dpl"] == 2 and rd.lane_layout["dpl"] == 2


def test_models_above_the_one_lane_limit_pick_the_lane_layout_themselves():
    ir, ncp, spec, _ = GM.baseline_pair("radon")
    gen = cg.generate(ir, ncp=ncp)
    assert gen.lanes == 64 and "EXMC_GEN_ONE_LANE" not in gen.header and gen.lane_layout is not None
    ir, ncp, spec, _ = GM.baseline_pair("logistic")
    assert cg.generate(ir, ncp=ncp).lanes == 16            # d = 21: 16 lanes, two dimensions per lane
    with pytest.raises(cg.CodegenError):
        cg.generate(GM.simple_ir(), lanes=8)


@pytest.mark.parametrize("name", ["zoo", "walk", "eight_schools", "simple"])
def test_lane_layout_of_small_models_equals_their_one_lane_layout(name):
    """The lane layout is general: models of the one-lane tests (every distribution and transform,
    refs, the non-centred rewrite, vector obs, a random walk, MvNormal, a Custom closure, meas_obs)
    give the same density and gradient in both layouts up to the order of the sums."""
    ir = dict(zoo=GM.zoo_ir, walk=GM.walk_ir, eight_schools=GM.eight_schools_ir, simple=GM.simple_ir)[name]()
    gen = cg.generate(ir, lanes=16)
    assert "EXMC_GEN_ONE_LANE" in gen.header and gen.lane_layout is not None and gen.vec is None
    rng = np.random.default_rng(8)
    for t in range(40):
        q = rng.normal(size=gen.d) * 0.8
        lp1, g1 = GC.logp_grad(gen, q, lanes=1)
        lpl, gl = GC.logp_grad(gen, q, lanes=16)
        assert abs(lp1 - lpl) <= 1e-12 * max(1.0, abs(lp1)), (name, t)
        assert np.all(np.abs(g1 - gl) <= 1e-11 * np.maximum(1.0, np.abs(g1))), (name, t)


def test_a_reduction_written_by_hand_is_split_into_units():
    """validate_posteriordb.exs:279-295 reduces with Enum.reduce / Nx.add, not Nx.sum: a closure
    result without a registered sum is split at every add at its top."""
    y = np.random.default_rng(1).normal(size=24)
    ir = cg.IR()
    ir.rv("mu", "normal", dict(mu=0.0, sigma=5.0))
    ir.rv("tau", "half_cauchy", dict(scale=5.0), transform="log")
    for j in range(24):
        ir.rv("t_%02d" % j, "normal", dict(mu=0.0, sigma=1.0))

    def lik(o, _x, p):
        acc = o.lit(0.0)
        for j in range(24):
            theta = o.add(p["mu"], o.mul(p["tau"], p["t_%02d" % j]))
            z = o.div(o.sub(o.data(y[j]), theta), o.data(1.0 + 0.1 * j))
            acc = o.add(acc, o.sub(o.mul(o.lit(-0.5), o.mul(z, z)), o.log(o.data(1.0 + 0.1 * j))))
        return acc
    params = {"t_%02d" % j: "t_%02d" % j for j in range(24)}
    params.update(mu="mu", tau="tau", logpdf=lik)
    ir.rv("lik", "custom", params)
    ir.obs("lik_obs", "lik", 0.0)
    gen = cg.generate(ir, lanes=16)
    assert gen.d == 26 and sorted(gen.lane_layout["family_sizes"]) == [24, 25]
    q = np.random.default_rng(3).normal(size=26) * 0.5
    lp, g = GC.logp_grad(gen, q, lanes=16)
    # closed form
    assert gen.var_names == ["mu"] + ["t_%02d" % j for j in range(24)] + ["tau"]
    mu, ltau, tau, th = q[0], q[25], np.exp(q[25]), q[1:25]
    s = 1.0 + 0.1 * np.arange(24)
    ref = (-0.5 * ((mu / 5) ** 2 + np.log(np.float32(2 * np.pi)) + 2 * np.log(5.0))
           + np.log(2 / np.pi) - np.log(5.0) - np.log1p((tau / 5) ** 2) + ltau
           + np.sum(-0.5 * (th ** 2 + np.log(np.float32(2 * np.pi))))
           + np.sum(-0.5 * ((y - (mu + tau * th)) / s) ** 2 - np.log(s)))
    assert abs(lp - ref) < 1e-6 * abs(ref)


def test_f32_params_follow_nx_type_inference():
    """Nx.log of Nx.tensor(50.0) (f32) is an f32 tensor: exponential.ex:16 with the benchmark's
    untyped literals gives f32(log 50), which is what the hand-written sv kind carries."""
    g = cg._Graph()
    lam = g.datum32(0.1)
    assert g.data[g.ops[lam][1]] == float(np.float32(0.1))
    ll = g.log(lam)
    assert ll in g.f32 and g.data[g.ops[ll][1]] == float(np.float32(np.log(float(np.float32(0.1)))))
    x = g.q(0)
    assert g.mul(lam, x) not in g.f32          # f32 x f64 -> f64, computed at run time


def test_flatten_rules():
    g = cg._Graph()
    a, b, c, d = (g.q(i) for i in range(4))
    s = cg._sum_left(g, [g.exp(a), g.exp(b), g.exp(c)])
    t = g.add(g.log(d), s)                                   # e.g. logp_init + sum(steps)
    assert cl._flatten(g, t, False) == [g.log(d), g.exp(a), g.exp(b), g.exp(c)]
    u = g.add(g.add(g.exp(a), g.exp(b)), g.exp(c))            # an add chain without a registered sum ...
    g.sums.clear()
    assert cl._flatten(g, u, False) == [u]                   # ... is one term of the model,
    assert cl._flatten(g, u, True) == [g.exp(a), g.exp(b), g.exp(c)]   # a reduction inside a closure


# ---- randomly drawn models: the lane layout against the one-lane layout (d <= 20) and against
# central differences (any d), at 16, 32 and 64 lanes per chain ----
def _random_big_ir(seed):
    """A hierarchical model with 1-3 plates of random size and kind, hyper-parameters with random
    transforms, a vector obs and a random walk: d between ~10 and ~150."""
    rng = np.random.default_rng(1000 + seed)
    ir = cg.IR()
    ir.rv("a_loc", "normal", dict(mu=0.3, sigma=2.0))
    sd = str(rng.choice(["half_cauchy", "half_normal", "exponential"]))
    sp = {"half_cauchy": dict(scale=2.0), "half_normal": dict(sigma=2.0), "exponential": {"lambda": 0.5}}[sd]
    ir.rv("b_scale", sd, sp, transform=str(rng.choice(["log", "softplus"])))
    ir.rv("c_df", "exponential", {"lambda": 0.2}, transform="log")
    for p in range(int(rng.integers(1, 4))):
        k = int(rng.integers(3, 45))
        like = str(rng.choice(["normal", "student_t", "laplace", "bernoulli"]))
        centred = bool(rng.integers(0, 2))
        for j in range(k):
            name = "g%d_%02d" % (p, j)
            ir.rv(name, "normal", dict(mu="a_loc", sigma=(float(rng.uniform(0.5, 2.0)) if centred else "b_scale")))
            nobs = int(rng.integers(1, 4))
            y = rng.normal(size=nobs) * 1.5
            if like == "normal":
                ir.rv("y%d_%02d" % (p, j), "normal", dict(mu=name, sigma=float(rng.uniform(0.5, 3.0))))
            elif like == "student_t":
                ir.rv("y%d_%02d" % (p, j), "student_t", dict(df="c_df", loc=name, scale="b_scale"))
            elif like == "laplace":
                ir.rv("y%d_%02d" % (p, j), "laplace", dict(mu=name, b=float(rng.uniform(0.5, 3.0))))
            else:
                ir.rv("q%d_%02d" % (p, j), "normal", dict(mu=name, sigma=1.0), transform="logit")
                ir.rv("y%d_%02d" % (p, j), "bernoulli", dict(p="q%d_%02d" % (p, j)))
                y = (rng.uniform(size=nobs) < 0.5).astype(float)
            ir.obs("o%d_%02d" % (p, j), "y%d_%02d" % (p, j), y if nobs > 1 else float(y[0]))
    if rng.integers(0, 2):
        ir.rv("w", "gaussian_random_walk", dict(sigma="b_scale", steps=int(rng.integers(5, 40))))
    return ir, rng


@pytest.mark.parametrize("seed", range(10))
def test_random_models_in_the_lane_layout(seed):
    ir, rng = _random_big_ir(seed)
    lanes = (16, 32, 64)[seed % 3]
    gen = cg.generate(ir, lanes=lanes)
    assert gen.lane_layout is not None and gen.lane_layout["n_families"] >= 1
    q = rng.normal(size=gen.d) * 0.6
    lp, g = GC.logp_grad(gen, q, lanes=lanes)
    assert np.isfinite(lp) and np.all(np.isfinite(g))
    if "EXMC_GEN_ONE_LANE" in gen.header:          # d <= 20: the one-lane form is an independent emission
        for _ in range(4):
            q2 = rng.normal(size=gen.d) * 0.8
            a, ga = GC.logp_grad(gen, q2, 1)
            b, gb = GC.logp_grad(gen, q2, lanes=lanes)
            assert abs(a - b) <= 1e-12 * max(1.0, abs(a))
            np.testing.assert_allclose(ga, gb, rtol=1e-10, atol=1e-10)
    idx = rng.choice(gen.d, size=min(gen.d, 12), replace=False)
    for i in idx:
        e = np.zeros(gen.d)
        e[i] = 1e-6
        fd = (GC.logp_grad(gen, q + e, lanes=lanes)[0] - GC.logp_grad(gen, q - e, lanes=lanes)[0]) / 2e-6
        assert abs(fd - g[i]) <= 2e-5 * max(1.0, abs(g[i]), abs(lp) * 1e-3), (seed, i, fd, g[i])
    # the two layouts of one model differ in the order of their sums only
    other = 64 if lanes != 64 else 16
    gen2 = cg.generate(ir, lanes=other)
    lp2, g2 = GC.logp_grad(gen2, q, lanes=other)
    assert abs(lp - lp2) <= 1e-11 * max(1.0, abs(lp))
    np.testing.assert_allclose(g, g2, rtol=1e-9, atol=1e-9)


def test_uniform_part_is_spread_over_the_lanes_too():
    """sv's shared part holds two Lanczos series (StudentT's normaliser, math.ex:27-52), six
    logarithms and their tangents. The series are evaluated one quotient per lane and reduced in a
    butterfly of their own; the logarithms / reciprocals of one dependency level are evaluated
    together (lane i takes argument i); a quotient whose denominator's reciprocal the adjoint needs
    anyway is a product with that reciprocal, and a reciprocal of a shared value leaves the family
    loop. (The values: test_generated_equals_the_handwritten_oracle_model.)"""
    sv, _ = _gen("sv")
    lay = sv.lane_layout
    assert lay["spread_sizes"] == [8, 8] and lay["n_spread_sums"] == 4 and lay["n_batches"] >= 3
    text = lay["text"]
    assert "EXMC_GEN_BATCH_LOG(" in text and "EXMC_GEN_BATCH_RCP(" in text and "EXMC_GEN_ALLSUM_W(w)" in text
    loop0 = text[text.index("/* family 0"):text.index("/* family 1")]
    loop1 = text[text.index("/* family 1"):text.index("EXMC_GEN_ALLSUM(s)")]
    body0 = loop0[loop0.index("  for (int "):]
    body1 = loop1[loop1.index("  for (int "):]
    assert body0.count(" / ") == 2 and body1.count(" / ") == 0       # were 5 and 2
    assert loop1[:loop1.index("  for (int ")].count("1.0 / ") == 1   # 1 / sigma, once per leapfrog
    # short table rows and gathered variables are fetched for a block of slots before the arithmetic
    assert "for (int jb = 0;" in body0 and "v_[j][0] = EXMC_GEN_SH(ix_[j][0]);" in body0
    # radon: the observation loop divides by the shared noise scale -- no quotient left in it
    rd, _ = _gen("radon")
    t = rd.lane_layout["text"]
    obs = t[t.index("/* family 1"):t.index("EXMC_GEN_ALLSUM(s)")]
    assert obs[obs.index("  for (int "):].count(" / ") == 0 and rd.lane_layout["spread_sizes"] == []


def test_a_spread_sum_does_not_feed_another_one():
    """One level of spreading: a series whose argument is computed from another series' sum stays in
    the uniform part (evaluated by every lane), the inner one is spread; both layouts agree."""
    ir = cg.IR()
    ir.rv("a", "normal", dict(mu=0.0, sigma=1.0), transform=None)
    ir.rv("b", "normal", dict(mu=0.0, sigma=1.0), transform=None)
    for j in range(16):                 # d = 18: the one-lane layout (d <= 20) is the comparison
        ir.rv("t_%02d" % j, "normal", dict(mu="a", sigma=1.5))

    def lik(o, _x, p):
        alpha = o.add(o.lit(2.0), o.exp(p["a"]))
        inner = o.logpdf("gamma", o.lit(1.3), {"alpha": alpha, "beta": o.lit(2.0)})       # lgamma(alpha)
        alpha2 = o.add(o.lit(3.0), o.exp(o.mul(o.lit(0.01), inner)))
        outer = o.logpdf("gamma", o.add(o.lit(0.5), o.exp(p["b"])), {"alpha": alpha2, "beta": o.lit(1.0)})
        return o.add(inner, outer)
    ir.rv("lik", "custom", dict(logpdf=lik, a="a", b="b"))
    ir.obs("lik_obs", "lik", 0.0)
    gen = cg.generate(ir, lanes=16)
    assert gen.lane_layout["spread_sizes"] == [8]
    rng = np.random.default_rng(5)
    for t in range(20):
        q = rng.normal(size=gen.d) * 0.6
        lp1, g1 = GC.logp_grad(gen, q, lanes=1)
        lpl, gl = GC.logp_grad(gen, q, lanes=16)
        assert abs(lp1 - lpl) <= 1e-12 * max(1.0, abs(lp1)), t
        assert np.all(np.abs(g1 - gl) <= 1e-11 * np.maximum(1.0, np.abs(g1))), t


@pytest.mark.parametrize("which,lanes", [("logistic", 16), ("sv", 32)])
def test_one_chain_form_spreads_the_units_over_the_wavefront(which, lanes):
    """The shared warmup runs ONE chain, so a layout of fewer than 64 lanes per chain has a form in
    which the 64 / G lane groups of the wavefront share the units of every family (group g takes the
    slots g, g + 64 / G, ...) and the groups' reduced sums are added in group order: the same
    density and gradient up to the order of the sums (the checker runs both forms of the same text)."""
    gen, hand = _gen(which, lanes)
    assert "EXMC_GEN_NG" in gen.lane_layout["text"] and "EXMC_GEN_XGROUP(s)" in gen.lane_layout["text"]
    rng = np.random.default_rng(21)
    differ = 0
    for t in range(25):
        q = rng.normal(size=gen.d) * (0.1 if which == "sv" else 0.4)
        a, ga = GC.logp_grad(gen, q, lanes=lanes)
        b, gb = GC.logp_grad(gen, q, lanes=lanes, wave_split=True)
        assert abs(a - b) <= 1e-12 * max(1.0, abs(a)), t
        assert np.all(np.abs(ga - gb) <= 1e-11 * np.maximum(1.0, np.abs(ga))), t
        differ += int(a != b or not np.array_equal(ga, gb))
    assert differ > 0          # a different order of the sums: a layout of its own for the checker


def test_zero_factor_is_folded_only_when_the_other_factor_is_provably_finite():
    """y * log p + (1 - y) * log(1 - p) with an UNCLIPPED p = sigmoid(eta): log p can be -inf, and the
    reference's 0 * -inf is NaN. The observations are still split by their zero factors, but nothing is
    folded -- both logarithms stay in both halves -- and at an extreme position the lane layout gives
    the NaN the one-lane layout gives. With the probability clipped (Bernoulli's own logpdf) the fold
    happens (test_structure_found_in_the_graphs)."""
    rng = np.random.default_rng(9)
    n = 12
    x = rng.normal(size=n)
    y = (rng.uniform(size=n) < 0.5).astype(float)
    y[:4] = 1.0
    y[4:8] = 0.0
    ir = cg.IR()
    ir.rv("a", "normal", dict(mu=0.0, sigma=2.0))
    ir.rv("b", "normal", dict(mu=0.0, sigma=2.0))

    def lik(o, _x, p):
        terms = []
        for i in range(n):
            pr = o.sigmoid(o.add(p["a"], o.mul(p["b"], o.data(x[i]))))
            terms.append(o.add(o.mul(o.data(y[i]), o.log(pr)),
                               o.mul(o.data(1.0 - y[i]), o.log(o.sub(o.lit(1.0), pr)))))
        return o.sum(terms)
    ir.rv("lik", "custom", dict(logpdf=lik, a="a", b="b"))
    ir.obs("lik_obs", "lik", 0.0)
    gen = cg.generate(ir, lanes=16)
    lay = gen.lane_layout
    assert len(lay["family_sizes"]) == 2 and sum(lay["family_sizes"]) == n      # split by which factor is zero ...
    text = lay["text"]
    halves = (text[text.index("/* family 0"):text.index("/* family 1")],
              text[text.index("/* family 1"):text.index("EXMC_GEN_ALLSUM(s)")])
    assert all(h.count("EXMC_GENL_LOG(") == 2 for h in halves)                  # ... nothing folded
    for q in (np.array([0.3, -0.7]), np.array([800.0, 0.0]), np.array([-800.0, 0.0])):
        a, ga = GC.logp_grad(gen, q, lanes=1)
        b, gb = GC.logp_grad(gen, q, lanes=16)
        assert (np.isnan(a) and np.isnan(b)) or abs(a - b) <= 1e-12 * max(1.0, abs(a)), (q, a, b)
        assert np.array_equal(np.isnan(ga), np.isnan(gb)), (q, ga, gb)
    assert np.isnan(GC.logp_grad(gen, np.array([800.0, 0.0]), lanes=16)[0])
