"""Generator coverage: Builder.obs meta (compiler.ex:272-336, 401-418: censored, weight, mask,
reduce :sum | :mean | :logsumexp), observations of transformed rvs (compiler.ex:284-291), the
censored likelihoods of dist/censored.ex:14-68 (its own erfc polynomial) and dist/mixture.ex:13-27.
CPU: the generated text compiled for the host against scipy / numpy and central differences."""
import math

import numpy as np
import pytest
from scipy import special, stats

import gen_checker as GC
import gen_models as GM
from exmc_amd import codegen as cg


def _erfc_as(x):
    # the reference's 7.1.26 polynomial (max error 1.5e-7); the test compares against it exactly
    # enough and against scipy's erfc loosely
    t = 1.0 / (1.0 + 0.3275911 * abs(x))
    poly = 0.0
    for c in reversed([0.254829592, -0.284496736, 1.421413741, -1.453152027, 1.061405429]):
        poly = c + t * poly
    r = t * poly * math.exp(-x * x)
    return 2.0 - r if x < 0 else r


def _cdf(z):
    return 0.5 * _erfc_as(-z / math.sqrt(2.0))


def numpy_logp(q, ir):
    # flat order = ids sorted: k, lam, m, m1, m2, s
    n = ir.nodes
    k, lam, m, m1, m2, s = math.exp(q[0]), math.exp(q[1]), q[2], q[3], q[4], math.exp(q[5])
    lp = stats.gamma.logpdf(k, 2.0, scale=1.0) + q[0]
    lp += stats.lognorm.logpdf(lam, 0.8, scale=math.exp(0.5)) + q[1]
    lp += stats.norm.logpdf(m, 0.0, 2.0) + stats.norm.logpdf(m1, -1.0, 1.0) + stats.norm.logpdf(m2, 2.0, 1.0)
    lp += stats.halfnorm.logpdf(s, scale=1.5) + q[5]
    lp += np.sum(stats.weibull_min.logpdf(n["t_exact"]["value"], k, scale=lam))
    lp += sum(-((t / lam) ** k) for t in n["t_cens"]["value"])
    lp += sum(math.log(_cdf((x - m) / s)) for x in n["x_left"]["value"])
    lp += math.log(_cdf(-(1.7 - m) / s))
    lp += sum(math.log(_cdf((hi - m) / s) - _cdf((lo - m) / s)) for lo, hi in n["x_int"]["value"])
    w, mk = n["x_w"]["meta"]["weight"], n["x_w"]["meta"]["mask"]
    lp += sum(stats.norm.logpdf(x, m, s) * wi for x, wi, on in zip(n["x_w"]["value"], w, mk) if on)
    lp += np.mean(stats.norm.logpdf(n["x_mean"]["value"], m, s) * 2.0)
    lp += special.logsumexp(stats.norm.logpdf(n["x_lse"]["value"], m, s))
    for x in n["pos"]["value"]:                          # obs of a :log-transformed rv: + log x (Jacobian at z = log x)
        lp += stats.lognorm.logpdf(x, 0.7, scale=math.exp(m)) + math.log(x)
    for x in n["mix"]["value"]:
        lp += special.logsumexp([math.log(0.35) + stats.norm.logpdf(x, m1, 0.6),
                                 math.log(0.65) + stats.norm.logpdf(x, m2, 1.1)])
    lp += np.sum(stats.truncnorm.logpdf(n["tn"]["value"], (-2.0 - m) / s, (3.0 - m) / s, loc=m, scale=s))
    return lp


def test_value_and_gradient():
    ir = GM.survival_ir()
    gen = cg.generate(ir)
    assert gen.d == 6 and gen.var_names == ["k", "lam", "m", "m1", "m2", "s"] and gen.lanes == 1
    rng = np.random.default_rng(5)
    for _ in range(10):
        q = rng.normal(size=gen.d) * 0.5
        lp, g = GC.logp_grad(gen, q)
        want = numpy_logp(q, ir)
        # f32-rounded literals (the erfc coefficients, sqrt 2, the Lanczos table): 1e-6 level
        assert abs(lp - want) <= 5e-6 * (1 + abs(want)), (lp, want)
        for i in range(gen.d):
            h = 1e-6
            e = np.zeros(gen.d); e[i] = h
            fd = (GC.logp_grad(gen, q + e)[0] - GC.logp_grad(gen, q - e)[0]) / (2 * h)
            assert abs(fd - g[i]) <= 5e-5 * (1 + abs(g[i])), (i, fd, g[i])


def test_reference_erfc_polynomial_is_what_is_generated():
    # censored.ex:53-54 states a maximum error of ~1.5e-7 against the true function
    for x in np.linspace(-3, 3, 25):
        assert abs(_erfc_as(x) - special.erfc(x)) < 2e-7
    # a single right-censored Normal obs: log(1 - Phi((x - mu) / sigma)) with that erfc
    ir = cg.IR()
    ir.rv("mu", "normal", dict(mu=0.0, sigma=10.0))
    ir.rv("x_rv", "normal", dict(mu="mu", sigma=2.0))
    ir.obs("x", "x_rv", 1.0, censored="right")
    gen = cg.generate(ir)
    lp = GC.logp_grad(gen, np.array([0.3]))[0]
    want = stats.norm.logpdf(0.3, 0.0, 10.0) + math.log(_cdf(-(1.0 - 0.3) / 2.0))
    assert abs(lp - want) < 1e-6 and abs(lp - (stats.norm.logpdf(0.3, 0.0, 10.0) + stats.norm.logsf(1.0, 0.3, 2.0))) < 1e-6


def test_refusals():
    ir = cg.IR()
    ir.rv("a", "gamma", dict(alpha=2.0, beta=1.0), transform="log")
    ir.rv("x_rv", "gamma", dict(alpha="a", beta=1.0))
    ir.obs("x", "x_rv", 1.0, censored="right")                 # censored.ex has no Gamma clause
    with pytest.raises(cg.CodegenError):
        cg.generate(ir)
    with pytest.raises(cg.CodegenError):
        cg.IR().rv("x", "normal", dict(mu=0.0, sigma=1.0)).obs("o", "x", [1.0, 2.0], mask=[True])
    with pytest.raises(cg.CodegenError):
        cg.IR().rv("x", "normal", dict(mu=0.0, sigma=1.0)).obs("o", "x", 1.0, censored="interval")


def test_truncated_normal_doctest_and_density():
    # test/new_dist_test.exs:68-77: TN(0, 1, -1, 1) at 0 = -log(2 pi)/2 - log(2 Phi(1) - 1), delta 1e-4;
    # :79-85: bounds at +-100 give the Normal density. (The module's doctest literal, -0.2676, is
    # not that value and the reference does not run it.)
    ir = cg.IR()
    ir.rv("x", "truncated_normal", dict(mu=0.0, sigma=1.0, lower=-1.0, upper=1.0))
    gen = cg.generate(ir)
    lp, g = GC.logp_grad(gen, np.array([0.0]))
    phi_1 = 0.5 * (1.0 + math.erf(1.0 / math.sqrt(2.0)))
    assert abs(lp - (-0.5 * math.log(2.0 * math.pi) - math.log(2.0 * phi_1 - 1.0))) < 1e-6 and g[0] == 0.0
    ir = cg.IR()
    ir.rv("x", "truncated_normal", dict(mu=0.0, sigma=1.0, lower=-100.0, upper=100.0))
    assert abs(GC.logp_grad(cg.generate(ir), np.array([0.5]))[0] - stats.norm.logpdf(0.5)) < 1e-6
    # free location and scale, constant bounds, observed values inside them
    ir = cg.IR()
    ir.rv("mu", "normal", dict(mu=0.0, sigma=3.0))
    ir.rv("sigma", "half_normal", dict(sigma=2.0), transform="log")
    ir.rv("y_rv", "truncated_normal", dict(mu="mu", sigma="sigma", lower=-0.5, upper=4.0))
    ys = [0.1, 1.3, 3.9, -0.4, 2.2]
    ir.obs("y", "y_rv", ys)
    gen = cg.generate(ir)
    rng = np.random.default_rng(2)
    for _ in range(10):
        q = rng.normal(size=2)
        mu, sg = q[0], math.exp(q[1])
        want = stats.norm.logpdf(mu, 0, 3) + stats.halfnorm.logpdf(sg, scale=2.0) + q[1]
        a, b = (-0.5 - mu) / sg, (4.0 - mu) / sg
        want += np.sum(stats.truncnorm.logpdf(ys, a, b, loc=mu, scale=sg))
        lp, g = GC.logp_grad(gen, q)
        assert abs(lp - want) <= 1e-6 * (1 + abs(want))       # f32-rounded 2 pi and sqrt 2
        for i in range(2):
            e = np.zeros(2); e[i] = 1e-6
            fd = (GC.logp_grad(gen, q + e)[0] - GC.logp_grad(gen, q - e)[0]) / 2e-6
            assert abs(fd - g[i]) <= 1e-6 * (1 + abs(g[i]))
