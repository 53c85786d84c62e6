"""Generator coverage added at the end of round 2: Gamma, Beta, Weibull, Poisson,
Uniform01 (dist/gamma.ex:15-27, beta.ex:15-24, weibull.ex:17-27, poisson.ex:16-20,
uniform01.ex:14-16) and the Dirichlet distribution behind the :stick_breaking transform
(dist/dirichlet.ex:19-27; transform.ex:125-143, 186-203, 234-249). CPU: the generated text compiled
for the host against scipy's densities and a numerically differentiated Jacobian."""
import math

import numpy as np
import pytest
from scipy import stats

import gen_checker as GC
import gen_models as GM
from exmc_amd import codegen as cg, sampler


def _one(dist, params, x, transform, to_z):
    ir = cg.IR()
    ir.rv("x", dist, params, transform=transform)
    return GC.logp_grad(cg.generate(ir), np.array([to_z(x)]))[0]


def test_doctest_literals():
    logit = lambda x: math.log(x / (1 - x))   # noqa: E731
    # each value = the module's doctest + the transform's log-Jacobian at that point
    assert round(_one("gamma", dict(alpha=2.0, beta=1.0), 1.0, "log", math.log) - 0.0, 6) == -1.0          # gamma.ex:8-9
    # beta.ex:8-9 prints 0.546966 (an all-f32 doctest); the f64 path gives ln 1.728 = 0.546965
    assert abs(_one("beta", dict(alpha=2.0, beta=3.0), 0.4, "logit", logit) - math.log(0.4 * 0.6) - 0.546966) < 3e-6
    assert round(_one("weibull", {"k": 2.0, "lambda": 1.0}, 1.0, "log", math.log), 6) == -0.306853        # weibull.ex:9-10
    assert abs(_one("uniform01", {}, 0.3, "logit", logit) - math.log(0.3 * 0.7)) < 1e-12                 # uniform01.ex:8-9: 0.0
    ir = cg.IR()
    ir.rv("th", "dirichlet", dict(alpha=[1.0, 1.0, 1.0]), transform="stick_breaking")
    gen = cg.generate(ir)
    assert gen.d == 2 and gen.var_names == ["th[0]", "th[1]"] and gen.simplex_entries == {"th": (0, 2)}
    z = cg.inverse_stick_breaking([1 / 3, 1 / 3, 1 / 3])
    np.testing.assert_allclose(cg.stick_breaking(z), [1 / 3] * 3, atol=1e-15)
    ladj = math.log(abs(np.linalg.det(_jac(z))))
    assert round(GC.logp_grad(gen, z)[0] - ladj, 4) == 0.6931                                              # dirichlet.ex:13-16


def _jac(z, h=1e-6):
    """d x[:K-1] / d z of the stick-breaking map, central differences."""
    n = z.shape[0]
    J = np.zeros((n, n))
    for j in range(n):
        e = np.zeros(n); e[j] = h
        J[:, j] = (cg.stick_breaking(z + e)[:n] - cg.stick_breaking(z - e)[:n]) / (2 * h)
    return J


def scipy_logp(q, ir):
    # flat order = ids sorted: k, p, rate, theta[0..2], u
    n = ir.nodes
    zk, zp, zr, zth, zu = q[0], q[1], q[2], q[3:6], q[6]
    k, rate = math.exp(zk), math.exp(zr)
    sig = lambda v: 1.0 / (1.0 + math.exp(-v))   # noqa: E731
    p, u = sig(zp), sig(zu)
    theta = cg.stick_breaking(zth)
    lp = stats.dirichlet.logpdf(theta, n["theta"]["params"]["alpha"]) + math.log(abs(np.linalg.det(_jac(zth))))
    lp += stats.gamma.logpdf(rate, 3.0, scale=1 / 2.0) + zr
    lp += stats.beta.logpdf(p, 2.0, 5.0) + math.log(p * (1 - p))
    lp += stats.weibull_min.logpdf(k, 1.5, scale=2.0) + zk
    lp += 0.0 + math.log(u * (1 - u))
    lp += np.sum(stats.poisson.logpmf(n["cnt"]["value"], rate))
    lp += np.sum(stats.weibull_min.logpdf(n["wait"]["value"], k, scale=1.3))
    lp += np.sum(stats.gamma.logpdf(n["g"]["value"], 2.5, scale=1 / rate))
    lp += np.sum(stats.bernoulli.logpmf(n["b"]["value"].astype(int), p))
    lp += stats.dirichlet.logpdf(n["mix"]["value"], [4.0, 2.0, 1.0, 1.0])
    lp += stats.norm.logpdf(0.6, u, 0.5)
    return lp


def test_value_against_scipy_and_gradient_against_central_differences():
    ir = GM.simplex_ir()
    gen = cg.generate(ir)
    assert gen.d == 7 and gen.var_names == ["k", "p", "rate", "theta[0]", "theta[1]", "theta[2]", "u"]
    rng = np.random.default_rng(3)
    for _ in range(10):
        q = rng.normal(size=gen.d) * 0.7
        lp, g = GC.logp_grad(gen, q)
        want = scipy_logp(q, ir)
        # the reference's lgamma carries f32-rounded Lanczos coefficients (math.ex:10-20): 1e-6 level
        assert abs(lp - want) <= 5e-6 * (1 + abs(want)), (lp, want)
        for i in range(gen.d):
            h = 1e-6
            e = np.zeros(gen.d); e[i] = h
            fd = (GC.logp_grad(gen, q + e)[0] - GC.logp_grad(gen, q - e)[0]) / (2 * h)
            assert abs(fd - g[i]) <= 3e-5 * (1 + abs(g[i])), (i, fd, g[i])


def test_simplex_init_and_trace():
    ir = GM.simplex_ir()
    gen = cg.generate(ir)

    class Spec(cg.GeneratedSpec):
        pass
    spec = Spec(gen, lib_path=None, default_init=GM.SIMPLEX_INIT)
    q0 = spec.to_unconstrained(GM.SIMPLEX_INIT)
    np.testing.assert_allclose(cg.stick_breaking(q0[3:6]), [0.25] * 4, atol=1e-15)
    assert abs(q0[2]) < 1e-15 and abs(q0[0] - math.log(1.2)) < 1e-15          # rate = 1, k = 1.2
    draws = np.random.default_rng(0).normal(size=(5, gen.d))
    tr = sampler._build_trace(spec, draws)
    assert tr["theta"].shape == (5, 4) and np.allclose(tr["theta"].sum(axis=1), 1.0) and np.all(tr["theta"] > 0)
    assert "theta[0]" not in tr and np.all((tr["p"] > 0) & (tr["p"] < 1)) and np.all(tr["rate"] > 0)


def test_refusals():
    ir = cg.IR()
    ir.rv("th", "dirichlet", dict(alpha=[1.0, 1.0, 1.0]))                  # a free simplex rv needs its transform
    with pytest.raises(cg.CodegenError):
        cg.generate(ir)
    ir = cg.IR()
    ir.rv("th", "dirichlet", dict(alpha=[1.0]), transform="stick_breaking")
    with pytest.raises(cg.CodegenError):
        cg.generate(ir)
    ir = cg.IR()
    ir.rv("x", "inverse_gamma", dict(alpha=2.0, beta=1.0), transform="log")   # not one of lib/exmc/dist
    with pytest.raises(cg.CodegenError):
        cg.generate(ir)
