"""The known answers of the reference's own distribution tests that the GENERATOR's catalogue covers beyond
dist_test.exs / new_dist_test.exs (those are in tests/golden/reference_known_answers.json): test/mv_normal_test.exs,
gaussian_random_walk_test.exs, censored_test.exs, dirichlet_test.exs, mixture_dist_test.exs, weibull_test.exs.
Inputs and expected values are transcribed case by case (path:line cited at each), the tolerances are the
reference's `assert_in_delta` deltas; every value is computed by the generated text of exmc_amd/codegen.py compiled for
the host (tests/gen_checker.py), i.e. by the arithmetic the HIP functors are compiled from. SURVEY 8 row f3."""
import math

import numpy as np
import pytest

import gen_checker as GC
import oracle as O
from exmc_amd import codegen as cg

LOG_2PI = math.log(2.0 * math.pi)


def _logp(ir, q, **kw):
    gen = cg.generate(ir, ncp=False, **kw)
    lp, g = GC.logp_grad(gen, np.asarray(q, dtype=np.float64))
    return lp, g, gen


def _normal(x, mu, sigma):
    return -0.5 * (((x - mu) / sigma) ** 2 + LOG_2PI + 2.0 * math.log(sigma))


# ---- test/mv_normal_test.exs ----
@pytest.mark.parametrize("mu,cov,x,expected,tol", [
    ([0.0, 0.0], [[1.0, 0.0], [0.0, 1.0]], [0.0, 0.0], -LOG_2PI, 1e-6),                     # :10-19
    ([0.0, 0.0], [[1.0, 0.0], [0.0, 1.0]], [1.0, 0.0], -LOG_2PI - 0.5, 1e-6),               # :21-30
    ([1.0, 2.0, 3.0], [[2.0, 0.5, 0.0], [0.5, 1.0, 0.0], [0.0, 0.0, 3.0]], [1.0, 2.0, 3.0],
     -0.5 * (3.0 * LOG_2PI + math.log(5.25)), 1e-5),                                        # :32-42
])
def test_mv_normal_logpdf_cases(mu, cov, x, expected, tol):
    ir = cg.IR().rv("x", "mv_normal", dict(mu=mu, cov=cov))
    lp, _, gen = _logp(ir, x)
    assert gen.d == len(mu) and abs(lp - expected) <= tol    # pm.size == 2 / one entry of length K (:61-79)


def test_mv_normal_gradient_is_minus_x():
    """mv_normal_test.exs:99-120: standard MvNormal, grad logp = -x at [0.5, -0.3] (1e-4)."""
    ir = cg.IR().rv("x", "mv_normal", dict(mu=[0.0, 0.0], cov=[[1.0, 0.0], [0.0, 1.0]]))
    lp, g, _ = _logp(ir, [0.5, -0.3])
    assert np.allclose(g, [-0.5, 0.3], atol=1e-4)
    assert abs(lp - (-LOG_2PI - 0.5 * (0.25 + 0.09))) <= 1e-5


# ---- test/gaussian_random_walk_test.exs ----
@pytest.mark.parametrize("sigma,x,expected,tol", [
    (1.0, [0.1, 0.3, 0.2], _normal(0.1, 0, 1) + _normal(0.2, 0, 1) + _normal(-0.1, 0, 1), 1e-5),   # :10-23
    (0.5, [0.1, 0.2], 2 * _normal(0.1, 0, 0.5), 1e-5),                                              # :25-37
    (2.0, [0.5], _normal(0.5, 0, 2.0), 1e-6),                                                       # :39-47
])
def test_gaussian_random_walk_logpdf_cases(sigma, x, expected, tol):
    ir = cg.IR().rv("s", "gaussian_random_walk", dict(sigma=sigma, steps=len(x)))
    lp, _, gen = _logp(ir, x)
    assert gen.d == len(x) and abs(lp - expected) <= tol


def test_gaussian_random_walk_with_a_scale_that_is_another_rv():
    """gaussian_random_walk_test.exs:114-132: sigma ~ Exponential(1) [:log], s ~ GRW(sigma), shape {3}: pm.size 4, the
    log-density at log sigma = 0, s = [0.1, 0.3, 0.2] is a negative number -- here also its value: the three Normal
    terms of :10-23 plus Exponential(1)'s -1 plus the Jacobian 0."""
    ir = cg.IR()
    ir.rv("sigma", "exponential", {"lambda": 1.0}, transform="log")
    ir.rv("s", "gaussian_random_walk", dict(sigma="sigma", steps=3))
    lp, _, gen = _logp(ir, [0.1, 0.3, 0.2, 0.0])     # flat order: "s" < "sigma"
    assert gen.d == 4 and lp < 0.0
    assert abs(lp - (_normal(0.1, 0, 1) + _normal(0.2, 0, 1) + _normal(-0.1, 0, 1) - 1.0)) <= 1e-5


# ---- test/censored_test.exs (a free auxiliary N(0, 1) at 0 stands in for "pm.size == 0": the generator has no
# model without a free variable; its own term, -0.5 log 2 pi, is taken off) ----
PHI_NEG1 = math.erfc(1.0 / math.sqrt(2.0)) / 2.0


@pytest.mark.parametrize("kind,value,expected,tol", [
    ("right", 1.0, math.log(PHI_NEG1), 1e-5),                                                # :45-58
    ("left", -1.0, math.log(PHI_NEG1), 1e-5),                                                # :62-75
    ("interval", dict(lower=-1.0, upper=1.0),
     math.log(0.5 * math.erfc(-1.0 / math.sqrt(2.0)) - PHI_NEG1), 1e-5),                     # :79-99
    ("interval", dict(lower=-1.0, upper=1.0), math.log(0.6827), 0.01),                       # :129-144
])
def test_censored_normal_log_likelihoods(kind, value, expected, tol):
    ir = cg.IR().rv("aux", "normal", dict(mu=0.0, sigma=1.0))
    ir.rv("x", "normal", dict(mu=0.0, sigma=1.0))
    ir.obs("x_obs", "x", value, censored=kind)
    lp, _, gen = _logp(ir, [0.0])
    assert gen.d == 1
    assert abs((lp + 0.5 * cg.LOG_2PI_F32) - expected) <= tol


def test_censored_observation_of_a_hierarchical_mean():
    """censored_test.exs:148-165: mu ~ N(0, 10) free, x ~ N(mu, 1) right-censored at 2: pm.size == 1, logp(0) < 0."""
    ir = cg.IR().rv("mu", "normal", dict(mu=0.0, sigma=10.0))
    ir.rv("x", "normal", dict(mu="mu", sigma=1.0))
    ir.obs("x_obs", "x", 2.0, censored="right")
    lp, g, gen = _logp(ir, [0.0])
    assert gen.d == 1 and lp < 0.0 and g[0] > 0.0      # a larger mean makes "x > 2" likelier


# ---- test/weibull_test.exs ----
@pytest.mark.parametrize("t,k,lam,expected,tol", [
    (1.0, 2.0, 1.0, math.log(2.0) - 1.0, 1e-6),                                              # :11-17
    (2.0, 1.0, 1.0, -2.0, 1e-6),                                                             # :19-25 (= Exponential(1))
    (0.5, 3.0, 2.0, math.log(3.0) - 3.0 * math.log(2.0) + 2.0 * math.log(0.5) - (0.5 / 2.0) ** 3, 1e-6),   # :27-36
])
def test_weibull_logpdf_cases(t, k, lam, expected, tol):
    ir = cg.IR().rv("t", "weibull", {"k": k, "lambda": lam})
    lp, _, _ = _logp(ir, [t])
    assert abs(lp - expected) <= tol


@pytest.mark.parametrize("t,k,lam,expected,tol", [
    (0.001, 2.0, 1.0, 0.0, 1e-4),                                                            # :76-79 log_survival
    (2.0, 1.0, 2.0, -1.0, 1e-6),                                                             # :81-85
    (1.0, 2.0, 1.0, -1.0, 1e-6),                                                             # :89-94 Censored :right
])
def test_weibull_log_survival_as_a_right_censored_observation(t, k, lam, expected, tol):
    ir = cg.IR().rv("aux", "normal", dict(mu=0.0, sigma=1.0))
    ir.rv("t_rv", "weibull", {"k": k, "lambda": lam})
    ir.obs("t", "t_rv", t, censored="right")
    lp, _, _ = _logp(ir, [0.0])
    assert abs((lp + 0.5 * cg.LOG_2PI_F32) - expected) <= tol


# ---- test/mixture_dist_test.exs ----
@pytest.mark.parametrize("x,w,expected", [
    (0.0, [0.5, 0.5], math.log(0.5 * math.exp(_normal(0.0, -2, 1)) + 0.5 * math.exp(_normal(0.0, 2, 1)))),     # :9-29
    (-2.0, [0.8, 0.2], math.log(0.8 * math.exp(_normal(-2.0, -2, 1)) + 0.2 * math.exp(_normal(-2.0, 2, 1)))),  # :31-50
])
def test_mixture_logpdf_cases(x, w, expected):
    ir = cg.IR().rv("x", "mixture", dict(components=["normal", "normal"],
                                         params=[dict(mu=-2.0, sigma=1.0), dict(mu=2.0, sigma=1.0)], weights=w))
    lp, _, _ = _logp(ir, [x])
    assert abs(lp - expected) <= 1e-5


def test_mixture_three_components_and_compiled_gradient():
    """mixture_dist_test.exs:52-70 (three components: a finite number) and :116-160 (the compiled model at 0: finite
    value and gradient; by symmetry of the +-1 components the gradient at 0 is 0)."""
    ir = cg.IR().rv("x", "mixture", dict(components=["normal"] * 3,
                                         params=[dict(mu=-3.0, sigma=0.5), dict(mu=0.0, sigma=1.0), dict(mu=3.0, sigma=0.5)],
                                         weights=[0.2, 0.6, 0.2]))
    lp, _, _ = _logp(ir, [0.0])
    assert math.isfinite(lp)
    ir = cg.IR().rv("x", "mixture", dict(components=["normal", "normal"],
                                         params=[dict(mu=-1.0, sigma=1.0), dict(mu=1.0, sigma=1.0)], weights=[0.5, 0.5]))
    lp, g, gen = _logp(ir, [0.0])
    assert gen.d == 1 and math.isfinite(lp) and abs(g[0]) <= 1e-12


# ---- test/dirichlet_test.exs ----
def _stick_ladj(z):
    """log |det J| of the stick-breaking map the test's literals describe (z = 0 -> [0.5, 0.25, 0.25], :50-64):
    y_k = sigmoid(z_k), x_k = y_k * rem_k, rem_{k+1} = rem_k (1 - y_k); triangular, dx_k/dz_k = y_k (1 - y_k) rem_k."""
    y = 1.0 / (1.0 + np.exp(-np.asarray(z, dtype=np.float64)))
    rem, out = 1.0, 0.0
    for yk in y:
        out += math.log(yk * (1.0 - yk) * rem)
        rem *= 1.0 - yk
    return out


def test_stick_breaking_forward_literals():
    """dirichlet_test.exs:50-75: z = 0 -> [0.5, 0.25, 0.25]; any z -> a point of the simplex."""
    x = cg.stick_breaking(np.array([0.0, 0.0]))
    assert np.allclose(x, [0.5, 0.25, 0.25], atol=1e-6) and abs(x.sum() - 1.0) <= 1e-6
    x = cg.stick_breaking(np.array([1.0, -0.5, 0.3]))
    assert x.shape == (4,) and np.all(x > 0) and abs(x.sum() - 1.0) <= 1e-6


@pytest.mark.parametrize("alpha,z,expected,tol", [
    ([1.0, 1.0, 1.0], [math.log(0.5), 0.0], math.log(2.0), 1e-5),                             # :10-18, x = centre
    ([2.0, 2.0, 2.0], [math.log(0.5), 0.0], -3.0 * math.log(3.0) + math.log(120.0), 1e-4),    # :20-29
    ([1.0, 1.0], [0.0], 0.0, 1e-6),                                                           # :31-40, x = [0.5, 0.5]
    ([1.0, 1.0, 1.0], [0.0, 0.0], math.log(2.0), 1e-5),                                       # :196-217, x = [.5, .25, .25]
])
def test_dirichlet_logpdf_cases_through_the_compiled_model(alpha, z, expected, tol):
    """The compiled log-density of a free Dirichlet is logpdf(x(z)) + log|J|(z) (compiler.ex:222-229,
    dirichlet_test.exs:196-217); taking the Jacobian off leaves the test's logpdf literals."""
    ir = cg.IR().rv("w", "dirichlet", dict(alpha=alpha), transform="stick_breaking")
    lp, _, gen = _logp(ir, z)
    assert gen.d == len(alpha) - 1                              # unconstrained_length, :138-147
    assert abs((lp - _stick_ladj(z)) - expected) <= tol


def test_dirichlet_gradient_against_central_differences():
    """dirichlet_test.exs:219-245: alpha = [2, 3, 1], z = [0.5, -0.3], gradient against finite differences (0.01)."""
    ir = cg.IR().rv("w", "dirichlet", dict(alpha=[2.0, 3.0, 1.0]), transform="stick_breaking")
    gen = cg.generate(ir, ncp=False)
    z = np.array([0.5, -0.3])
    _, g = GC.logp_grad(gen, z)
    fd = np.zeros(2)
    for i in range(2):
        e = np.zeros(2)
        e[i] = 1e-4
        fd[i] = (GC.logp_grad(gen, z + e)[0] - GC.logp_grad(gen, z - e)[0]) / 2e-4
    assert np.allclose(g, fd, atol=0.01)
