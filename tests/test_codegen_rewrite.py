"""The reference's IR passes as the generator applies them (rewrite.ex:13-34): default transforms
(rewrite/attach_default_transforms.ex with each module's transform/1), the measurable lifts
(lift_measurable_affine.ex, lift_measurable_matmul.ex) of obs-of-det into meas_obs with the obs meta
carried over, `likelihood: false`, and the clause-order consequences in compiler.ex for targets
that carry a transform. CPU only (values against scipy, equivalences between two spellings)."""
import math

import numpy as np
import pytest
from scipy import stats

import gen_checker as GC
from exmc_amd import codegen as cg


def _lp(gen, q):
    return GC.logp_grad(gen, np.asarray(q, dtype=np.float64))


def test_default_transforms_are_each_modules_transform():
    ir = cg.IR()
    for i, (dist, params) in enumerate([
            ("normal", dict(mu=0.0, sigma=1.0)), ("exponential", {"lambda": 1.0}), ("half_normal", dict(sigma=1.0)),
            ("half_cauchy", dict(scale=1.0)), ("beta", dict(alpha=2.0, beta=2.0)), ("gamma", dict(alpha=2.0, beta=1.0)),
            ("lognormal", dict(mu=0.0, sigma=1.0)), ("uniform01", {}), ("cauchy", dict(loc=0.0, scale=1.0)),
            ("weibull", {"k": 1.5, "lambda": 1.0}), ("student_t", dict(nu=3.0, loc=0.0, scale=1.0)),
            ("truncated_normal", dict(mu=0.0, sigma=1.0, lower=-1.0, upper=1.0)),
            ("dirichlet", dict(alpha=[1.0, 2.0, 3.0]))]):
        ir.rv("v%02d" % i, dist, params)
    ir.rv("mix", "mixture", dict(components=["exponential", "exponential"], params=[{"lambda": 1.0}, {"lambda": 3.0}],
                                 weights=[0.5, 0.5]))
    out = cg.rewrite(ir)
    got = [out.nodes[k]["transform"] for k in sorted(out.nodes)]
    assert got == ["log", None, "log", "softplus", "log", "logit", "log", "log", "logit", None, "log", None, None,
                   "stick_breaking"], got
    ir2 = cg.IR().rv("s", "half_normal", dict(sigma=1.0), transform="log")      # an explicit one stays
    assert cg.rewrite(ir2).nodes["s"]["transform"] == "log"


def test_exponential_poisson_model_of_the_reference_tests():
    # test/new_dist_test.exs:270-290: mu ~ Exponential(0.1), y ~ Poisson(mu), ten counts; the model is
    # written without transforms there. The passes give mu :log, and y -- observed -- :log as well, so
    # the obs term is evaluated at exp(log y) and log y joins it (compiler.ex:325-334).
    data = [2.0, 4.0, 3.0, 1.0, 5.0, 3.0, 2.0, 4.0, 3.0, 3.0]
    ir = cg.IR()
    ir.rv("mu", "exponential", {"lambda": 0.1})
    ir.rv("y", "poisson", dict(mu="mu"))
    ir.obs("y_obs", "y", data)
    gen = cg.generate(ir, rewrite_passes=True)
    assert gen.transforms == {"mu": "log"} and gen.d == 1
    for z in (-1.0, 0.3, 1.1, 2.0):
        mu = math.exp(z)
        want = stats.expon.logpdf(mu, scale=10.0) + z + np.sum(stats.poisson.logpmf(data, mu)) + np.sum(np.log(data))
        lp, g = _lp(gen, [z])
        assert abs(lp - want) <= 2e-5 * (1 + abs(want))             # the f32 Lanczos table
        gw = -0.1 * mu + 1.0 + (sum(data) - len(data) * mu)
        assert abs(g[0] - gw) <= 1e-9 * (1 + abs(gw))
    # without the passes the IR is taken as written: no transform on mu, nothing added to the obs term
    plain = cg.generate(ir)
    lp0 = _lp(plain, [1.3])[0]
    assert abs(lp0 - (stats.expon.logpdf(1.3, scale=10.0) + np.sum(stats.poisson.logpmf(data, 1.3)))) <= 2e-5 * 20


def test_lifted_det_obs_equals_the_meas_obs_spelling():
    def base():
        ir = cg.IR()
        ir.rv("m", "normal", dict(mu=0.0, sigma=2.0))
        ir.rv("x", "normal", dict(mu=1.0, sigma=0.5))
        ir.rv("lik_rv", "normal", dict(mu="m", sigma=1.0))
        ir.obs("lik", "lik_rv", [0.3, -0.2])
        ir.rv("v", "normal", dict(mu=0.0, sigma=1.5))
        return ir
    a_mat = [[2.0, 0.5], [0.0, 1.5]]
    w = [1.0, 0.25, 3.0]
    lifted = base()
    lifted.det("ax", "affine", [2.0, -1.0, "x"])
    lifted.obs("ax_obs", "ax", [0.5, 1.5, 2.5], weight=w)
    lifted.det("mv", "matmul", [a_mat, "v"])
    lifted.obs("mv_obs", "mv", [0.7, -0.4])
    direct = base()
    direct.meas_obs("ax_obs", "x", [0.5, 1.5, 2.5], ("affine", 2.0, -1.0), meta=dict(weight=np.asarray(w), reduce="sum"))
    direct.meas_obs("mv_obs", "v", [0.7, -0.4], ("matmul", a_mat))
    ga, gb = cg.generate(lifted, rewrite_passes=True), cg.generate(direct)
    assert ga.var_names == gb.var_names == ["m"]
    q = [0.4]
    (la, da), (lb, db) = _lp(ga, q), _lp(gb, q)
    assert la == lb and np.array_equal(da, db)
    xs = (np.array([0.5, 1.5, 2.5]) + 1.0) / 2.0
    sol = np.linalg.solve(np.array(a_mat), [0.7, -0.4])
    want = (stats.norm.logpdf(0.4, 0, 2) + np.sum(stats.norm.logpdf([0.3, -0.2], 0.4, 1.0))
            + np.sum((stats.norm.logpdf(xs, 1.0, 0.5) - math.log(2.0)) * np.array(w))
            + np.sum(stats.norm.logpdf(sol, 0.0, 1.5) - math.log(abs(np.linalg.det(a_mat)))))
    assert abs(la - want) <= 1e-6 * (1 + abs(want))
    with pytest.raises(cg.CodegenError):      # not lifted: an obs of a det node has no term to build
        cg.generate(lifted)


def test_meas_obs_of_a_transformed_target_and_likelihood_false():
    # compiler.ex:371-382: x = (y - b) / a, z = log x, the term is logpdf(exp z) + z - log|a|
    ir = cg.IR()
    ir.rv("m", "normal", dict(mu=0.0, sigma=1.0))
    ir.rv("lik_rv", "normal", dict(mu="m", sigma=1.0))
    ir.obs("lik", "lik_rv", 0.2)
    ir.rv("r", "gamma", dict(alpha=2.0, beta=1.5), transform="log")
    ir.meas_obs("r_obs", "r", [3.0, 5.0], ("affine", 2.0, 1.0))
    ir.rv("off_rv", "normal", dict(mu="m", sigma=0.1))
    ir.obs("off", "off_rv", [9.0, 9.0], likelihood=False)
    gen = cg.generate(ir)
    lp = _lp(gen, [0.1])[0]
    xs = np.array([1.0, 2.0])
    want = (stats.norm.logpdf(0.1) + stats.norm.logpdf(0.2, 0.1, 1.0)
            + np.sum(stats.gamma.logpdf(xs, 2.0, scale=1 / 1.5) + np.log(xs) - math.log(2.0)))
    assert abs(lp - want) <= 2e-5 * (1 + abs(want))
    assert gen.var_names == ["m"]             # off_rv is observed all the same (point_map.ex:124-137)


def test_censoring_is_not_applied_to_a_target_that_carries_a_transform():
    # compiler.ex:274 and :298-311 match {:rv, dist, params} only; a Weibull target written without
    # a transform gets :log from the first pass and lands in the 4-tuple clause (:325-334)
    def model(**opts):
        ir = cg.IR()
        ir.rv("k", "gamma", dict(alpha=2.0, beta=1.0))
        ir.rv("t_rv", "weibull", {"k": "k", "lambda": 2.0})
        ir.obs("t", "t_rv", [1.0, 2.5], **opts)
        return cg.generate(ir, rewrite_passes=True)
    a, b = model(censored="right"), model()
    assert _lp(a, [0.2])[0] == _lp(b, [0.2])[0]
    # an explicit "no transform" target keeps the survival term
    ir = cg.IR()
    ir.rv("k", "gamma", dict(alpha=2.0, beta=1.0), transform="log")
    ir.rv("t_rv", "weibull", {"k": "k", "lambda": 2.0})
    ir.obs("t", "t_rv", [1.0, 2.5], censored="right")
    k = math.exp(0.2)
    want = stats.gamma.logpdf(k, 2.0) + 0.2 + np.sum(stats.weibull_min.logsf([1.0, 2.5], k, scale=2.0))
    assert abs(_lp(cg.generate(ir), [0.2])[0] - want) <= 2e-5 * (1 + abs(want))


def test_dirichlet_observation_on_its_default_transform():
    ir = cg.IR()
    ir.rv("a", "gamma", dict(alpha=2.0, beta=1.0))
    ir.rv("th", "dirichlet", dict(alpha=[2.0, 3.0, 1.5]))
    ir.obs("th_obs", "th", [0.2, 0.5, 0.3])
    ir.rv("y_rv", "normal", dict(mu="a", sigma=1.0))
    ir.obs("y", "y_rv", 1.0)
    gen = cg.generate(ir, rewrite_passes=True)
    x = np.array([0.2, 0.5, 0.3])
    z = cg.inverse_stick_breaking(x)
    h = 1e-6
    jac = np.zeros((2, 2))
    for j in range(2):
        e = np.zeros(2); e[j] = h
        jac[:, j] = (cg.stick_breaking(z + e)[:2] - cg.stick_breaking(z - e)[:2]) / (2 * h)
    want = (stats.gamma.logpdf(math.exp(0.3), 2.0) + 0.3 + stats.norm.logpdf(1.0, math.exp(0.3), 1.0)
            + stats.dirichlet.logpdf(x, [2.0, 3.0, 1.5]) + math.log(abs(np.linalg.det(jac))))
    assert abs(_lp(gen, [0.3])[0] - want) <= 2e-5 * (1 + abs(want))


def test_builder_data_resolves_obs_data_refs():
    """Builder.data (builder.ex:19-21) + a Custom distribution whose params name "__obs_data"
    (compiler.ex:103-118): the tensor reaches the closure as constants of the data array, so two
    data sets of one shape share the generated text."""
    def model(y):
        def logpdf(o, x, p):           # sum_i N(y_i | x, sigma) over the data tensor, up to its constant
            terms = []
            for yi in p["y"]:
                z = o.div(o.sub(yi, x), p["sigma"])
                terms.append(o.mul(o.lit(-0.5), o.mul(z, z)))
            return o.sub(o.sum(terms), o.mul(o.lit(float(len(p["y"]))), o.log(p["sigma"])))
        ir = cg.IR().data(y)
        ir.rv("sigma", "half_cauchy", dict(scale=2.0), transform="log")
        ir.rv("m", "custom", dict(logpdf=logpdf, y="__obs_data", sigma="sigma"))
        return cg.generate(ir)
    y1, y2 = np.array([0.3, -1.2, 2.2, 0.9]), np.array([5.0, 4.0, 6.5, 5.5])
    g1, g2 = model(y1), model(y2)
    assert g1.header == g2.header and not np.array_equal(g1.data, g2.data)
    for gen, y in ((g1, y1), (g2, y2)):
        q = np.array([0.7, -0.2])                          # m, log sigma
        sg = math.exp(q[1])
        want = (stats.halfcauchy.logpdf(sg, scale=2.0) + q[1]
                + np.sum(-0.5 * ((y - q[0]) / sg) ** 2) - len(y) * math.log(sg))
        lp, grad = _lp(gen, q)
        assert abs(lp - want) <= 2e-6 * (1 + abs(want))
        assert abs(grad[0] - np.sum((y - q[0]) / sg ** 2)) <= 1e-9 * (1 + abs(grad[0]))
    with pytest.raises(cg.CodegenError):
        ir = cg.IR()
        ir.rv("m", "custom", dict(logpdf=lambda o, x, p: x, y="__obs_data"))
        cg.generate(ir)
    # a matrix arrives as rows
    ir = cg.IR().data([[1.0, 2.0], [3.0, 4.0]])
    ir.rv("m", "custom", dict(logpdf=lambda o, x, p: o.mul(o.neg(o.mul(x, x)), p["a"][1][0]), a="__obs_data"))
    assert abs(_lp(cg.generate(ir), [0.5])[0] + 0.75) < 1e-15


def test_json_front_door_carries_det_meas_obs_and_data(tmp_path):
    import json
    doc = {"rewrite": True, "data": [0.5, 1.5], "nodes": {
        "m": {"op": "rv", "dist": "normal", "params": {"mu": 0.0, "sigma": 2.0}},
        "s": {"op": "rv", "dist": "half_cauchy", "params": {"scale": 1.0}},              # default :log
        "y_rv": {"op": "rv", "dist": "normal", "params": {"mu": "m", "sigma": "s"}},
        "y": {"op": "obs", "target": "y_rv", "value": [0.1, 0.4, -0.3], "weight": [1.0, 2.0, 0.5]},
        "x": {"op": "rv", "dist": "normal", "params": {"mu": 1.0, "sigma": 0.5}},
        "ax": {"op": "det", "fun": "affine", "args": [2.0, -1.0, "x"]},
        "ax_obs": {"op": "obs", "target": "ax", "value": [0.5, 1.5]},
        "z": {"op": "rv", "dist": "normal", "params": {"mu": 0.0, "sigma": 1.0}},
        "z_obs": {"op": "meas_obs", "target": "z", "value": 0.7, "info": ["affine", 1.0, 0.2]}}}
    src = tmp_path / "m.json"
    src.write_text(json.dumps(doc))
    cg.main([str(src), str(tmp_path / "out"), "--no-build"])
    meta = json.loads((tmp_path / "out" / "model.json").read_text())
    assert meta["var_names"] == ["m", "s"] and meta["transforms"] == {"s": "log"}
    gen = cg.generate(cg.ir_from_json(doc), rewrite_passes=True)
    assert gen.digest == meta["digest"]
    q = np.array([0.2, -0.1])
    s_ = math.exp(q[1])
    w = np.array([1.0, 2.0, 0.5])
    want = (stats.norm.logpdf(q[0], 0, 2) + stats.halfcauchy.logpdf(s_, scale=1.0) + q[1]
            + np.sum(stats.norm.logpdf([0.1, 0.4, -0.3], q[0], s_) * w)
            + np.sum(stats.norm.logpdf((np.array([0.5, 1.5]) + 1.0) / 2.0, 1.0, 0.5) - math.log(2.0))
            + stats.norm.logpdf(0.5, 0.0, 1.0))
    assert abs(_lp(gen, q)[0] - want) <= 2e-6 * (1 + abs(want))
