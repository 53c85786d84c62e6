"""Builder-IR test models for the generator (exmc_amd/codegen.py), shared by CPU and GPU tests."""
import numpy as np

from exmc_amd import codegen as cg


simple_ir = cg.simple_ir
eight_schools_ir = cg.eight_schools_ir


def zoo_ir(seed=5):
    """Every covered distribution and transform at least once; d = 9."""
    rng = np.random.default_rng(seed)
    ir = cg.IR()
    ir.rv("a_loc", "cauchy", dict(loc=0.5, scale=2.0))
    ir.rv("b_scale", "half_normal", dict(sigma=2.0), transform="softplus")
    ir.rv("c_df", "exponential", {"lambda": 0.2}, transform="log")
    ir.rv("d_p", "normal", dict(mu=0.0, sigma=1.5), transform="logit")
    ir.rv("e_lap", "laplace", dict(mu="a_loc", b="b_scale"))
    ir.rv("f_ln", "lognormal", dict(mu=0.1, sigma=0.7), transform="log")
    ir.rv("g_t", "student_t", dict(df="c_df", loc="a_loc", scale="f_ln"))
    ir.rv("h_hc", "half_cauchy", dict(scale="b_scale"), transform="log")
    ir.rv("i_centered", "normal", dict(mu="a_loc", sigma=1.0))   # one ref only: no NCP
    ir.rv("t_obs_rv", "student_t", dict(df=4.0, loc="e_lap", scale="h_hc"))
    ir.obs("t_obs", "t_obs_rv", rng.normal(size=7) * 2.0)
    ir.rv("n_obs_rv", "normal", dict(mu="i_centered", sigma=np.abs(rng.normal(size=5)) + 0.5))
    ir.obs("n_obs", "n_obs_rv", rng.normal(size=5))
    ir.rv("bern_rv", "bernoulli", dict(p="d_p"))
    ir.obs("bern", "bern_rv", (rng.uniform(size=9) < 0.4).astype(float))
    ir.rv("l_obs_rv", "laplace", dict(mu="g_t", b=1.3))
    ir.obs("l_obs", "l_obs_rv", 0.25)
    return ir


ZOO_INIT = dict(a_loc=0.3, b_scale=1.2, c_df=5.0, d_p=0.4, e_lap=-0.2, f_ln=0.9, g_t=0.1, h_hc=1.5,
                i_centered=0.0)


def walk_ir(seed=3):
    """Round-2 generator coverage in one model (d = 12): a GaussianRandomWalk latent path driven by a
    log-scale rv, an MvNormal block with a constant covariance, a Custom-distribution likelihood
    written as a closure over the declarative op set (the shape of validate_posteriordb.exs:279-295),
    a vector obs whose mean is the random walk, and two meas_obs terms (affine, matmul)."""
    rng = np.random.default_rng(seed)
    ir = cg.IR()
    ir.rv("sigma", "exponential", {"lambda": 2.0}, transform="log")
    ir.rv("w", "gaussian_random_walk", dict(sigma="sigma", steps=6))
    cov = np.array([[1.0, 0.3, 0.1], [0.3, 2.0, 0.2], [0.1, 0.2, 1.5]])
    ir.rv("m", "mv_normal", dict(mu=[0.1, -0.2, 0.3], cov=cov))
    ir.rv("tau", "half_cauchy", dict(scale=2.0), transform="log")
    ir.rv("c", "normal", dict(mu=0.0, sigma=3.0))
    ir.rv("y_rv", "normal", dict(mu="w", sigma=0.5))
    ir.obs("y", "y_rv", rng.normal(size=6) * 0.4)

    def lik(o, x, p):
        # sum_j Normal(x_j; c + tau * m_j, s_j) up to the constant the posteriordb script drops
        terms = []
        for xj, mj, sj in zip(x, p["m"], p["s"]):
            z = o.div(o.sub(xj, o.add(p["c"], o.mul(p["tau"], mj))), sj)
            terms.append(o.sub(o.mul(o.lit(-0.5), o.mul(z, z)), o.log(sj)))
        return o.sum(terms)
    ir.rv("z_rv", "custom", dict(logpdf=lik, m="m", c="c", tau="tau", s=[1.0, 2.0, 0.7]))
    ir.obs("z", "z_rv", [0.4, -1.1, 2.0])
    ir.rv("k_rv", "normal", dict(mu=1.0, sigma=2.0))
    ir.meas_obs("k", "k_rv", 3.0, ("affine", 2.0, 1.0))
    ir.rv("v_rv", "normal", dict(mu=0.0, sigma=1.0))
    ir.meas_obs("v", "v_rv", [0.5, -0.25], ("matmul", [[2.0, 1.0], [0.0, 3.0]]))
    return ir


WALK_INIT = dict(sigma=0.5, w=[0.0, 0.1, 0.0, -0.1, 0.05, 0.0], m=[0.0, 0.0, 0.0], tau=1.0, c=0.2)


def simplex_ir(seed=7):
    """The distributions added at the end of round 2 (d = 8): a Dirichlet rv on the 4-simplex behind
    the stick-breaking transform, Gamma / Beta / Weibull / Uniform01 free rvs with their default
    transforms, a Poisson likelihood whose rate is the Gamma rv, Gamma- and Weibull-distributed
    observations, and a Dirichlet-distributed observation."""
    rng = np.random.default_rng(seed)
    ir = cg.IR()
    ir.rv("theta", "dirichlet", dict(alpha=[2.0, 1.5, 1.0, 3.0]), transform="stick_breaking")
    ir.rv("rate", "gamma", dict(alpha=3.0, beta=2.0), transform="log")
    ir.rv("p", "beta", dict(alpha=2.0, beta=5.0), transform="logit")
    ir.rv("k", "weibull", {"k": 1.5, "lambda": 2.0}, transform="log")
    ir.rv("u", "uniform01", {}, transform="logit")
    ir.rv("cnt_rv", "poisson", dict(mu="rate"))
    ir.obs("cnt", "cnt_rv", rng.poisson(1.5, size=6).astype(float))
    ir.rv("wait_rv", "weibull", {"k": "k", "lambda": 1.3})
    ir.obs("wait", "wait_rv", rng.weibull(1.5, size=5) * 1.3 + 0.05)
    ir.rv("g_rv", "gamma", dict(alpha=2.5, beta="rate"))
    ir.obs("g", "g_rv", rng.gamma(2.5, 0.6, size=4) + 0.05)
    ir.rv("b_rv", "bernoulli", dict(p="p"))
    ir.obs("b", "b_rv", (rng.uniform(size=7) < 0.3).astype(float))
    ir.rv("mix_rv", "dirichlet", dict(alpha=[4.0, 2.0, 1.0, 1.0]))
    ir.obs("mix", "mix_rv", [0.4, 0.3, 0.2, 0.1])
    ir.rv("n_rv", "normal", dict(mu="u", sigma=0.5))
    ir.obs("n", "n_rv", 0.6)
    return ir


SIMPLEX_INIT = dict(theta=[0.25, 0.25, 0.25, 0.25], rate=1.0, p=0.3, k=1.2, u=0.5)


def survival_ir(seed=11):
    """The Builder.obs meta and the remaining distributions (d = 6): right-censored Weibull survival
    times and exact ones, left- / right- / interval-censored Normal measurements, a weighted and
    masked vector obs, reduce :mean and :logsumexp, an observation of a :log-transformed rv, and a
    two-component Normal mixture likelihood whose component means are free, and a TruncatedNormal
    likelihood (exmc_erf)."""
    rng = np.random.default_rng(seed)
    ir = cg.IR()
    ir.rv("k", "gamma", dict(alpha=2.0, beta=1.0), transform="log")
    ir.rv("lam", "lognormal", dict(mu=0.5, sigma=0.8), transform="log")
    ir.rv("m", "normal", dict(mu=0.0, sigma=2.0))
    ir.rv("s", "half_normal", dict(sigma=1.5), transform="log")
    ir.rv("m1", "normal", dict(mu=-1.0, sigma=1.0))
    ir.rv("m2", "normal", dict(mu=2.0, sigma=1.0))
    ir.rv("t_rv", "weibull", {"k": "k", "lambda": "lam"})
    ir.obs("t_exact", "t_rv", rng.weibull(1.5, size=5) * 2.0 + 0.1)
    ir.obs("t_cens", "t_rv", [2.5, 3.0, 3.0], censored="right")
    ir.rv("x_rv", "normal", dict(mu="m", sigma="s"))
    ir.obs("x_left", "x_rv", [-0.5, 0.2], censored="left")
    ir.obs("x_right", "x_rv", 1.7, censored="right")
    ir.obs("x_int", "x_rv", dict(lower=[-1.0, 0.0], upper=[0.5, 2.0]), censored="interval")
    ir.obs("x_w", "x_rv", rng.normal(size=6), weight=[1.0, 0.5, 2.0, 1.0, 0.25, 3.0],
           mask=[True, True, False, True, True, False])
    ir.obs("x_mean", "x_rv", rng.normal(size=4) + 0.3, reduce="mean", weight=2.0)
    ir.obs("x_lse", "x_rv", [0.1, 0.9, -0.4], reduce="logsumexp")
    ir.rv("pos_rv", "lognormal", dict(mu="m", sigma=0.7), transform="log")
    ir.obs("pos", "pos_rv", [0.8, 1.9])
    ir.rv("mix_rv", "mixture", dict(components=["normal", "normal"],
                                    params=[dict(mu="m1", sigma=0.6), dict(mu="m2", sigma=1.1)],
                                    weights=[0.35, 0.65]))
    ir.obs("mix", "mix_rv", rng.normal(size=7) * 1.5 + 0.5)
    ir.rv("tn_rv", "truncated_normal", dict(mu="m", sigma="s", lower=-2.0, upper=3.0))
    ir.obs("tn", "tn_rv", [-1.2, 0.4, 2.6])
    return ir


SURVIVAL_INIT = dict(k=1.2, lam=1.5, m=0.1, s=1.0, m1=-0.8, m2=1.7)


# ---- the three larger BASELINE configs as Builder IR (exmc_amd/codegen.py sv_ir / radon_ir /
# logistic_ir) next to their hand-written kinds: same variable names, so an init map or a point of
# one is a point of the other after a permutation ----
def sv_returns(seed=3):
    return np.random.default_rng(seed).normal(size=100) * 0.02


def baseline_pair(which):
    """(ir, ncp, hand-written ModelSpec, lanes of the generated layout)"""
    from exmc_amd import models
    if which == "sv":
        r = sv_returns()
        return cg.sv_ir(r), False, models.sv(r), 64
    if which == "logistic":
        X, y = models.logistic_data()
        return cg.logistic_ir(X, y), True, models.logistic(X, y), 16
    if which == "radon":
        spec = models.radon()
        J = 85
        d = spec.data
        start = d[J:2 * J + 1].astype(int)
        n = start[-1]
        names = [v for v in spec.var_names if v.startswith("alpha_raw")]
        ir = cg.radon_ir(d[:J], start, d[2 * J + 1:2 * J + 1 + n], d[2 * J + 1 + n:], names=names)
        return ir, False, spec, 64
    raise ValueError(which)


def to_spec_order(gen, spec):
    """idx with q_generated = q_handwritten[idx] (the generated kernel order is the flat order)."""
    return [spec.var_names.index(n) for n in gen.var_names]
