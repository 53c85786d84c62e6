"""Randomly drawn Builder models through the generator: the plate (16-lane) form against the
scalar form, and the scalar gradient against central differences. Catches wrong lane
assignments, masks and adjoint routing that the three fixed models would not."""
import numpy as np
import pytest

import gen_checker as GC
from exmc_amd import codegen as cg


def _random_ir(seed):
    rng = np.random.default_rng(seed)
    ir = cg.IR()
    # hyper-parameters: a location and a positive scale
    if rng.integers(0, 2):
        ir.rv("a_loc", "normal", dict(mu=0.3, sigma=2.0))
    else:
        ir.rv("a_loc", "cauchy", dict(loc=0.3, scale=2.0))
    scale_dist = str(rng.choice(["half_cauchy", "half_normal", "exponential", "lognormal"]))
    sp = {"half_cauchy": dict(scale=2.0), "half_normal": dict(sigma=2.0),
          "exponential": {"lambda": 0.5}, "lognormal": dict(mu=0.0, sigma=0.8)}[scale_dist]
    ir.rv("b_scale", scale_dist, sp, transform=str(rng.choice(["log", "softplus"])))
    k = int(rng.integers(2, 9))                     # plate size
    centered = bool(rng.integers(0, 2))
    like = str(rng.choice(["normal", "student_t", "laplace", "cauchy"]))
    for j in range(k):
        name = "g_%d" % j
        if centered:
            ir.rv(name, "normal", dict(mu="a_loc", sigma=float(rng.uniform(0.5, 2.0))))
        else:
            ir.rv(name, "normal", dict(mu="a_loc", sigma="b_scale"))   # non-centred rewrite
        y = float(rng.normal() * 2.0)
        s = float(rng.uniform(0.5, 3.0))
        if like == "normal":
            ir.rv("y_%d" % j, "normal", dict(mu=name, sigma=s))
        elif like == "student_t":
            ir.rv("y_%d" % j, "student_t",
                  dict(df=float(rng.uniform(2.5, 8.0)), loc=name, scale="b_scale"))
        elif like == "laplace":
            ir.rv("y_%d" % j, "laplace", dict(mu=name, b=s))
        else:
            ir.rv("y_%d" % j, "cauchy", dict(loc=name, scale="b_scale"))
        ir.obs("yo_%d" % j, "y_%d" % j, y)
    if rng.integers(0, 2):
        n = int(rng.integers(3, 30))
        ir.rv("z", "normal", dict(mu="a_loc", sigma="b_scale"))
        ir.obs("z_obs", "z", rng.normal(size=n))
    assert len(ir.nodes) <= cg.MAX_NODES_SORTED
    return ir, rng


@pytest.mark.parametrize("seed", range(12))
def test_random_models_plate_form_equals_scalar_form(seed):
    ir, rng = _random_ir(seed)
    gen = cg.generate(ir)
    assert gen.vec is not None and gen.lanes == 16
    for _ in range(6):
        q = rng.normal(size=gen.d) * 0.9
        a, ga = GC.logp_grad(gen, q, 1)
        b, gb = GC.logp_grad(gen, q, 16)
        assert np.isfinite(a)
        assert abs(a - b) <= 1e-12 * max(1.0, abs(a))
        np.testing.assert_allclose(ga, gb, rtol=1e-11, atol=1e-11)
    q = rng.normal(size=gen.d) * 0.5
    lp, g = GC.logp_grad(gen, q, 1)
    fd = np.zeros(gen.d)
    for i in range(gen.d):
        e = np.zeros(gen.d)
        e[i] = 1e-6
        fd[i] = (GC.logp_grad(gen, q + e, 1)[0] - GC.logp_grad(gen, q - e, 1)[0]) / 2e-6
    np.testing.assert_allclose(g, fd, rtol=5e-6, atol=5e-6)
