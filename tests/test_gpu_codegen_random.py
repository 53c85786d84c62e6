"""Randomly drawn Builder models on the GPU: the generated plug-in (plates across 16 lanes and
one lane per chain) against the same generated text on the CPU, bit for bit -- value and gradient
at random points, then a short warmup + sampling run."""
import ctypes as C

import numpy as np
import pytest

import gen_checker as GC
import oracle as O
from exmc_amd import codegen as cg, sampler
from test_codegen_random import _random_ir

pytestmark = pytest.mark.gpu

SEEDS = [1, 4, 7]


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


@pytest.mark.parametrize("seed", SEEDS)
def test_random_generated_model_bit_exact(hip, seed):
    ir, rng = _random_ir(seed)
    init = {}
    for id_, n in ir.nodes.items():
        if n["op"] == "rv" and not id_.startswith("y_") and id_ != "z":
            init[id_] = 1.0 if id_ == "b_scale" else 0.1
    spec = cg.compile_ir(ir, name="gen_random_%d" % seed, default_init=init)
    comp = sampler.compile(spec)
    for lanes in (16, 1):
        om = GC.model(spec.gen, lanes)
        cfg = O.Cfg(1, lanes)
        n = 65
        q = np.ascontiguousarray(rng.normal(size=(n, spec.d)))
        q[0] = spec.to_unconstrained(init)
        lp = np.zeros(n)
        g = np.zeros((n, spec.d))
        comp.check(comp.L.exmc_hip_logp_grad_host(comp.h, _dp(q), n, lanes, _dp(lp), _dp(g)))
        for c in range(n):
            olp, og = om.logp_grad(q[c], cfg)
            assert olp == lp[c], (seed, lanes, c, olp, lp[c])
            assert np.array_equal(og, g[c]), (seed, lanes, c)
        opts = dict(num_warmup=80, num_samples=60, seed=3, lanes_per_chain=lanes)
        trace, stats = sampler.sample_compiled(comp, init, opts)
        t, st = O.sample(om, init_q=spec.to_unconstrained(init), num_warmup=80, num_samples=60,
                         seed=3, cfg=cfg)
        assert stats["step_size"] == st.step_size
        assert np.array_equal(stats["raw"]["draws"][0], t["draws"])
        assert np.array_equal(stats["raw"]["n_steps"][0], t["n_steps"])
