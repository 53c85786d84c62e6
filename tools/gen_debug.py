#!/usr/bin/env python3
"""Development aid for generated models: where does the plug-in's warmup / leapfrog first differ
from the CPU checker?  python tools/gen_debug.py zoo"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import gen_checker as GC  # noqa: E402
import gen_models as GM  # noqa: E402
import oracle as O  # noqa: E402
from exmc_amd import codegen as cg, sampler  # noqa: E402


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "zoo"
    ir, init = {"zoo": (GM.zoo_ir(), GM.ZOO_INIT),
                "simple": (GM.simple_ir(), {"mu": 2.0, "sigma": 1.0})}[which]
    spec = cg.compile_ir(ir, default_init=init)
    comp = sampler.compile(spec)
    om = GC.model(spec.gen)
    q0 = spec.to_unconstrained(init)
    det = O.Cfg(1, 1)
    d = spec.d
    # wild leapfrog trajectories
    lp0, g0 = om.logp_grad(q0, det)
    rng = np.random.default_rng(0)
    for eps in (0.05, 0.5, 2.0, 8.0, 64.0):
        p = rng.normal(size=d)
        n = 24
        aq, ap, alp, ag = om.multi_step(q0, p, g0, eps, np.ones(d), n, det)
        hq = np.zeros((1, n, d)); hp = np.zeros((1, n, d)); hg = np.zeros((1, n, d)); hl = np.zeros((1, n))
        q1 = np.ascontiguousarray(q0[None, :]); p1 = np.ascontiguousarray(p[None, :])
        g1 = np.ascontiguousarray(g0[None, :])
        comp.check(comp.L.exmc_hip_multi_step_host(comp.h, _dp(q1), _dp(p1), _dp(g1), eps,
                                                   _dp(np.ones(d)), n, 1, 1, _dp(hq), _dp(hp),
                                                   _dp(hl), _dp(hg)))
        same = [np.array_equal(aq[i], hq[0, i], equal_nan=True) and
                np.array_equal(ag[i], hg[0, i], equal_nan=True) and
                (alp[i] == hl[0, i] or (np.isnan(alp[i]) and np.isnan(hl[0, i]))) for i in range(n)]
        first = same.index(False) if False in same else -1
        print("eps %-5g first differing step %d" % (eps, first))
        if first >= 0:
            i = first
            print("   oracle logp %r hip %r" % (alp[i], hl[0, i]))
            print("   oracle q", aq[i]); print("   hip    q", hq[0, i])
            print("   oracle g", ag[i]); print("   hip    g", hg[0, i])
            if i > 0:
                print("   previous q", aq[i - 1])
    for eps in (0.05, 0.3, 1.0):
        tuning = dict(epsilon=eps, inv_mass=np.ones(d))
        _, _, extra = sampler.sample_compiled_tuned(comp, tuning, init, dict(num_samples=40, seed=17),
                                                    num_chains=3)
        raw = extra["raw"]
        t, _ = O.sample_chains(om, 3, init_q=q0, num_warmup=0, num_samples=40, seed=17, cfg=det) \
            if False else (None, None)
        for c in range(3):
            to, _ = O.sample_tuned(om, eps, np.ones(d), init_q=q0, num_samples=40, seed=17 + 7919 * c,
                                   cfg=det)
            same = [np.array_equal(to["draws"][s], raw["draws"][c, s]) for s in range(40)]
            print("tuned eps %g chain %d first differing draw %d" % (
                eps, c, same.index(False) if False in same else -1))
    for nw in (1, 2, 3, 5, 8, 12, 20, 40, 80, 150):
        tun = sampler.warmup(comp, init, dict(num_warmup=nw, seed=17))
        st = O.warmup(om, q0, num_warmup=nw, seed=17, cfg=det)
        print("warmup %3d eps equal %s  (%.6g vs %.6g)  mass equal %s" % (
            nw, tun["epsilon"] == st.step_size, tun["epsilon"], st.step_size,
            np.array_equal(tun["inv_mass"], np.array(st.inv_mass[:d]))))


if __name__ == "__main__":
    main()
