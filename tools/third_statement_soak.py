#!/usr/bin/env python3
"""CPU soak of the checker against its plain-Python third statements (tests/py_tree.py, py_sampler.py,
py_native_tree.py) with RANDOM settings -- the fixed-setting versions are tests/test_*_third_statement.py:

  tree     one Tree.build transition: model, start, mass, step size over three decades, depth cap 1..9
  sampler  a whole sample/3 chain: num_warmup around the window schedule's edges, target_accept, depth cap, start
           (random / given / far away), warm_start in a fifth of the runs
  native   the crate's full tree on pre-computed trajectories: budget, depth cap, step size, poisoned entries

Both sides share the model arithmetic, the leapfrog and the random stream (ctypes into the checker); the control flow
is written twice. Every output bit for bit, NaN for NaN.      python tools/third_statement_soak.py <first> <last>
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle as O  # noqa: E402
import py_native_tree as PN  # noqa: E402
import py_sampler as PS  # noqa: E402
import py_tree as PT  # noqa: E402

NW = [0, 1, 2, 9, 24, 25, 26, 49, 50, 51, 74, 75, 76, 99, 100, 101, 120, 149, 150, 151, 199, 200, 201, 230, 300]
POISON = [float("nan"), float("inf"), float("-inf"), 1e300, -1e300, 1e200, -1e200, 0.0, -0.0, 5e-324]
MODELS = [("eight_schools", O.eight_schools, 0.4), ("simple", O.simple, 0.5), ("std_normal5", lambda: O.std_normal(5), 0.9)]
KEYS = ("draws", "logp", "tree_depth", "n_steps", "divergent", "accept_prob", "energy")


def _same(a, b):
    return np.array_equal(np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64), equal_nan=True)


def tree(rng, m, eps0):
    L = O.lib()
    eps = float(eps0 * 10.0 ** rng.uniform(-1.7, 1.3))
    max_depth = int(rng.integers(1, 10))
    q = rng.normal(size=m.d) * float(rng.choice([0.3, 1.0, 3.0, 10.0]))
    im = np.ascontiguousarray(10.0 ** rng.uniform(-0.7, 0.7, size=m.d))
    lp, g = m.logp_grad(q)
    r0 = O.Rng()
    L.exo_rng_seed(C.byref(r0), int(rng.integers(0, 2 ** 31)))
    p = np.array([L.exo_rng_normal(C.byref(r0), 0) for _ in range(m.d)]) / np.sqrt(im)
    jlp0 = lp - L.exo_kinetic_energy(O.dptr(p), O.dptr(im), m.d, O.Cfg(0, 1))
    ra, rb = O.Rng(), O.Rng()
    C.memmove(C.byref(ra), C.byref(r0), C.sizeof(O.Rng))
    C.memmove(C.byref(rb), C.byref(r0), C.sizeof(O.Rng))
    qo, go, res = m.tree_build(q, p, lp, g, eps, im, max_depth, ra, jlp0, O.Cfg(0, 1))
    py = PT.build(m, q, p, lp, g, eps, im, max_depth, rb, jlp0)
    ok = (res.depth, res.n_steps, bool(res.divergent)) == (py["depth"], py["n_steps"], py["divergent"]) and \
        _same([res.accept_sum, res.logp], [py["accept_sum"], py["logp"]]) and _same(qo, py["q"]) and _same(go, py["grad"])
    return ok, "tree eps=%.3g depth<=%d: depth %d, %d steps%s" % (eps, max_depth, res.depth, res.n_steps,
                                                                 ", divergent" if res.divergent else "")


def sampler(rng, m, _eps0):
    nw = int(rng.choice(NW))
    ns = int(rng.integers(1, 10))
    depth = int(rng.integers(2, 9))
    ta = float(rng.choice([0.5, 0.65, 0.8, 0.9, 0.95]))
    seed = int(rng.integers(0, 2 ** 31))
    start = str(rng.choice(["random", "given", "far"]))
    q0 = None if start == "random" else rng.normal(size=m.d) * (0.5 if start == "given" else 30.0)
    if rng.integers(5) == 0:
        pe = float(10.0 ** rng.uniform(-2.5, 0.0))
        pim = np.ascontiguousarray(10.0 ** rng.uniform(-0.7, 0.7, size=m.d))
        q0 = np.zeros(m.d) if q0 is None else q0
        t, st = O.sample_warm(m, pe, pim, q0, num_warmup=nw, num_samples=ns, max_tree_depth=depth, target_accept=ta, seed=seed)
        p, ps = PS.sample(m, q0, num_warmup=nw, num_samples=ns, max_tree_depth=depth, target_accept=ta, seed=seed,
                          warm_start=(pe, pim))
        what = "warm"
    else:
        t, st = O.sample(m, q0, num_warmup=nw, num_samples=ns, max_tree_depth=depth, target_accept=ta, seed=seed)
        p, ps = PS.sample(m, q0, num_warmup=nw, num_samples=ns, max_tree_depth=depth, target_accept=ta, seed=seed)
        what = "cold"
    ok = st.step_size == ps["step_size"] and _same(st.inv_mass[:m.d], ps["inv_mass"]) and \
        (what == "warm" or st.divergences == ps["divergences"]) and all(_same(t[k], p[k]) for k in KEYS if k in p)
    return ok, "sample/3 %s nw=%d ns=%d depth<=%d accept=%g start=%s: eps %.4g" % (what, nw, ns, depth, ta, start, st.step_size)


def native(rng, m, eps0):
    L = O.lib()
    L.exo_nt_set_math_mode(0)
    eps = float(eps0 * 10.0 ** rng.uniform(-1.5, 1.3))
    budget = int(rng.integers(1, 70))
    max_depth = int(rng.integers(1, 9))
    d = m.d
    q = rng.normal(size=d) * 0.7
    im = np.ascontiguousarray(10.0 ** rng.uniform(-0.5, 0.5, size=d))
    p = rng.normal(size=d) / np.sqrt(im)
    lp, g = m.logp_grad(q)
    jlp0 = lp - sum(0.5 * a * b * a for a, b in zip(p, im))
    ch = {}
    for name, e in (("fwd", eps), ("bwd", -eps)):
        aq, ap, alp, ag = m.multi_step(q, p, g, e, im, budget)
        ch[name] = dict(q=aq, p=ap, logp=alp, g=ag)
    poisoned = bool(rng.integers(3) == 0)
    if poisoned:
        for _ in range(int(rng.integers(1, 4))):
            side = ch["fwd" if rng.integers(2) else "bwd"]
            key = ("q", "p", "logp", "g")[int(rng.integers(4))]
            step = int(rng.integers(min(budget, 8)))
            v = POISON[int(rng.integers(len(POISON)))]
            if key == "logp":
                side[key][step] = v
            else:
                side[key][step, int(rng.integers(d))] = v
    seed = int(rng.integers(0, 10 ** 12))
    qo, go, r = np.zeros(d), np.zeros(d), O.TreeResult()
    f, b = ch["fwd"], ch["bwd"]
    L.exo_nt_build_full_tree(O.dptr(q), O.dptr(p), O.dptr(g), lp, O.dptr(f["q"]), O.dptr(f["p"]), O.dptr(f["logp"]),
                             O.dptr(f["g"]), budget, O.dptr(b["q"]), O.dptr(b["p"]), O.dptr(b["logp"]), O.dptr(b["g"]),
                             budget, O.dptr(im), jlp0, max_depth, d, seed, O.dptr(qo), O.dptr(go), C.byref(r))
    py = PN.build_full_tree(q, p, g, lp, f, b, list(im), jlp0, max_depth, seed)
    ok = (r.depth, r.n_steps, bool(r.divergent)) == (py["depth"], py["n_steps"], py["divergent"]) and \
        _same([r.accept_sum, r.logp], [py["accept_sum"], py["logp"]]) and _same(qo, py["q"]) and _same(go, py["grad"])
    return ok, "crate tree eps=%.3g budget %d depth<=%d%s: depth %d, %d steps%s" % (
        eps, budget, max_depth, ", poisoned" if poisoned else "", r.depth, r.n_steps, ", divergent" if r.divergent else "")


def one(seed, cache):
    rng = np.random.default_rng(seed)
    name, factory, eps0 = MODELS[int(rng.integers(len(MODELS)))]
    if name not in cache:
        cache[name] = factory()
    f = (tree, sampler, native)[int(rng.integers(3))]
    ok, info = f(rng, cache[name], eps0)
    if not ok:
        raise AssertionError("differs: %s %s" % (name, info))
    return "%s %s" % (name, info)


def main():
    lo, hi = int(sys.argv[1]), int(sys.argv[2])
    cache, bad = {}, []
    for seed in range(lo, hi):
        try:
            print(seed, "ok", one(seed, cache), flush=True)
        except Exception as e:   # noqa: BLE001
            print(seed, "FAIL", repr(e)[:600], flush=True)
            bad.append(seed)
    print("failed seeds:", bad)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
