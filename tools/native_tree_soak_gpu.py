#!/usr/bin/env python3
"""GPU soak of the NativeTree seam (build_full_tree_bin and build_subtree_bin through the C ABI, batched over chains)
with RANDOM settings: the model the trajectories come from (eight schools d=10, the d=2 plumbing model, a standard
normal d=4), budget 1..80 states per direction, depth cap 0..10, step size over two and a half decades, mass entries,
the batch size, and in a third of the runs a few entries of the pre-computed trajectories replaced by NaN / +-inf /
1e300 -- against the checker's restatement of native/exmc_tree/src/tree.rs, every output bit for bit (NaN for NaN).

    gpurun -- 'python tools/native_tree_soak_gpu.py <first seed> <last seed>'
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle as O  # noqa: E402
from exmc_amd import _lib, native_tree  # noqa: E402

POISON = [np.nan, np.inf, -np.inf, 1e300, -1e300, 1e200, -1e200, 0.0, -0.0, 5e-324]
MODELS = [("eight_schools", O.eight_schools, 0.35), ("simple", O.simple, 0.4), ("std_normal4", lambda: O.std_normal(4), 0.8)]


def _same(a, b):
    return np.array_equal(np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64), equal_nan=True)


def full_tree(rng, om, d, eps, cfg, L):
    Cn = int(rng.integers(1, 40))
    budget = int(rng.integers(1, 81))
    max_depth = int(rng.integers(1, 11))
    im = np.ascontiguousarray(10.0 ** rng.uniform(-0.5, 0.5, size=d))
    q0 = rng.normal(size=(Cn, d)) * float(rng.choice([0.3, 1.0, 3.0]))
    p0 = rng.normal(size=(Cn, d)) / np.sqrt(im)
    g0 = np.zeros((Cn, d)); logp0 = np.zeros(Cn); jlp0 = np.zeros(Cn)
    fwd = [np.zeros((Cn, budget, d)) for _ in range(3)] + [np.zeros((Cn, budget))]
    bwd = [np.zeros((Cn, budget, d)) for _ in range(3)] + [np.zeros((Cn, budget))]
    for c in range(Cn):
        logp0[c], g0[c] = om.logp_grad(q0[c], cfg)
        jlp0[c] = logp0[c] - L.exo_kinetic_energy(O.dptr(np.ascontiguousarray(p0[c])), O.dptr(im), d, cfg)
        for arrs, e in ((fwd, eps), (bwd, -eps)):
            aq, ap, alp, ag = om.multi_step(q0[c], p0[c], g0[c], e, im, budget, cfg)
            arrs[0][c], arrs[1][c], arrs[2][c], arrs[3][c] = aq, ap, ag, alp
    poisoned = bool(rng.integers(3) == 0)
    if poisoned:
        for c in range(Cn):
            for _ in range(int(rng.integers(0, 4))):
                side = fwd if rng.integers(2) else bwd
                which, step = int(rng.integers(4)), int(rng.integers(min(budget, 8)))
                v = POISON[int(rng.integers(len(POISON)))]
                if which == 3:
                    side[3][c, step] = v
                else:
                    side[which][c, step, int(rng.integers(d))] = v
    seeds = rng.integers(0, 10 ** 12, size=Cn).astype(np.uint64)
    res = native_tree.build_full_tree_bin(q0, p0, g0, logp0, fwd[0], fwd[1], fwd[3], fwd[2],
                                          bwd[0], bwd[1], bwd[3], bwd[2], im, jlp0, max_depth, d, seeds)
    n_div = 0
    depths = set()
    for c in range(Cn):
        qo, go = np.zeros(d), np.zeros(d)
        r = O.TreeResult()
        a = [O.arr(x) for x in (q0[c], p0[c], g0[c], fwd[0][c], fwd[1][c], fwd[3][c], fwd[2][c], bwd[0][c], bwd[1][c],
                                bwd[3][c], bwd[2][c], im)]
        L.exo_nt_build_full_tree(O.dptr(a[0]), O.dptr(a[1]), O.dptr(a[2]), float(logp0[c]), O.dptr(a[3]), O.dptr(a[4]),
                                 O.dptr(a[5]), O.dptr(a[6]), budget, O.dptr(a[7]), O.dptr(a[8]), O.dptr(a[9]),
                                 O.dptr(a[10]), budget, O.dptr(a[11]), float(jlp0[c]), max_depth, d, int(seeds[c]),
                                 O.dptr(qo), O.dptr(go), C.byref(r))
        if (res["n_steps"][c], res["depth"][c], bool(res["divergent"][c])) != (r.n_steps, r.depth, bool(r.divergent)):
            raise AssertionError("chain %d: steps / depth / divergent" % c)
        if not (_same([res["accept_sum"][c], res["logp"][c]], [r.accept_sum, r.logp]) and _same(res["q_bin"][c], qo)
                and _same(res["grad_bin"][c], go)):
            raise AssertionError("chain %d: floats" % c)
        n_div += bool(r.divergent)
        depths.add(int(r.depth))
    return "full tree: %d chains, budget %d, depth<=%d%s: depths %s, %d divergent" % (
        Cn, budget, max_depth, ", poisoned" if poisoned else "", sorted(depths), n_div)


def subtree(rng, om, d, eps, cfg, L):
    Cn = int(rng.integers(1, 40))
    depth = int(rng.integers(0, 7))
    n = 1 << depth
    im = np.ascontiguousarray(10.0 ** rng.uniform(-0.5, 0.5, size=d))
    aq = np.zeros((Cn, n, d)); ap = np.zeros((Cn, n, d)); ag = np.zeros((Cn, n, d))
    alp = np.zeros((Cn, n)); jlp0 = np.zeros(Cn)
    going_right = rng.integers(0, 2, size=Cn).astype(np.int32)
    seeds = rng.integers(0, 10 ** 12, size=Cn).astype(np.uint64)
    for c in range(Cn):
        q0 = rng.normal(size=d) * 0.7
        p0 = rng.normal(size=d) / np.sqrt(im)
        lp0, g0 = om.logp_grad(q0, cfg)
        jlp0[c] = lp0 - L.exo_kinetic_energy(O.dptr(np.ascontiguousarray(p0)), O.dptr(im), d, cfg)
        aq[c], ap[c], alp[c], ag[c] = om.multi_step(q0, p0, g0, eps if going_right[c] else -eps, im, n, cfg)
    res = native_tree.build_subtree_bin(aq, ap, alp, ag, im, jlp0, depth, d, going_right, seeds)
    keys = ["q_left_bin", "p_left_bin", "grad_left_bin", "q_right_bin", "p_right_bin", "grad_right_bin", "q_prop_bin",
            "grad_prop_bin", "rho_bin"]
    early = 0
    for c in range(Cn):
        vecs, sc, it = np.zeros(9 * d), np.zeros(3), np.zeros(4, np.int32)
        L.exo_nt_build_subtree(O.dptr(np.ascontiguousarray(aq[c])), O.dptr(np.ascontiguousarray(ap[c])),
                               O.dptr(np.ascontiguousarray(alp[c])), O.dptr(np.ascontiguousarray(ag[c])), O.dptr(im),
                               float(jlp0[c]), depth, d, int(going_right[c]), int(seeds[c]), O.dptr(vecs), O.dptr(sc),
                               it.ctypes.data_as(C.POINTER(C.c_int)))
        for i, k in enumerate(keys):
            if not _same(res[k][c], vecs[i * d:(i + 1) * d]):
                raise AssertionError("chain %d: %s" % (c, k))
        if not _same([res["logp_prop"][c], res["log_sum_weight"][c], res["accept_sum"][c]], sc):
            raise AssertionError("chain %d: scalars" % c)
        if (res["n_steps"][c], int(res["divergent"][c]), int(res["turning"][c]), res["depth"][c]) != tuple(int(x) for x in it):
            raise AssertionError("chain %d: ints" % c)
        early += int(res["n_steps"][c] < n)
    return "subtree: %d chains, depth %d: %d stopped early" % (Cn, depth, early)


def one(seed):
    rng = np.random.default_rng(seed)
    L = O.lib()
    name, factory, eps0 = MODELS[int(rng.integers(len(MODELS)))]
    om = factory()
    eps = float(eps0 * 10.0 ** rng.uniform(-1.5, 1.0))
    cfg = O.Cfg(1, 1)
    L.exo_nt_set_math_mode(1)
    try:
        f = full_tree if rng.integers(3) else subtree
        return "%s eps=%.3g %s" % (name, eps, f(rng, om, om.d, eps, cfg, L))
    finally:
        L.exo_nt_set_math_mode(0)


def main():
    lo, hi = int(sys.argv[1]), int(sys.argv[2])
    _lib.load()
    bad = []
    for seed in range(lo, hi):
        try:
            print(seed, "ok", one(seed), flush=True)
        except Exception as e:   # noqa: BLE001
            print(seed, "FAIL", repr(e)[:600], flush=True)
            bad.append(seed)
    print("failed seeds:", bad)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
