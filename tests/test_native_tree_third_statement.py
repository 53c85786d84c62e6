"""The checker's restatement of the Rust crate (oracle/exmc_oracle.c exo_nt_build_full_tree / exo_nt_build_subtree,
libm mode) against a third statement of native/exmc_tree/src/tree.rs in plain Python (tests/py_native_tree.py): leapfrog
chains of a model pre-computed forwards and backwards as Tree.build_full_tree_nif does (tree.ex:155-263), then the whole
doubling tree on both sides from the same seed -- every output bit; and single subtrees with every field of the record
build_subtree_bin returns (lib.rs:115-212)."""
import ctypes as C

import numpy as np
import pytest

import oracle as O
import py_native_tree as PN


def _chains(m, q, p, g, eps, im, n):
    out = {}
    for name, e in (("fwd", eps), ("bwd", -eps)):
        aq, ap, alp, ag = m.multi_step(q, p, g, e, im, n)
        out[name] = dict(q=aq, p=ap, logp=alp, g=ag)
    return out


@pytest.mark.parametrize("eps,max_depth,budget", [(0.3, 5, 32), (0.05, 4, 16), (1.6, 6, 64), (0.3, 6, 7), (30.0, 5, 32)])
def test_full_tree_agrees_with_the_third_statement(eps, max_depth, budget):
    L = O.lib()
    L.exo_nt_set_math_mode(0)
    m = O.eight_schools()
    rng = np.random.default_rng(int(eps * 1000) + budget)
    kinds = dict(div=0, turn=0, budget=0, full=0)
    for k in range(30):
        q = rng.normal(size=10) * 0.7
        im = 0.5 + rng.uniform(size=10)
        p = rng.normal(size=10) / np.sqrt(im)
        lp, g = m.logp_grad(q)
        jlp0 = lp - sum(0.5 * a * b * a for a, b in zip(p, im))
        ch = _chains(m, q, p, g, eps, im, budget)
        seed = 1000 + k
        qo, go, r = np.zeros(10), np.zeros(10), O.TreeResult()
        f, b = ch["fwd"], ch["bwd"]
        L.exo_nt_build_full_tree(O.dptr(q), O.dptr(p), O.dptr(g), lp, O.dptr(f["q"]), O.dptr(f["p"]), O.dptr(f["logp"]),
                                 O.dptr(f["g"]), budget, O.dptr(b["q"]), O.dptr(b["p"]), O.dptr(b["logp"]), O.dptr(b["g"]),
                                 budget, O.dptr(np.ascontiguousarray(im)), jlp0, max_depth, 10, seed, O.dptr(qo), O.dptr(go),
                                 C.byref(r))
        py = PN.build_full_tree(q, p, g, lp, f, b, list(im), jlp0, max_depth, seed)
        assert (r.depth, r.n_steps, bool(r.divergent)) == (py["depth"], py["n_steps"], py["divergent"]), k
        assert r.accept_sum == py["accept_sum"] and r.logp == py["logp"], k
        assert np.array_equal(qo, np.array(py["q"])) and np.array_equal(go, np.array(py["grad"])), k
        kinds["div"] += bool(r.divergent)
        kinds["full"] += r.depth == max_depth
        kinds["turn"] += (not r.divergent) and r.depth < max_depth
    SEEN[(eps, max_depth, budget)] = kinds


SEEN = {}


def test_the_cases_cover_divergent_turning_capped_and_budget_limited_trees():
    if len(SEEN) < 5:
        pytest.skip("the parametrised cases ran in another process")
    assert SEEN[(30.0, 5, 32)]["div"] >= 25            # every tree diverges (a divergent leaf keeps the new state)
    assert SEEN[(0.05, 4, 16)]["full"] >= 20           # the depth cap
    assert SEEN[(0.3, 5, 32)]["turn"] + SEEN[(1.6, 6, 64)]["turn"] >= 20
    assert SEEN[(0.3, 6, 7)]["turn"] >= 1              # 7 states per direction: doubling 3 never fits (tree.rs:300-305)


@pytest.mark.parametrize("depth,right", [(0, True), (2, True), (3, False), (4, True)])
def test_subtree_record_agrees_with_the_third_statement(depth, right):
    """build_subtree_bin's record (lib.rs:115-212): both endpoints, the proposal, rho, the weights, the flags."""
    L = O.lib()
    L.exo_nt_set_math_mode(0)
    m = O.eight_schools()
    rng = np.random.default_rng(depth + 5 * right)
    for k in range(20):
        q = rng.normal(size=10) * 0.6
        im = 0.5 + rng.uniform(size=10)
        p = rng.normal(size=10) / np.sqrt(im)
        lp, g = m.logp_grad(q)
        eps = (0.45 if right else -0.45) * (1.0 + (k % 3))
        jlp0 = lp - sum(0.5 * a * b * a for a, b in zip(p, im))
        n = 1 << depth
        aq, ap, alp, ag = m.multi_step(q, p, g, eps, im, n)
        vecs, sc, ints = np.zeros(9 * 10), np.zeros(3), (C.c_int * 4)()
        seed = 77 + k
        L.exo_nt_build_subtree(O.dptr(aq), O.dptr(ap), O.dptr(alp), O.dptr(ag), O.dptr(np.ascontiguousarray(im)), jlp0, depth,
                               10, int(right), seed, O.dptr(vecs), O.dptr(sc), ints)
        s = PN.subtree(dict(q=aq, p=ap, logp=alp, g=ag), list(im), jlp0, depth, right, [0], PN.Rng(seed))
        got = vecs.reshape(9, 10)
        for row, key in enumerate(("ql", "pl", "gl", "qr", "pr", "gr", "qp", "gp", "rho")):
            assert np.array_equal(got[row], np.array(s[key])), (k, key)
        assert (sc[0], sc[1], sc[2]) == (s["lpp"], s["lsw"], s["acc"]), k
        assert list(ints) == [s["n"], int(s["div"]), int(s["turn"]), s["depth"]], k



POISON = [float("nan"), float("inf"), float("-inf"), 1e300, -1e300, 1e200, -1e200, 0.0, -0.0, 5e-324]


@pytest.mark.parametrize("eps,max_depth,budget", [(0.3, 5, 32), (0.05, 4, 16), (1.6, 6, 64)])
def test_full_tree_on_poisoned_trajectories(eps, max_depth, budget):
    """The same comparison on trajectories with NaN / infinite / 1e300 entries in a few early states: what
    tree.rs does with them (f64 comparisons that are false for NaN, exp / ln on infinities, the U-turn dot
    products, the multinomial pick) is restated twice -- in C and in Python -- and both must agree in every
    output bit, NaN for NaN."""
    L = O.lib()
    L.exo_nt_set_math_mode(0)
    m = O.eight_schools()
    rng = np.random.default_rng(7000 + budget)
    n_div = n_cases = 0
    for k in range(40):
        q = rng.normal(size=10) * 0.7
        im = 0.5 + rng.uniform(size=10)
        p = rng.normal(size=10) / np.sqrt(im)
        lp, g = m.logp_grad(q)
        jlp0 = lp - sum(0.5 * a * b * a for a, b in zip(p, im))
        ch = _chains(m, q, p, g, eps, im, budget)
        for _ in range(int(rng.integers(1, 4))):
            side = ch["fwd" if rng.integers(2) else "bwd"]
            key = ("q", "p", "logp", "g")[int(rng.integers(4))]
            step = int(rng.integers(min(budget, 8)))       # early states: the tree reaches them
            v = POISON[int(rng.integers(len(POISON)))]
            if key == "logp":
                side[key][step] = v
            else:
                side[key][step, int(rng.integers(10))] = v
        seed = 3000 + k
        qo, go, r = np.zeros(10), np.zeros(10), O.TreeResult()
        f, b = ch["fwd"], ch["bwd"]
        L.exo_nt_build_full_tree(O.dptr(q), O.dptr(p), O.dptr(g), lp, O.dptr(f["q"]), O.dptr(f["p"]), O.dptr(f["logp"]),
                                 O.dptr(f["g"]), budget, O.dptr(b["q"]), O.dptr(b["p"]), O.dptr(b["logp"]), O.dptr(b["g"]),
                                 budget, O.dptr(np.ascontiguousarray(im)), jlp0, max_depth, 10, seed, O.dptr(qo), O.dptr(go),
                                 C.byref(r))
        py = PN.build_full_tree(q, p, g, lp, f, b, list(im), jlp0, max_depth, seed)
        assert (r.depth, r.n_steps, bool(r.divergent)) == (py["depth"], py["n_steps"], py["divergent"]), k
        assert np.array_equal(np.array([r.accept_sum, r.logp]), np.array([py["accept_sum"], py["logp"]]), equal_nan=True), k
        assert np.array_equal(qo, np.array(py["q"]), equal_nan=True), k
        assert np.array_equal(go, np.array(py["grad"]), equal_nan=True), k
        n_div += bool(r.divergent)
        n_cases += 1
    assert 0 < n_div < n_cases
