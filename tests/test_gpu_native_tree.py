"""GPU parity for the NativeTree NIF seam: build_full_tree_bin batched over chains
through the C ABI vs the checker's restatement of native/exmc_tree/src/tree.rs (same numeric
contract: exmc_detmath), plus the structural expectations of test/native_tree_test.exs."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import oracle as O
from exmc_amd import native_tree

pytestmark = pytest.mark.gpu

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden",
                                   "reference_known_answers.json")))["native_tree"]


def _oracle_full_tree(q0, p0, g0, logp0, fq, fp, flp, fg, bq, bp, blp, bg, im, jlp0, max_depth, seed):
    L = O.lib()
    L.exo_nt_set_math_mode(1)
    d = len(q0)
    qo, go = np.zeros(d), np.zeros(d)
    r = O.TreeResult()
    a = [O.arr(x) for x in (q0, p0, g0, fq, fp, flp, fg, bq, bp, blp, bg, im)]
    L.exo_nt_build_full_tree(O.dptr(a[0]), O.dptr(a[1]), O.dptr(a[2]), float(logp0), O.dptr(a[3]),
                             O.dptr(a[4]), O.dptr(a[5]), O.dptr(a[6]), len(flp), O.dptr(a[7]),
                             O.dptr(a[8]), O.dptr(a[9]), O.dptr(a[10]), len(blp), O.dptr(a[11]),
                             float(jlp0), max_depth, d, int(seed), O.dptr(qo), O.dptr(go), C.byref(r))
    L.exo_nt_set_math_mode(0)
    return qo, go, r


def test_full_tree_fixture(hip):
    c = GOLD["full_tree"]
    k = lambda name: np.array(c[name])[None, :, None]  # noqa: E731
    res = native_tree.build_full_tree_bin(
        np.array([c["q0"]]), np.array([c["p0"]]), np.array([c["grad0"]]), [c["logp0"]],
        k("fwd_q"), k("fwd_p"), np.array([c["fwd_logp"]]), k("fwd_grad"),
        k("bwd_q"), k("bwd_p"), np.array([c["bwd_logp"]]), k("bwd_grad"),
        c["inv_mass"], [c["jlp0"]], c["max_depth"], 1, [c["seed"]])
    e = c["expect"]
    assert res["n_steps"][0] > e["n_steps_gt"] and res["accept_sum"][0] > e["accept_sum_gt"]
    assert e["depth_gt"] < res["depth"][0] <= e["depth_le"]
    qo, go, r = _oracle_full_tree(c["q0"], c["p0"], c["grad0"], c["logp0"], c["fwd_q"], c["fwd_p"],
                                  c["fwd_logp"], c["fwd_grad"], c["bwd_q"], c["bwd_p"],
                                  c["bwd_logp"], c["bwd_grad"], c["inv_mass"], c["jlp0"],
                                  c["max_depth"], c["seed"])
    assert (res["n_steps"][0], res["depth"][0], bool(res["divergent"][0])) == \
        (r.n_steps, r.depth, bool(r.divergent))
    assert res["accept_sum"][0] == r.accept_sum and res["logp"][0] == r.logp
    assert np.array_equal(res["q_bin"][0], qo) and np.array_equal(res["grad_bin"][0], go)


def test_full_tree_all_divergent_fixture(hip):
    c = GOLD["full_tree_divergent"]
    n = c["n"]
    full = lambda v: np.full((1, n, 1), v)  # noqa: E731
    res = native_tree.build_full_tree_bin(
        np.array([c["q0"]]), np.array([c["p0"]]), np.array([c["grad0"]]), [c["logp0"]],
        full(c["fwd_q_value"]), full(c["p_value"]), np.full((1, n), c["logp_value"]),
        full(c["grad_value"]), full(c["bwd_q_value"]), full(c["p_value"]),
        np.full((1, n), c["logp_value"]), full(c["grad_value"]), c["inv_mass"], [c["jlp0"]],
        c["max_depth"], 1, [c["seed"]])
    assert bool(res["divergent"][0]) and res["n_steps"][0] <= c["expect"]["n_steps_le"]


@pytest.mark.parametrize("budget,max_depth,eps", [(31, 5, 0.35), (15, 10, 0.5), (63, 6, 0.08)])
def test_full_tree_batched_bit_exact(hip, budget, max_depth, eps):
    """tree.ex:155-263 feeds the NIF two multi_step chains of `budget` states each; here 48 chains
    at once, every output identical to the checker (including trees cut short by the budget)."""
    L = O.lib()
    om = O.eight_schools()
    rng = np.random.default_rng(budget)
    Cn, d = 48, 10
    cfg = O.Cfg(1, 1)
    im = np.ascontiguousarray(rng.uniform(0.5, 2.0, size=d))
    q0 = rng.normal(size=(Cn, d)) * 0.5
    p0 = rng.normal(size=(Cn, d)) / np.sqrt(im)
    g0 = np.zeros((Cn, d)); logp0 = np.zeros(Cn); jlp0 = np.zeros(Cn)
    fwd = [np.zeros((Cn, budget, d)) for _ in range(3)] + [np.zeros((Cn, budget))]
    bwd = [np.zeros((Cn, budget, d)) for _ in range(3)] + [np.zeros((Cn, budget))]
    for c in range(Cn):
        logp0[c], g0[c] = om.logp_grad(q0[c], cfg)
        jlp0[c] = logp0[c] - L.exo_kinetic_energy(O.dptr(np.ascontiguousarray(p0[c])), O.dptr(im), d, cfg)
        aq, ap, alp, ag = om.multi_step(q0[c], p0[c], g0[c], eps, im, budget, cfg)
        fwd[0][c], fwd[1][c], fwd[2][c], fwd[3][c] = aq, ap, ag, alp
        aq, ap, alp, ag = om.multi_step(q0[c], p0[c], g0[c], -eps, im, budget, cfg)
        bwd[0][c], bwd[1][c], bwd[2][c], bwd[3][c] = aq, ap, ag, alp
    seeds = rng.integers(0, 10 ** 12, size=Cn).astype(np.uint64)   # trunc(uniform * 1e12), tree.ex:224
    res = native_tree.build_full_tree_bin(q0, p0, g0, logp0, fwd[0], fwd[1], fwd[3], fwd[2],
                                          bwd[0], bwd[1], bwd[3], bwd[2], im, jlp0, max_depth, d, seeds)
    depths = set()
    for c in range(Cn):
        qo, go, r = _oracle_full_tree(q0[c], p0[c], g0[c], logp0[c], fwd[0][c], fwd[1][c], fwd[3][c],
                                      fwd[2][c], bwd[0][c], bwd[1][c], bwd[3][c], bwd[2][c], im,
                                      jlp0[c], max_depth, seeds[c])
        assert (res["n_steps"][c], res["depth"][c], bool(res["divergent"][c])) == \
            (r.n_steps, r.depth, bool(r.divergent)), c
        assert res["accept_sum"][c] == r.accept_sum and res["logp"][c] == r.logp, c
        assert np.array_equal(res["q_bin"][c], qo) and np.array_equal(res["grad_bin"][c], go), c
        assert r.n_steps <= 2 ** r.depth - 1
        depths.add(r.depth)
    assert len(depths) > 1


def test_badarg_is_reported(hip):
    with pytest.raises(ValueError):
        native_tree.build_full_tree_bin(np.zeros((2, 3)), np.zeros((2, 2)), np.zeros((2, 3)),
                                        np.zeros(2), np.zeros((2, 1, 3)), np.zeros((2, 1, 3)),
                                        np.zeros((2, 1)), np.zeros((2, 1, 3)), np.zeros((2, 1, 3)),
                                        np.zeros((2, 1, 3)), np.zeros((2, 1)), np.zeros((2, 1, 3)),
                                        np.ones(3), np.zeros(2), 3, 3, [1, 2])


def _ms_chain(om, q, p, g, eps, im, n, cfg):
    aq, ap, alp, ag = om.multi_step(q, p, g, eps, im, n, cfg)
    return aq, ap, alp, ag


def test_incremental_trajectory_interface_bit_exact(hip):
    """The NIF's resource interface (init_trajectory_bin, get_endpoint_bin, build_and_merge_bin,
    is_terminated, get_result_bin; tree.ex:735-830 drives it one doubling at a time) for 40 chains
    at once against the checker's restatement of lib.rs / tree.rs driven the same way: chains
    terminate at different depths, terminated chains are skipped with depth -1."""
    L = O.lib()
    L.exo_nt_set_math_mode(1)
    try:
        om = O.eight_schools()
        rng = np.random.default_rng(11)
        Cn, d, eps, max_depth = 40, 10, 0.45, 6
        cfg = O.Cfg(1, 1)
        im = np.ascontiguousarray(rng.uniform(0.5, 2.0, size=d))
        q0 = rng.normal(size=(Cn, d)) * 0.5
        p0 = rng.normal(size=(Cn, d)) / np.sqrt(im)
        g0 = np.zeros((Cn, d)); logp0 = np.zeros(Cn); jlp0 = np.zeros(Cn)
        for c in range(Cn):
            logp0[c], g0[c] = om.logp_grad(q0[c], cfg)
            jlp0[c] = logp0[c] - L.exo_kinetic_energy(O.dptr(np.ascontiguousarray(p0[c])), O.dptr(im), d, cfg)
        T = native_tree.Trajectories(q0, p0, g0, logp0)
        refs = [L.exo_nt_init_trajectory(O.dptr(np.ascontiguousarray(q0[c])),
                                         O.dptr(np.ascontiguousarray(p0[c])),
                                         O.dptr(np.ascontiguousarray(g0[c])), float(logp0[c]), d)
                for c in range(Cn)]
        depth_now = np.zeros(Cn, np.int32)
        finished_at = set()
        for it in range(max_depth):
            term = T.is_terminated()
            assert [bool(L.exo_nt_is_terminated(r)) for r in refs] == term.tolist()
            if term.all():
                break
            go_right = rng.integers(0, 2, size=Cn).astype(np.int32)
            seeds = rng.integers(0, 10 ** 12, size=Cn).astype(np.uint64)
            eq, ep, eg = T.get_endpoint_bin(go_right)
            n = 1 << it
            aq = np.zeros((Cn, n, d)); ap = np.zeros((Cn, n, d)); ag = np.zeros((Cn, n, d))
            alp = np.zeros((Cn, n))
            depth = np.where(term, -1, depth_now).astype(np.int32)
            for c in range(Cn):
                oq, op_, og = np.zeros(d), np.zeros(d), np.zeros(d)
                L.exo_nt_get_endpoint(refs[c], int(go_right[c]), O.dptr(oq), O.dptr(op_), O.dptr(og))
                assert np.array_equal(oq, eq[c]) and np.array_equal(op_, ep[c]) and np.array_equal(og, eg[c])
                if term[c]:
                    finished_at.add(int(depth_now[c]))
                    continue
                assert depth_now[c] == it
                e = eps if go_right[c] else -eps
                aq[c], ap[c], alp[c], ag[c] = _ms_chain(om, eq[c], ep[c], eg[c], e, im, n, cfg)
                L.exo_nt_build_and_merge(refs[c], O.dptr(np.ascontiguousarray(aq[c])),
                                         O.dptr(np.ascontiguousarray(ap[c])),
                                         O.dptr(np.ascontiguousarray(alp[c])),
                                         O.dptr(np.ascontiguousarray(ag[c])), O.dptr(im),
                                         float(jlp0[c]), it, d, int(go_right[c]), int(seeds[c]))
            assert T.build_and_merge_bin(aq, ap, alp, ag, im, jlp0, depth, d, go_right, seeds) == "ok"
            depth_now = np.where(term, depth_now, depth_now + 1).astype(np.int32)
        res = T.get_result_bin()
        for c in range(Cn):
            qo, go = np.zeros(d), np.zeros(d)
            r = O.TreeResult()
            L.exo_nt_get_result(refs[c], O.dptr(qo), O.dptr(go), C.byref(r))
            assert (res["n_steps"][c], res["depth"][c], bool(res["divergent"][c])) == \
                (r.n_steps, r.depth, bool(r.divergent)), c
            assert res["accept_sum"][c] == r.accept_sum and res["logp"][c] == r.logp, c
            assert np.array_equal(res["q_bin"][c], qo) and np.array_equal(res["grad_bin"][c], go), c
            L.exo_nt_free(refs[c])
        assert len(set(res["depth"].tolist())) > 1   # chains stopped at different depths
    finally:
        L.exo_nt_set_math_mode(0)


@pytest.mark.parametrize("depth,eps", [(0, 0.3), (3, 0.3), (5, 0.6)])
def test_build_subtree_bin_bit_exact(hip, depth, eps):
    """build_subtree_bin (lib.rs:114-212) batched: every field of the subtree record, including
    subtrees that stop early on a U-turn or a divergence (large eps)."""
    L = O.lib()
    L.exo_nt_set_math_mode(1)
    try:
        om = O.eight_schools()
        rng = np.random.default_rng(depth + 3)
        Cn, d, n = 33, 10, 1 << depth
        cfg = O.Cfg(1, 1)
        im = np.ascontiguousarray(rng.uniform(0.5, 2.0, size=d))
        aq = np.zeros((Cn, n, d)); ap = np.zeros((Cn, n, d)); ag = np.zeros((Cn, n, d))
        alp = np.zeros((Cn, n)); jlp0 = np.zeros(Cn)
        going_right = rng.integers(0, 2, size=Cn).astype(np.int32)
        seeds = rng.integers(0, 10 ** 12, size=Cn).astype(np.uint64)
        for c in range(Cn):
            q0 = rng.normal(size=d) * 0.5
            p0 = rng.normal(size=d) / np.sqrt(im)
            lp0, g0 = om.logp_grad(q0, cfg)
            jlp0[c] = lp0 - L.exo_kinetic_energy(O.dptr(np.ascontiguousarray(p0)), O.dptr(im), d, cfg)
            e = eps if going_right[c] else -eps
            aq[c], ap[c], alp[c], ag[c] = _ms_chain(om, q0, p0, g0, e, im, n, cfg)
        res = native_tree.build_subtree_bin(aq, ap, alp, ag, im, jlp0, depth, d, going_right, seeds)
        keys = ["q_left_bin", "p_left_bin", "grad_left_bin", "q_right_bin", "p_right_bin",
                "grad_right_bin", "q_prop_bin", "grad_prop_bin", "rho_bin"]
        stopped_early = 0
        for c in range(Cn):
            vecs, sc, it = np.zeros(9 * d), np.zeros(3), np.zeros(4, np.int32)
            L.exo_nt_build_subtree(O.dptr(np.ascontiguousarray(aq[c])), O.dptr(np.ascontiguousarray(ap[c])),
                                   O.dptr(np.ascontiguousarray(alp[c])), O.dptr(np.ascontiguousarray(ag[c])),
                                   O.dptr(im), float(jlp0[c]), depth, d, int(going_right[c]),
                                   int(seeds[c]), O.dptr(vecs), O.dptr(sc),
                                   it.ctypes.data_as(C.POINTER(C.c_int)))
            for i, k in enumerate(keys):
                assert np.array_equal(res[k][c], vecs[i * d:(i + 1) * d]), (k, c)
            assert (res["logp_prop"][c], res["log_sum_weight"][c], res["accept_sum"][c]) == tuple(sc), c
            assert (res["n_steps"][c], int(res["divergent"][c]), int(res["turning"][c]),
                    res["depth"][c]) == tuple(int(x) for x in it), c
            stopped_early += int(res["n_steps"][c] < n)
        if depth == 5:
            assert stopped_early > 0
    finally:
        L.exo_nt_set_math_mode(0)


def test_trajectory_fixture_init_and_endpoints(hip):
    """test/native_tree_test.exs:38-61: a fresh trajectory returns its start state from both ends."""
    c = GOLD["init_get"]
    T = native_tree.Trajectories([c["q"]], [c["p"]], [c["grad"]], [c["logp"]])
    r = T.get_result_bin()
    assert r["q_bin"][0].tolist() == c["q"] and r["grad_bin"][0].tolist() == c["grad"]
    assert (r["logp"][0], r["n_steps"][0], bool(r["divergent"][0]), r["accept_sum"][0], r["depth"][0]) == \
        (c["logp"], 0, False, 0.0, 0)
    assert not T.is_terminated()[0]
    for go_right in (1, 0):
        q, p, g = T.get_endpoint_bin(go_right)
        assert q[0].tolist() == c["q"] and p[0].tolist() == c["p"] and g[0].tolist() == c["grad"]


@pytest.mark.parametrize("name", ["depth0", "depth1", "divergent"])
def test_trajectory_fixture_build_and_merge(hip, name):
    """test/native_tree_test.exs:63-178 through the GPU entry points: depth-0 merge => n_steps 1 and
    depth 1; depth-1 => n_steps 2; logp -1e10 => divergent and terminated."""
    c = GOLD[name]
    d = c["d"]
    T = native_tree.Trajectories([c["q"]], [c["p"]], [c["grad"]], [c["logp"]])
    n = len(c["all_logp"])
    shp = lambda k: np.array(c[k], dtype=np.float64).reshape(1, n, d)  # noqa: E731
    assert T.build_and_merge_bin(shp("all_q"), shp("all_p"), np.array([c["all_logp"]]), shp("all_grad"),
                                 c["inv_mass"], c["jlp0"], c["depth"], d, int(c["go_right"]),
                                 c["seed"]) == "ok"
    r = T.get_result_bin()
    e = c["expect"]
    if "n_steps" in e:
        assert r["n_steps"][0] == e["n_steps"]
    if "depth" in e:
        assert r["depth"][0] == e["depth"]
    if "divergent" in e:
        assert bool(r["divergent"][0]) == e["divergent"]
    if e.get("terminated"):
        assert T.is_terminated()[0]


def test_merge_invariants_on_the_gpu_subtree_records(hip):
    """test/nuts/statham_merge_test.exs:135-170 (merge_subtrees postconditions) on records produced
    by the GPU: a depth-3 subtree against the two depth-2 subtrees over the halves of the same
    states: log_sum_weight = log_sum_exp of the children (1e-10, as the reference asserts), n_steps
    additive, depth = max + 1, rho additive (exactly: the merge adds the two vectors), divergent
    monotonic, accept_sum additive; endpoints are the outer endpoints of the children."""
    om = O.eight_schools()
    rng = np.random.default_rng(99)
    Cn, d, n, eps = 24, 10, 8, 0.05     # small step: no U-turn, no early end
    cfg = O.Cfg(1, 1)
    L = O.lib()
    im = np.ascontiguousarray(rng.uniform(0.5, 2.0, size=d))
    aq = np.zeros((Cn, n, d)); ap = np.zeros((Cn, n, d)); ag = np.zeros((Cn, n, d))
    alp = np.zeros((Cn, n)); jlp0 = np.zeros(Cn)
    for c in range(Cn):
        q0 = rng.normal(size=d) * 0.5
        p0 = rng.normal(size=d) / np.sqrt(im)
        lp0, g0 = om.logp_grad(q0, cfg)
        jlp0[c] = lp0 - L.exo_kinetic_energy(O.dptr(np.ascontiguousarray(p0)), O.dptr(im), d, cfg)
        aq[c], ap[c], alp[c], ag[c] = _ms_chain(om, q0, p0, g0, eps, im, n, cfg)
    seeds = rng.integers(0, 10 ** 12, size=Cn).astype(np.uint64)
    full = native_tree.build_subtree_bin(aq, ap, alp, ag, im, jlp0, 3, d, 1, seeds)
    a = native_tree.build_subtree_bin(aq[:, :4], ap[:, :4], alp[:, :4], ag[:, :4], im, jlp0, 2, d, 1, seeds)
    b = native_tree.build_subtree_bin(aq[:, 4:], ap[:, 4:], alp[:, 4:], ag[:, 4:], im, jlp0, 2, d, 1, seeds)
    assert not full["turning"].any() and not full["divergent"].any()
    for c in range(Cn):
        lse = L.exo_log_sum_exp(float(a["log_sum_weight"][c]), float(b["log_sum_weight"][c]), 1)
        assert abs(full["log_sum_weight"][c] - lse) < 1.0e-10
    assert np.array_equal(full["n_steps"], a["n_steps"] + b["n_steps"]) and (full["n_steps"] == 8).all()
    assert np.array_equal(full["depth"], np.maximum(a["depth"], b["depth"]) + 1)
    assert np.array_equal(full["rho_bin"], a["rho_bin"] + b["rho_bin"])
    assert np.array_equal(full["divergent"], a["divergent"] | b["divergent"])
    assert np.allclose(full["accept_sum"], a["accept_sum"] + b["accept_sum"], rtol=1e-14, atol=0)
    assert np.array_equal(full["q_left_bin"], a["q_left_bin"]) and np.array_equal(full["p_left_bin"], a["p_left_bin"])
    assert np.array_equal(full["q_right_bin"], b["q_right_bin"]) and np.array_equal(full["p_right_bin"], b["p_right_bin"])
    assert np.array_equal(full["q_left_bin"], aq[:, 0]) and np.array_equal(full["q_right_bin"], aq[:, 7])


@pytest.mark.parametrize("budget,max_depth,eps", [(31, 5, 0.35), (15, 10, 0.5)])
def test_full_tree_on_poisoned_trajectories_bit_exact(hip, budget, max_depth, eps):
    """build_full_tree_bin consumes PRE-COMPUTED trajectories (tree.ex:155-263), so what it does with a NaN
    log-density, an infinite momentum or a 1e300 position is the tree code's own business (the divergence
    test, log-sum-exp, the U-turn products, the multinomial pick): the same 48 chains as above with a few
    entries of the forward / backward arrays replaced by such values -- every output against the checker's
    restatement of tree.rs."""
    L = O.lib()
    om = O.eight_schools()
    rng = np.random.default_rng(1000 + budget)
    Cn, d = 48, 10
    cfg = O.Cfg(1, 1)
    im = np.ascontiguousarray(rng.uniform(0.5, 2.0, size=d))
    q0 = rng.normal(size=(Cn, d)) * 0.5
    p0 = rng.normal(size=(Cn, d)) / np.sqrt(im)
    g0 = np.zeros((Cn, d)); logp0 = np.zeros(Cn); jlp0 = np.zeros(Cn)
    fwd = [np.zeros((Cn, budget, d)) for _ in range(3)] + [np.zeros((Cn, budget))]
    bwd = [np.zeros((Cn, budget, d)) for _ in range(3)] + [np.zeros((Cn, budget))]
    for c in range(Cn):
        logp0[c], g0[c] = om.logp_grad(q0[c], cfg)
        jlp0[c] = logp0[c] - L.exo_kinetic_energy(O.dptr(np.ascontiguousarray(p0[c])), O.dptr(im), d, cfg)
        aq, ap, alp, ag = om.multi_step(q0[c], p0[c], g0[c], eps, im, budget, cfg)
        fwd[0][c], fwd[1][c], fwd[2][c], fwd[3][c] = aq, ap, ag, alp
        aq, ap, alp, ag = om.multi_step(q0[c], p0[c], g0[c], -eps, im, budget, cfg)
        bwd[0][c], bwd[1][c], bwd[2][c], bwd[3][c] = aq, ap, ag, alp
    poison = [np.nan, np.inf, -np.inf, 1e300, -1e300, 1e200, -1e200, 0.0, -0.0, 5e-324]
    n_poisoned = 0
    for c in range(Cn):
        if c % 4 == 0:
            continue                                     # every fourth chain stays clean
        for _ in range(int(rng.integers(1, 4))):
            side = fwd if rng.integers(2) else bwd
            which = int(rng.integers(4))
            step = int(rng.integers(min(budget, 8)))     # early steps: the tree reaches them
            v = poison[int(rng.integers(len(poison)))]
            if which == 3:
                side[3][c, step] = v
            else:
                side[which][c, step, int(rng.integers(d))] = v
            n_poisoned += 1
    assert n_poisoned > Cn // 2
    seeds = rng.integers(0, 10 ** 12, size=Cn).astype(np.uint64)
    res = native_tree.build_full_tree_bin(q0, p0, g0, logp0, fwd[0], fwd[1], fwd[3], fwd[2],
                                          bwd[0], bwd[1], bwd[3], bwd[2], im, jlp0, max_depth, d, seeds)
    n_div = 0
    for c in range(Cn):
        qo, go, r = _oracle_full_tree(q0[c], p0[c], g0[c], logp0[c], fwd[0][c], fwd[1][c], fwd[3][c],
                                      fwd[2][c], bwd[0][c], bwd[1][c], bwd[3][c], bwd[2][c], im,
                                      jlp0[c], max_depth, seeds[c])
        assert (res["n_steps"][c], res["depth"][c], bool(res["divergent"][c])) == \
            (r.n_steps, r.depth, bool(r.divergent)), c
        assert np.array_equal(np.array([res["accept_sum"][c], res["logp"][c]]),
                              np.array([r.accept_sum, r.logp]), equal_nan=True), c
        assert np.array_equal(res["q_bin"][c], qo, equal_nan=True), c
        assert np.array_equal(res["grad_bin"][c], go, equal_nan=True), c
        n_div += bool(r.divergent)
    assert 0 < n_div < Cn
