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
