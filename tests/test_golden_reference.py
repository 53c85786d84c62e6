"""CPU tests: the oracle against every known-answer literal and invariant the reference's own
tests hold for the hot path (tests/golden/reference_known_answers.json cites each source line).
Both numeric modes of the oracle must satisfy them."""
import ctypes as C
import json
import math
import os

import numpy as np
import pytest

import oracle as O

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden",
                                   "reference_known_answers.json")))
MODES = [0, 1]


def _dist(L, fn, args, mode):
    f = getattr(L, "exo_dist_" + fn)
    return f(*[float(a) for a in args], mode)


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("case", GOLD["dist_logpdf"], ids=lambda c: c["cite"])
def test_dist_logpdf_literals(case, mode):
    got = _dist(O.lib(), case["fn"], case["args"], mode)
    assert abs(got - case["expect"]) <= case["tol"], case["cite"]


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("case", GOLD["lgamma"], ids=lambda c: c["cite"])
def test_lanczos_lgamma(case, mode):
    assert abs(O.lib().exo_lgamma_lanczos(case["x"], mode) - case["expect"]) <= case["tol"]


@pytest.mark.parametrize("mode", MODES)
def test_hierarchical_logp(mode):
    h = GOLD["hierarchical_logp"]
    got = sum(_dist(O.lib(), fn, args, mode) for fn, args in h["terms"])
    expect = -0.5 * (math.log(2 * math.pi) + 2 * math.log(10.0) + (2.0 / 10.0) ** 2) \
        - 0.5 * (math.log(2 * math.pi) + (3.0 - 2.0) ** 2)
    assert abs(got - expect) <= h["tol"]


def test_normal_gradient_is_minus_x():
    g = GOLD["normal_gradient"]
    m = O.std_normal(1)
    _, grad = m.logp_grad([g["x"]])
    assert abs(grad[0] - g["expect"]) <= g["tol"]


@pytest.mark.parametrize("lanes", [1, 2])
def test_kinetic_energy(lanes):
    k = GOLD["kinetic_energy"]
    p, im = O.arr(k["p"]), O.arr(k["inv_mass"])
    ke = O.lib().exo_kinetic_energy(O.dptr(p), O.dptr(im), 2, O.Cfg(0, lanes))
    assert abs(ke - k["expect"]) <= k["tol"]


@pytest.mark.parametrize("mode", MODES)
def test_leapfrog_energy_conservation(mode):
    e = GOLD["leapfrog"]["energy_conservation"]
    m = O.std_normal(1)
    cfg = O.Cfg(mode, 1)
    L = O.lib()
    r = O.Rng()
    L.exo_rng_seed(C.byref(r), 42)
    q = np.array([e["q0"]])
    p = np.array([L.exo_rng_normal(C.byref(r), mode)])
    im = np.array([1.0])
    lp, g = m.logp_grad(q, cfg)
    h0 = lp - L.exo_kinetic_energy(O.dptr(p), O.dptr(im), 1, cfg)
    for _ in range(e["steps"]):
        q, p, lp, g, jlp = m.leapfrog(q, p, g, e["epsilon"], im, cfg)
    assert abs(jlp - h0) < e["max_abs_dH"]


def test_leapfrog_reversibility():
    e = GOLD["leapfrog"]["reversibility"]
    m = O.std_normal(1)
    q0 = np.array([e["q0"]])
    _, g0 = m.logp_grad(q0)
    p0 = np.array([0.7312])
    im = np.array([1.0])
    q1, p1, _, g1, _ = m.leapfrog(q0, p0, g0, e["epsilon"], im)
    q2, _, _, _, _ = m.leapfrog(q1, -p1, g1, e["epsilon"], im)
    assert abs(q2[0] - q0[0]) <= e["tol"]


def test_multi_step_prefix_property():
    """test/nuts_test.exs:417-474: multi_step(32) prefix == multi_step(16) (1e-10; here exact)."""
    m = O.eight_schools()
    rng = np.random.default_rng(3)
    q = rng.normal(size=10) * 0.3
    p = rng.normal(size=10)
    im = np.ones(10)
    _, g = m.logp_grad(q)
    a = m.multi_step(q, p, g, 0.1, im, 32)
    b = m.multi_step(q, p, g, 0.1, im, 16)
    for x, y in zip(a, b):
        assert np.array_equal(x[:16], y)


def test_welford_and_regularisation():
    w = GOLD["welford"]
    L = O.lib()
    S = np.array(w["samples"])
    st = O.Welford()
    L.exo_welford_init(C.byref(st), 2)
    for s in S:
        s = O.arr(s)
        L.exo_welford_update(C.byref(st), O.dptr(s))
    assert np.allclose(list(st.mean[:2]), S.mean(axis=0), atol=w["tol"])
    im = np.zeros(2)
    L.exo_welford_finalize(C.byref(st), O.dptr(im))
    n = len(S)
    alpha = 5.0 / (n + 5.0)
    expect = (1 - alpha) * S.var(axis=0) * n / (n - 1) + alpha * 1e-3
    assert np.allclose(im, expect, atol=w["tol"])
    # n < 3 => identity
    st = O.Welford()
    L.exo_welford_init(C.byref(st), 3)
    im3 = np.zeros(3)
    L.exo_welford_finalize(C.byref(st), O.dptr(im3))
    assert list(im3) == w["identity_below_3"]["expect"]
    for s in w["identity_below_3"]["samples"]:
        s = O.arr(s)
        L.exo_welford_update(C.byref(st), O.dptr(s))
    L.exo_welford_finalize(C.byref(st), O.dptr(im3))
    assert list(im3) == w["identity_below_3"]["expect"]
    # variance floor + regularisation
    st = O.Welford()
    L.exo_welford_init(C.byref(st), 2)
    s = O.arr(w["floor"]["sample"])
    for _ in range(w["floor"]["repeat"]):
        L.exo_welford_update(C.byref(st), O.dptr(s))
    L.exo_welford_finalize(C.byref(st), O.dptr(im))
    assert np.allclose(im, 10.0 / 15.0 * 1e-6 + 5.0 / 15.0 * 1e-3, atol=w["floor"]["tol"])


def _run_da(init, accept, n):
    L = O.lib()
    s = O.DA()
    L.exo_da_init(C.byref(s), init[0], init[1])
    for _ in range(n):
        L.exo_da_update(C.byref(s), accept)
    return s


def test_dual_averaging():
    d = GOLD["dual_averaging"]
    lo = _run_da(**d["low"])
    hi = _run_da(**d["high"])
    assert math.exp(lo.log_epsilon) < math.exp(hi.log_epsilon)
    fin = _run_da(**d["finalize"])
    eps = O.lib().exo_da_finalize(C.byref(fin))
    assert eps > 0 and math.isfinite(eps)
    # step_size.ex:13-30: log_epsilon_bar starts at log(eps), mu = log(10 eps)
    s = O.DA()
    O.lib().exo_da_init(C.byref(s), 0.5, 0.8)
    assert s.log_epsilon_bar == math.log(0.5) and s.mu == math.log(5.0) and s.m == 0
    # one update by hand (step_size.ex:35-44)
    O.lib().exo_da_update(C.byref(s), 0.6)
    eta = 1.0 / 11.0
    hbar = eta * (0.8 - 0.6)
    assert s.h_bar == (1.0 - eta) * 0.0 + hbar
    assert s.log_epsilon == math.log(5.0) - math.sqrt(1) / 0.05 * s.h_bar
    assert s.log_epsilon_bar == 1.0 * s.log_epsilon + 0.0 * math.log(0.5)


def test_window_schedule():
    w = GOLD["windows"]
    L = O.lib()
    st = (C.c_int * 32)()
    en = (C.c_int * 32)()
    n = L.exo_build_windows(w["from"], w["to"], w["base"], st, en, 32)
    assert [[st[i], en[i]] for i in range(n)] == w["expect"]
    assert L.exo_build_windows(10, 10, 25, st, en, 32) == 0
    n = L.exo_build_windows(0, 30, 25, st, en, 32)      # remaining <= 1.5*w => one window
    assert [(st[i], en[i]) for i in range(n)] == [(0, 30)]


@pytest.mark.parametrize("lanes", [1, 2])
@pytest.mark.parametrize("case", GOLD["uturn"], ids=lambda c: c["cite"])
def test_uturn_rule(case, lanes):
    a = [O.arr(case[k]) for k in ("rho", "p_left", "p_right", "inv_mass")]
    got = O.lib().exo_check_uturn(*[O.dptr(x) for x in a], 2, O.Cfg(0, lanes))
    assert bool(got) == case["expect"]


def test_uturn_properties():
    """test/nuts/statham_merge_test.exs:275-322: aligned => no turn, reversed => turn, rho=0 => no."""
    rng = np.random.default_rng(0)
    L = O.lib()
    for _ in range(200):
        d = int(rng.integers(1, 12))
        p = rng.normal(size=d)
        im = rng.uniform(0.1, 10, size=d)
        z = np.zeros(d)
        cfg = O.Cfg(0, 1)
        assert not L.exo_check_uturn(O.dptr(p), O.dptr(p), O.dptr(p), O.dptr(im), d, cfg)
        mp = -p
        assert L.exo_check_uturn(O.dptr(p), O.dptr(p), O.dptr(mp), O.dptr(im), d, cfg)
        assert not L.exo_check_uturn(O.dptr(z), O.dptr(p), O.dptr(mp), O.dptr(im), d, cfg)


@pytest.mark.parametrize("mode", MODES)
def test_log_sum_exp(mode):
    L = O.lib()
    for c in GOLD["log_sum_exp"]:
        b = -math.inf if c["b"] == "-inf" else c["b"]
        assert abs(L.exo_log_sum_exp(c["a"], b, mode) - c["expect"]) <= c["tol"]
    # tree.ex:1600-1601 sentinel
    assert L.exo_log_sum_exp(-math.inf, -math.inf, mode) == -1.0e300
    assert L.exo_log_sum_exp(-1.0e300, -1.0e300, mode) == -1.0e300
    rng = np.random.default_rng(1)
    for a, b in rng.normal(0, 30, size=(200, 2)):
        ref = np.logaddexp(a, b)
        assert abs(L.exo_log_sum_exp(a, b, mode) - ref) <= 1e-12 * max(1.0, abs(ref))


# ---- NativeTree NIF structure (test/native_tree_test.exs) ----
def _nt_traj(c):
    L = O.lib()
    q, p, g = O.arr(c["q"]), O.arr(c["p"]), O.arr(c["grad"])
    return L.exo_nt_init_trajectory(O.dptr(q), O.dptr(p), O.dptr(g), c["logp"], len(q))


def _nt_result(t, d):
    L = O.lib()
    q, g = np.zeros(d), np.zeros(d)
    r = O.TreeResult()
    L.exo_nt_get_result(t, O.dptr(q), O.dptr(g), C.byref(r))
    return q, g, r


def test_nif_init_and_endpoints():
    c = GOLD["native_tree"]["init_get"]
    L = O.lib()
    t = _nt_traj(c)
    q, g, r = _nt_result(t, 3)
    assert list(q) == c["q"] and list(g) == c["grad"] and r.logp == c["logp"]
    assert (r.n_steps, r.divergent, r.accept_sum, r.depth) == (0, 0, 0.0, 0)
    assert L.exo_nt_is_terminated(t) == 0
    for go_right in (1, 0):
        qe, pe, ge = np.zeros(3), np.zeros(3), np.zeros(3)
        L.exo_nt_get_endpoint(t, go_right, O.dptr(qe), O.dptr(pe), O.dptr(ge))
        assert list(qe) == c["q"] and list(pe) == c["p"] and list(ge) == c["grad"]
    L.exo_nt_free(t)


@pytest.mark.parametrize("name", ["depth0", "depth1", "divergent"])
def test_nif_build_and_merge(name):
    c = GOLD["native_tree"][name]
    L = O.lib()
    t = _nt_traj(c)
    a = [O.arr(c[k]) for k in ("all_q", "all_p", "all_logp", "all_grad", "inv_mass")]
    L.exo_nt_build_and_merge(t, *[O.dptr(x) for x in a], c["jlp0"], c["depth"], c["d"],
                             int(c["go_right"]), c["seed"])
    _, _, r = _nt_result(t, c["d"])
    e = c["expect"]
    if "n_steps" in e:
        assert r.n_steps == e["n_steps"]
    if "depth" in e:
        assert r.depth == e["depth"]
    if "divergent" in e:
        assert bool(r.divergent) == e["divergent"]
    if e.get("terminated"):
        assert L.exo_nt_is_terminated(t) == 1
    L.exo_nt_free(t)


def test_nif_full_tree():
    c = GOLD["native_tree"]["full_tree"]
    L = O.lib()
    a = {k: O.arr(c[k]) for k in ("q0", "p0", "grad0", "fwd_q", "fwd_p", "fwd_logp", "fwd_grad",
                                  "bwd_q", "bwd_p", "bwd_logp", "bwd_grad", "inv_mass")}
    qo, go = np.zeros(1), np.zeros(1)
    r = O.TreeResult()
    L.exo_nt_build_full_tree(O.dptr(a["q0"]), O.dptr(a["p0"]), O.dptr(a["grad0"]), c["logp0"],
                             O.dptr(a["fwd_q"]), O.dptr(a["fwd_p"]), O.dptr(a["fwd_logp"]),
                             O.dptr(a["fwd_grad"]), 7, O.dptr(a["bwd_q"]), O.dptr(a["bwd_p"]),
                             O.dptr(a["bwd_logp"]), O.dptr(a["bwd_grad"]), 7, O.dptr(a["inv_mass"]),
                             c["jlp0"], c["max_depth"], 1, c["seed"], O.dptr(qo), O.dptr(go),
                             C.byref(r))
    e = c["expect"]
    assert r.n_steps > e["n_steps_gt"] and r.accept_sum > e["accept_sum_gt"]
    assert e["depth_gt"] < r.depth <= e["depth_le"]
    assert r.n_steps <= 2 ** r.depth - 1 or r.n_steps <= 7


def test_nif_full_tree_all_divergent():
    c = GOLD["native_tree"]["full_tree_divergent"]
    L = O.lib()
    n = c["n"]
    fq, bq = np.full(n, c["fwd_q_value"]), np.full(n, c["bwd_q_value"])
    ps, gs, lps = np.full(n, c["p_value"]), np.full(n, c["grad_value"]), np.full(n, c["logp_value"])
    q0, p0, g0, im = O.arr(c["q0"]), O.arr(c["p0"]), O.arr(c["grad0"]), O.arr(c["inv_mass"])
    qo, go = np.zeros(1), np.zeros(1)
    r = O.TreeResult()
    L.exo_nt_build_full_tree(O.dptr(q0), O.dptr(p0), O.dptr(g0), c["logp0"], O.dptr(fq), O.dptr(ps),
                             O.dptr(lps), O.dptr(gs), n, O.dptr(bq), O.dptr(ps), O.dptr(lps),
                             O.dptr(gs), n, O.dptr(im), c["jlp0"], c["max_depth"], 1, c["seed"],
                             O.dptr(qo), O.dptr(go), C.byref(r))
    assert bool(r.divergent) and r.n_steps <= c["expect"]["n_steps_le"]


# ---- Tree.build smoke (test/nuts_test.exs:248-297) ----
def _std_tree(eps, max_depth, seed, mode=0):
    L = O.lib()
    m = O.std_normal(1)
    cfg = O.Cfg(mode, 1)
    q = np.array([0.0])
    lp, g = m.logp_grad(q, cfg)
    r = O.Rng()
    L.exo_rng_seed(C.byref(r), seed)
    p = np.array([L.exo_rng_normal(C.byref(r), mode)])
    im = np.array([1.0])
    jlp0 = lp - L.exo_kinetic_energy(O.dptr(p), O.dptr(im), 1, cfg)
    return m.tree_build(q, p, lp, g, eps, im, max_depth, r, jlp0, cfg)[2]


@pytest.mark.parametrize("mode", MODES)
def test_tree_smoke(mode):
    t = GOLD["tree_smoke"]
    r = _std_tree(t["single_depth"]["epsilon"], t["single_depth"]["max_depth"],
                  t["single_depth"]["seed"], mode)
    assert r.n_steps >= t["single_depth"]["expect_n_steps_ge"]
    r = _std_tree(t["extreme_step"]["epsilon"], t["extreme_step"]["max_depth"],
                  t["extreme_step"]["seed"], mode)
    assert bool(r.divergent) == t["extreme_step"]["expect_divergent"]
    r = _std_tree(t["uturn"]["epsilon"], t["uturn"]["max_depth"], t["uturn"]["seed"], mode)
    assert r.depth < t["uturn"]["expect_depth_lt"] and not r.divergent
