"""The checker's tree (oracle/exmc_oracle.c: exo_tree_build) against a third, independent statement of
Tree.build written in plain Python from the reference's text (tests/py_tree.py): same leapfrog function and same
random stream on both sides, so every output must agree to the bit -- position, gradient, log-density, the integers,
the acceptance sum. Covers ordinary transitions, step sizes
that diverge or reject, depth caps, and a narrow target that U-turns early (nuts_test.exs:248-297's situations)."""
import ctypes as C

import numpy as np
import pytest

import oracle as O
import py_tree as PT


SEEN = {}


def _models():
    from exmc_amd import models
    return [("eight_schools", O.eight_schools(), 0.6), ("simple", O.simple(), 0.7), ("std_normal", O.std_normal(5), 1.0),
            ("radon", O.model_for(models.radon()), 0.15)]


@pytest.mark.parametrize("name,m,scale", _models(), ids=lambda x: x if isinstance(x, str) else "")
@pytest.mark.parametrize("eps,max_depth", [(0.1, 5), (0.4, 10), (1.9, 10), (0.02, 3), (25.0, 6)])
def test_transitions_agree_with_the_third_statement(name, m, scale, eps, max_depth):
    eps_in, max_depth_in = eps, max_depth
    if name == "radon":
        eps = eps * 0.05
        if max_depth > 6:
            max_depth = 6            # (a pure-Python tree of a 90-dimensional model: keep it small)
    L = O.lib()
    rng = np.random.default_rng(abs(hash((name, eps))) % 2 ** 31)
    seen = dict(div=0, turn_early=0, capped=0, moved=0)
    n = 6 if name == "radon" else 25
    for k in range(n):
        q = rng.normal(size=m.d) * scale
        im = 0.5 + rng.uniform(size=m.d) * 1.5
        lp, g = m.logp_grad(q)
        r0 = O.Rng()
        L.exo_rng_seed(C.byref(r0), 1000 + k)
        p = np.array([L.exo_rng_normal(C.byref(r0), 0) for _ in range(m.d)]) / np.sqrt(im)
        jlp0 = lp - L.exo_kinetic_energy(O.dptr(p), O.dptr(im), m.d, O.Cfg(0, 1))
        ra, rb = O.Rng(), O.Rng()
        C.memmove(C.byref(ra), C.byref(r0), C.sizeof(O.Rng))
        C.memmove(C.byref(rb), C.byref(r0), C.sizeof(O.Rng))
        qo, go, res = m.tree_build(q, p, lp, g, eps, im, max_depth, ra, jlp0, O.Cfg(0, 1))
        py = PT.build(m, q, p, lp, g, eps, im, max_depth, rb, jlp0)
        assert (res.depth, res.n_steps, bool(res.divergent)) == (py["depth"], py["n_steps"], py["divergent"]), (name, k)
        assert res.accept_sum == py["accept_sum"] and res.logp == py["logp"], (name, k)
        assert np.array_equal(qo, py["q"]) and np.array_equal(go, py["grad"]), (name, k)
        # (the checker takes its generator by value -- the sampler discards the tree's draws, sampler.ex:897 -- so
        # the streams are compared through what they decided: directions, both kinds of proposal selection)
        seen["div"] += bool(res.divergent)
        seen["capped"] += res.depth == max_depth
        seen["turn_early"] += (not res.divergent) and res.depth < max_depth
        seen["moved"] += not np.array_equal(qo, q)
    SEEN.setdefault((eps_in, max_depth_in), []).append(seen)
    assert seen["moved"] > 0 or eps_in >= 1.9


def test_the_cases_cover_every_way_a_tree_ends():
    """(runs after the parametrised test) over the four models: divergent trees at the largest step size, trees cut
    by the depth cap at the smallest, trees ended by a U-turn in between."""
    if len(SEEN) < 5:
        pytest.skip("the parametrised cases ran in another process")
    tot = lambda key, k: sum(s[k] for s in SEEN.get(key, []))   # noqa: E731
    assert tot((25.0, 6), "div") > 20
    assert tot((0.02, 3), "capped") > 20
    assert tot((0.4, 10), "turn_early") > 20 and tot((0.1, 5), "moved") > 20
