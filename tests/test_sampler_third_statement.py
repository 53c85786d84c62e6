"""The checker's sample/3 (oracle/exmc_oracle.c: exo_sample -- warmup phases, windows, dual averaging, Welford,
step-size search, the transition bookkeeping with its discarded tree draws) against a third statement of the same
control flow in plain Python (tests/py_sampler.py over tests/py_tree.py), written from the reference's text. Model
arithmetic, leapfrog and the random stream are shared (ctypes), the logic is not: whole chains must agree in every
bit -- the tuned step size and mass matrix, every draw, every per-draw statistic, the divergence count."""
import numpy as np
import pytest

import oracle as O
import py_sampler as PS


@pytest.mark.parametrize("name,nw,ns,seed,max_depth,init", [
    ("eight_schools", 150, 25, 42, 10, None),        # three Phase II windows fit: [50, 75) [75, 100)
    ("eight_schools", 260, 15, 7, 10, "given"),      # windows cross iteration 200: the depth cap of 8 ends inside one
    ("eight_schools", 40, 20, 3, 6, None),           # adapt_end <= init_buffer: Phase I only, DA.finalize
    ("simple", 120, 30, 0, 10, "given"),
    ("std_normal", 90, 30, 11, 5, None),
    ("simple", 0, 25, 5, 10, None),                  # no warmup at all: the searched step size, unit mass
])
def test_whole_chains_agree_with_the_third_statement(name, nw, ns, seed, max_depth, init):
    m = {"eight_schools": O.eight_schools, "simple": O.simple, "std_normal": lambda: O.std_normal(4)}[name]()
    q0 = None
    if init == "given":
        q0 = np.zeros(m.d)
        if name == "simple":
            q0[0] = 2.0
    t, st = O.sample(m, q0, num_warmup=nw, num_samples=ns, max_tree_depth=max_depth, seed=seed)
    p, ps = PS.sample(m, q0, num_warmup=nw, num_samples=ns, max_tree_depth=max_depth, seed=seed)
    assert st.step_size == ps["step_size"]
    assert np.array_equal(np.array(st.inv_mass[:m.d]), ps["inv_mass"])
    assert st.divergences == ps["divergences"]
    for k in ("draws", "logp", "tree_depth", "n_steps", "divergent", "accept_prob", "energy"):
        assert np.array_equal(t[k], p[k]), (name, k)
    assert np.unique(t["draws"], axis=0).shape[0] > ns // 3            # the chains move


@pytest.mark.parametrize("nw", [0, 30, 400])
def test_warm_start_agrees_with_the_third_statement(nw):
    """opts[:warm_start] (sampler.ex:167-197): no step-size search, the given mass, min(num_warmup, 50) iterations."""
    m = O.eight_schools()
    im = 0.5 + np.arange(10) / 10.0
    t, st = O.sample_warm(m, 0.3, im, np.zeros(10), num_warmup=nw, num_samples=20, seed=4)
    p, ps = PS.sample(m, np.zeros(10), num_warmup=nw, num_samples=20, seed=4, warm_start=(0.3, im))
    assert st.step_size == ps["step_size"] and np.array_equal(np.array(st.inv_mass[:10]), ps["inv_mass"])
    for k in ("draws", "tree_depth", "n_steps", "divergent", "accept_prob", "energy"):
        assert np.array_equal(t[k], p[k]), k


def test_window_schedule_literals():
    """sampler.ex:765-785 for num_warmup = 1000: [75, 100) [100, 150) [150, 250) [250, 450) [450, 950) (SURVEY
    appendix A; golden `window_schedule`)."""
    assert PS.windows(75, 950) == [(75, 100), (100, 150), (150, 250), (250, 450), (450, 950)]
    assert PS.windows(50, 100) == [(50, 75), (75, 100)]
    assert PS.windows(10, 10) == []


@pytest.mark.parametrize("name,spread,seed", [("eight_schools", 6.0, 1), ("eight_schools", 40.0, 2), ("simple", 40.0, 3),
                                              ("std_normal", 300.0, 4)])
def test_whole_chains_from_hostile_starts_agree_with_the_third_statement(name, spread, seed):
    """The same comparison started far from the mode (the step-size search halves dozens of times, first trees diverge
    at their first leaf or run into the transforms' clamps, Welford windows see a drifting chain): where NaN or an
    infinite energy enters the control flow, both statements must take the same branch."""
    m = {"eight_schools": O.eight_schools, "simple": O.simple, "std_normal": lambda: O.std_normal(4)}[name]()
    q0 = np.random.default_rng(seed).normal(size=m.d) * spread
    t, st = O.sample(m, q0, num_warmup=60, num_samples=20, max_tree_depth=8, seed=seed)
    p, ps = PS.sample(m, q0, num_warmup=60, num_samples=20, max_tree_depth=8, seed=seed)
    assert st.step_size == ps["step_size"]
    assert np.array_equal(np.array(st.inv_mass[:m.d]), ps["inv_mass"], equal_nan=True)
    assert st.divergences == ps["divergences"]
    for k in ("draws", "logp", "tree_depth", "n_steps", "divergent", "accept_prob", "energy"):
        assert np.array_equal(t[k], p[k], equal_nan=True), (name, k)
