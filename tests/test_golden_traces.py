"""Committed per-draw golden traces (tests/golden/oracle_traces.npz).

CPU: the checker still reproduces them bit for bit (catches drift of the checker or of the numeric
contract). GPU (-m gpu): the HIP path reproduces the same committed bits through the C ABI."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import make_oracle_traces as G  # noqa: E402

GOLD = np.load(os.path.join(HERE, "golden", "oracle_traces.npz"))
KEYS = ("draws", "logp", "tree_depth", "n_steps", "divergent", "accept_prob", "energy")


@pytest.mark.parametrize("name", sorted(G.CASES))
def test_oracle_reproduces_committed_traces(name):
    got = G.run_case(name)
    for k in KEYS:
        assert np.array_equal(got[k], GOLD["%s/%s" % (name, k)]), (name, k)


def test_golden_traces_cover_edge_cases():
    assert GOLD["es_g16_div/divergent"].sum() > 0            # divergent transitions
    assert GOLD["es_g8_deep/tree_depth"].max() == 7          # depth cap reached
    assert GOLD["es_g8_deep/n_steps"].max() == 127
    assert GOLD["sv_g64/draws"].shape == (2, 10, 102)
    assert GOLD["logistic_g16/draws"].shape == (2, 12, 21)
    assert GOLD["radon_g64/draws"].shape == (2, 12, 90)
    assert np.array_equal(GOLD["sv_returns"], G.sv_returns())


@pytest.mark.parametrize("name", ["radon_g64", "logistic_g16", "logistic_g4_mfma", "sv_g64_fine", "es_g16"])
def test_golden_traces_pin_real_trees(name):
    """a case whose transitions all die on the first leapfrog pins nothing of the
    per-observation arithmetic. These cases must build real trees, accept, and move."""
    n_steps, div = GOLD[name + "/n_steps"], GOLD[name + "/divergent"]
    draws, logp = GOLD[name + "/draws"], GOLD[name + "/logp"]
    assert n_steps.max() > 1 and n_steps.mean() > 3
    assert div.sum() < div.size // 4
    moved = (np.diff(draws, axis=1) != 0).any(axis=2)
    assert moved.mean() > 0.7                      # most transitions leave their start
    assert np.unique(logp).size > logp.size // 2   # and the log-density is not a constant
    assert GOLD[name + "/accept_prob"].mean() > 0.3


def test_golden_traces_keep_the_degenerate_cases_too():
    """every transition divergent after real steps (radon_g64_div), on the first leapfrog
    (radon_g64_div0), or rejected outright (logistic_g16_stuck): the revert paths of tree.ex:1042-1048."""
    assert GOLD["radon_g64_div/divergent"].all() and GOLD["radon_g64_div/n_steps"].min() >= 3
    assert GOLD["radon_g64_div0/divergent"].all() and GOLD["radon_g64_div0/n_steps"].max() == 1
    assert not GOLD["logistic_g16_stuck/divergent"].any()
    assert (GOLD["logistic_g16_stuck/draws"] == 0.0).all()


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(G.CASES))
def test_hip_reproduces_committed_traces(name, hip):
    from exmc_amd import models, sampler
    mname, lanes, eps, nc, nd, md, seed = G.CASES[name]
    spec = {"eight_schools": models.eight_schools, "simple": models.simple,
            "sv": lambda: models.sv(GOLD["sv_returns"]), "logistic": models.logistic,
            "radon": models.radon}[mname]()
    comp = sampler.compile(spec)
    tuning = dict(epsilon=eps, inv_mass=G.inv_mass_for(spec.d))
    opts = dict(num_samples=nd, max_tree_depth=md, seed=seed, lanes_per_chain=lanes)
    _, _, extra = sampler.sample_compiled_tuned(comp, tuning, spec.default_init, opts, num_chains=nc)
    assert np.array_equal(spec.to_unconstrained(spec.default_init), G.init_for(mname, spec.d))
    for k in KEYS:
        assert np.array_equal(extra["raw"][k], GOLD["%s/%s" % (name, k)]), (name, k)
