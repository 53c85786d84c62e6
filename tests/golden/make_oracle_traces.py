#!/usr/bin/env python3
"""Generate tests/golden/oracle_traces.npz: seeded per-draw outputs of the CPU checker in
deterministic-math mode with fixed tuning (no libm on the path => the bits are platform
independent). Used (a) on CPU to catch any change of the checker / numeric contract and (b) on
the GPU as a committed golden fixture for the HIP path.  Run:  python tests/golden/make_oracle_traces.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle as O  # noqa: E402

CASES = {
    # name: (model factory, lanes, eps, n_chains, n_draws, max_depth, base_seed)
    "es_g1": ("eight_schools", 1, 0.40, 3, 50, 10, 42),
    "es_g16": ("eight_schools", 16, 0.40, 3, 50, 10, 42),
    "es_g8_deep": ("eight_schools", 8, 0.05, 2, 12, 7, 7),
    "es_g16_div": ("eight_schools", 16, 1.70, 3, 40, 10, 11),
    "simple_g1": ("simple", 1, 0.30, 2, 40, 10, 0),
    "sv_g64": ("sv", 64, 0.05, 2, 10, 6, 42),
    # round 6: step sizes at which these three models build real trees and MOVE. At the
    # round-1 values (logistic 0.30, radon 0.20, with inv_mass_for's 0.5..2) every transition was
    # rejected or diverged on its first leapfrog: the draws never left the start, so the traces could
    # not see a change of the per-observation arithmetic. The old cases stay as *_stuck / *_div0.
    "sv_g64_fine": ("sv", 64, 0.02, 2, 10, 6, 42),
    "logistic_g16": ("logistic", 16, 0.06, 2, 12, 6, 5),
    "logistic_g16_stuck": ("logistic", 16, 0.30, 2, 12, 6, 5),
    "radon_g64": ("radon", 64, 0.02, 2, 12, 6, 6),
    "radon_g64_div": ("radon", 64, 0.04, 2, 12, 6, 6),
    "radon_g64_div0": ("radon", 64, 0.20, 2, 12, 6, 6),
    "logistic_g4_mfma": ("logistic", 4, 0.06, 3, 12, 6, 5),
}


def sv_returns(seed=42, T=100):
    rng = np.random.default_rng(seed)
    s = np.cumsum(rng.normal(0, 0.15, T))
    return np.exp(s) * rng.standard_t(10.0, T)


def inv_mass_for(d):
    return 0.5 + 1.5 * (np.arange(d) % 7) / 6.0


def init_for(name, d):
    if name == "simple":
        return np.array([2.0, 0.0])
    if name == "sv":
        q = np.zeros(d)
        q[100], q[101] = np.log(0.1), np.log(10.0)
        return q
    return np.zeros(d)   # eight_schools, logistic, radon: zeros (log of unit scales = 0)


def spec_for(name):
    """The product's model descriptions double as the source of the synthetic data sets."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from exmc_amd import models
    return {"logistic": models.logistic, "radon": models.radon}[name]()


def model_for(name):
    if name == "eight_schools":
        return O.eight_schools()
    if name == "simple":
        return O.simple()
    if name in ("logistic", "radon"):
        spec = spec_for(name)
        return O.model_for(spec)
    return O.Model(O.SV, 102, sv_returns())


def run_case(name):
    mname, lanes, eps, nc, nd, md, seed = CASES[name]
    m = model_for(mname)
    im = inv_mass_for(m.d)
    out = {}
    for c in range(nc):
        t, _ = O.sample_tuned(m, eps, im, init_for(mname, m.d), num_samples=nd, max_tree_depth=md,
                              seed=seed + 7919 * c, cfg=O.Cfg(1, lanes))
        for k, v in t.items():
            out.setdefault(k, []).append(v)
    return {k: np.stack(v) for k, v in out.items()}


def main():
    blob = {}
    for name in CASES:
        for k, v in run_case(name).items():
            blob["%s/%s" % (name, k)] = v
    blob["sv_returns"] = sv_returns()
    np.savez_compressed(os.path.join(HERE, "oracle_traces.npz"), **blob)
    print("wrote oracle_traces.npz with", len(blob), "arrays")


if __name__ == "__main__":
    main()
