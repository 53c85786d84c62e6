#!/usr/bin/env python3
"""Ablation probe for the NUTS kernel: time a launch in the lockstep regime (tiny step, fixed
max depth => every chain builds the same tree shape, no inter-group divergence) next to the
adapted regime, per leapfrog. Run on the GPU box:  python tools/kernel_probe.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from exmc_amd import models, sampler  # noqa: E402


def run(comp, spec, eps, max_depth, n_chains, n_draws, lanes):
    tuning = dict(epsilon=eps, inv_mass=np.ones(spec.d))
    opts = dict(num_samples=n_draws, max_tree_depth=max_depth, seed=42, lanes_per_chain=lanes)
    _, _, ex = sampler.sample_compiled_tuned(comp, tuning, spec.default_init, opts, num_chains=n_chains)
    lf = ex["total_leapfrogs"]
    ms = ex["kernel_ms"]
    return lf, ms


def main():
    spec = models.eight_schools()
    comp = sampler.compile(spec)
    lane_list = [int(x) for x in os.environ.get("PROBE_LANES", "16,8,4,1").split(",")]
    for lanes in lane_list:
        for (eps, md, nd, label) in ((1e-4, 3, 300, "lockstep d3"), (1e-4, 5, 100, "lockstep d5"),
                                     (0.45, 10, 300, "adapted")):
            run(comp, spec, eps, md, 4096, 20, lanes)
            lf, ms = run(comp, spec, eps, md, 4096, nd, lanes)
            iters = lf / 4096.0
            print("G=%2d %-12s leapfrogs=%9d kernel=%8.3f ms  %.3f us per chain-leapfrog-iteration  %.2e lf/s"
                  % (lanes, label, lf, ms, ms * 1e3 / iters, lf / ms * 1e3), flush=True)


if __name__ == "__main__":
    main()
