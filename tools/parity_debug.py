#!/usr/bin/env python3
"""Development aid: first differing draw between the HIP path and the CPU checker for a bench
model.  python tools/parity_debug.py logistic 16 37 25"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import bench  # noqa: E402
import oracle as O  # noqa: E402
from exmc_amd import sampler  # noqa: E402


def main():
    name = sys.argv[1]
    lanes, n_chains, n_draws = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    nw = int(sys.argv[5]) if len(sys.argv) > 5 else 1000
    spec, _ = bench.make_spec(name)
    comp = sampler.compile(spec)
    om = O.model_for(spec)
    opts = dict(num_warmup=nw, num_samples=n_draws, seed=42, lanes_per_chain=lanes)
    tuning = sampler.warmup(comp, spec.default_init, opts)
    q0 = spec.to_unconstrained(spec.default_init)
    t, st = O.sample_chains(om, n_chains, init_q=q0, num_warmup=nw, num_samples=n_draws, seed=42,
                            n_threads=8, cfg=O.Cfg(1, lanes))
    print("eps equal", st.step_size == tuning["epsilon"], st.step_size, tuning["epsilon"])
    _, _, extra = sampler.sample_compiled_tuned(comp, tuning, spec.default_init, opts,
                                                num_chains=n_chains)
    raw = extra["raw"]
    for c in range(n_chains):
        for s in range(n_draws):
            bad = [k for k in ("tree_depth", "n_steps", "divergent", "draws", "logp", "accept_prob",
                               "energy") if not np.array_equal(t[k][c, s], raw[k][c, s])]
            if bad:
                print("chain %d draw %d differs in %s" % (c, s, bad))
                for k in ("tree_depth", "n_steps", "divergent", "logp", "accept_prob", "energy"):
                    print("   %-12s oracle %r   hip %r" % (k, t[k][c, s], raw[k][c, s]))
                d = np.flatnonzero(t["draws"][c, s] != raw["draws"][c, s])
                print("   draws differ at dims", d[:8], t["draws"][c, s][d[:4]], raw["draws"][c, s][d[:4]])
                if s > 0:
                    print("   previous draw equal:", np.array_equal(t["draws"][c, s - 1], raw["draws"][c, s - 1]))
                break
    print("done")


if __name__ == "__main__":
    main()
