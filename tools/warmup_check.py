#!/usr/bin/env python3
"""Development aid: run the on-device warmup of one bench model and print the tuning next to the
CPU checker's (deterministic-math mode, same lanes).  python tools/warmup_check.py sv [lanes] [num_warmup]
"""
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
    name = sys.argv[1] if len(sys.argv) > 1 else "sv"
    spec, _ = bench.make_spec(name)
    comp = sampler.compile(spec)
    lanes = int(sys.argv[2]) if len(sys.argv) > 2 else comp.default_lanes
    nw = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
    tun = sampler.warmup(comp, spec.default_init, dict(num_warmup=nw, seed=42, lanes_per_chain=lanes))
    om = O.model_for(spec)
    st = O.warmup(om, spec.to_unconstrained(spec.default_init), num_warmup=nw, seed=42,
                  cfg=O.Cfg(1, lanes))
    im = np.array(st.inv_mass[:spec.d])
    print("%s lanes=%d warmup=%d" % (name, lanes, nw))
    print("  hip    eps %.17g  divergences %d" % (tun["epsilon"], tun["warmup_divergences"]))
    print("  oracle eps %.17g  divergences %d" % (st.step_size, st.divergences))
    print("  eps equal: %s   inv_mass equal: %s" % (tun["epsilon"] == st.step_size,
                                                   np.array_equal(tun["inv_mass"], im)))


if __name__ == "__main__":
    main()
