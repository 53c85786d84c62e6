#!/usr/bin/env python3
"""GPU soak of the NUTS transition for every hand-written kind: batches of chains under a GIVEN tuning
(sample_compiled_tuned/4 = exmc_hip_sample_chains_host) with RANDOM settings -- step size over two decades
around the kind's usual one, inverse mass entries over a decade and a half, depth caps 1..9, a start 0.1..3 units from
the default position, every lane layout of the kind -- against the checker's sample_tuned chain by chain, every
per-draw output bit for bit. The test suite runs a handful of fixed settings per kind (tests/test_gpu_parity.py);
this runs as many as asked for.

    gpurun -- 'python tools/tree_soak_gpu.py <first seed> <last seed>'
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle as O  # noqa: E402
import test_golden_traces as TG  # noqa: E402
from exmc_amd import _lib, models, sampler  # noqa: E402

KINDS = [("simple", models.simple, [1], 0.5),
         ("eight_schools", models.eight_schools, [1, 2, 4, 8, 16], 0.45),
         ("sv", lambda: models.sv(TG.GOLD["sv_returns"]), [32, 64], 0.03),
         ("logistic", models.logistic, [4, 8, 16], 0.08),
         ("radon", models.radon, [32, 64], 0.03)]


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def one(hip, seed, cache):
    rng = np.random.default_rng(seed)
    name, factory, lane_list, eps0 = KINDS[int(rng.integers(len(KINDS)))]
    if name not in cache:
        spec = factory()
        cache[name] = (spec, sampler.compile(spec), O.model_for(spec))
    spec, comp, om = cache[name]
    lanes = int(lane_list[int(rng.integers(len(lane_list)))])
    eps = float(eps0 * 10.0 ** rng.uniform(-1.5, 0.6))
    max_depth = int(rng.integers(1, 10 if spec.d <= 21 else 8))
    spread = float(rng.choice([0.1, 0.3, 1.0, 3.0]))
    C_ = int(rng.integers(3, 40 if spec.d <= 21 else 9))
    n_draws = int(rng.integers(2, 9))
    d = spec.d
    cfg = O.Cfg(1, lanes)
    # the batch starts at ONE position (sample_compiled_tuned's contract: chain i differs by its seed), given as
    # constrained init values by name, as the API takes them
    q_far = spec.to_unconstrained(spec.default_init) + rng.normal(size=d) * spread
    init = {n: float(np.exp(q_far[i])) if spec.transforms.get(n) == "log" else float(q_far[i])
            for i, n in enumerate(spec.var_names)}
    q0 = spec.to_unconstrained(init)
    im = np.ascontiguousarray(10.0 ** rng.uniform(-0.8, 0.8, size=d))
    base = int(rng.integers(0, 2 ** 31))
    opts = dict(num_samples=n_draws, seed=base, lanes_per_chain=lanes, max_tree_depth=max_depth)
    _, _, extra = sampler.sample_compiled_tuned(comp, dict(epsilon=eps, inv_mass=im), init, opts, num_chains=C_)
    t = extra["raw"]
    cfg_s = "%s G=%d eps=%.17g depth<=%d spread=%g chains=%d draws=%d seed=%d" % (name, lanes, eps, max_depth, spread, C_,
                                                                                 n_draws, base)
    for c in range(C_):
        o, _ = O.sample_tuned(om, eps, im, q0, num_samples=n_draws, max_tree_depth=max_depth, seed=base + 7919 * c, cfg=cfg)
        for k in ("tree_depth", "n_steps", "divergent", "draws", "logp", "accept_prob", "energy"):
            if not np.array_equal(o[k], t[k][c], equal_nan=True):
                raise AssertionError("%s of chain %d differs [%s]: checker %r gpu %r" % (k, c, cfg_s, o[k], t[k][c]))
    return "%s G=%d eps=%.3g depth<=%d spread=%g chains=%d draws=%d: depth %d..%d, %d divergent of %d" % (
        name, lanes, eps, max_depth, spread, C_, n_draws, t["tree_depth"].min(), t["tree_depth"].max(),
        int(t["divergent"].sum()), C_ * n_draws)


def main():
    lo, hi = int(sys.argv[1]), int(sys.argv[2])
    hip = _lib.load()
    cache, bad = {}, []
    for seed in range(lo, hi):
        try:
            print(seed, "ok", one(hip, seed, cache), flush=True)
        except Exception as e:   # noqa: BLE001
            print(seed, "FAIL", repr(e)[:600], flush=True)
            bad.append(seed)
    print("failed seeds:", bad)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
