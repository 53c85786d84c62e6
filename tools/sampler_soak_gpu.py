#!/usr/bin/env python3
"""GPU soak of Sampler.sample/3 for every hand-written kind: whole runs (initial position, step-size search, the
three warmup phases with their windows, the draws) with RANDOM options -- num_warmup from a list built around the
schedule's edges (0, 1, the init buffer 75, a first window, 150, the depth cap's end 200, ...), target_accept
0.5..0.95, depth caps 2..10, seeds, the random or a given start, every lane layout of the kind, and in a third of
the runs the mode of the run: warm_start from a previous run's tuning, dense_mass, or n independently adapting chains
(vectorized: false) -- against the checker's sample / sample_warm / warmup_dense, every output bit for bit. The test
suite runs fixed options (tests/test_gpu_parity.py, test_gpu_independent.py); this runs as many as asked for.

    gpurun -- 'python tools/sampler_soak_gpu.py <first seed> <last seed>'
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle as O  # noqa: E402
import test_golden_traces as TG  # noqa: E402
from exmc_amd import _lib, models, sampler  # noqa: E402

# kind, factory, lane layouts, largest num_warmup asked of it (the checker runs on one CPU thread), dense layouts
KINDS = [("simple", models.simple, [1], 420, [1]),
         ("eight_schools", models.eight_schools, [1, 2, 4, 8, 16], 420, [1, 16]),
         ("sv", lambda: models.sv(TG.GOLD["sv_returns"]), [32, 64], 120, []),
         ("logistic", models.logistic, [4, 8, 16], 230, []),
         ("radon", models.radon, [32, 64], 160, [])]
NW = [0, 1, 2, 9, 24, 25, 26, 49, 50, 51, 74, 75, 76, 99, 100, 101, 120, 149, 150, 151, 199, 200, 201, 230, 300, 420]
KEYS = ("tree_depth", "n_steps", "divergent", "draws", "logp", "accept_prob", "energy")


def _same(a, b):
    return np.array_equal(np.asarray(a), np.asarray(b), equal_nan=True)


def one(hip, seed, cache):
    rng = np.random.default_rng(seed)
    name, factory, lane_list, nw_max, dense_lanes = KINDS[int(rng.integers(len(KINDS)))]
    if name not in cache:
        spec = factory()
        cache[name] = (spec, sampler.compile(spec), O.model_for(spec))
    spec, comp, om = cache[name]
    lanes = int(lane_list[int(rng.integers(len(lane_list)))])
    nw = int(rng.choice([n for n in NW if n <= nw_max]))
    ns = int(rng.integers(1, 12))
    depth = int(rng.integers(2, 11 if spec.d <= 21 else 8))
    ta = float(rng.choice([0.5, 0.65, 0.8, 0.9, 0.95]))
    base = int(rng.integers(0, 2 ** 31))
    given = bool(rng.integers(2))
    init = spec.default_init if given else None
    q0 = spec.to_unconstrained(spec.default_init) if given else None
    cfg = O.Cfg(1, lanes)
    mode = str(rng.choice(["plain", "plain", "plain", "plain", "warm", "dense", "independent"]))
    if mode == "dense" and not dense_lanes:
        mode = "plain"
    opts = dict(num_warmup=nw, num_samples=ns, seed=base, lanes_per_chain=lanes, max_tree_depth=depth, target_accept=ta)
    tag = "%s G=%d nw=%d ns=%d depth<=%d accept=%g seed=%d init=%s %s" % (name, lanes, nw, ns, depth, ta, base,
                                                                        "given" if given else "random", mode)
    if mode == "plain":
        _, st = sampler.sample_compiled(comp, init, opts)
        t, ost = O.sample(om, init_q=q0, num_warmup=nw, num_samples=ns, max_tree_depth=depth, target_accept=ta, seed=base,
                          cfg=cfg)
        ok = st["step_size"] == ost.step_size and _same(st["inv_mass_diag"], ost.inv_mass[:spec.d]) and \
            st["divergences"] == ost.divergences and all(_same(st["raw"][k][0], t[k]) for k in KEYS)
        info = "eps %.4g, %d leapfrogs" % (ost.step_size, int(t["n_steps"].sum()))
    elif mode == "warm":
        pe = float(10.0 ** rng.uniform(-2.5, 0.0))
        pim = np.ascontiguousarray(10.0 ** rng.uniform(-0.7, 0.7, size=spec.d))
        _, st = sampler.sample_compiled(comp, init, dict(opts, warm_start=dict(step_size=pe, inv_mass_diag=pim)))
        t, ost = O.sample_warm(om, pe, pim, q0, num_warmup=nw, num_samples=ns, max_tree_depth=depth, target_accept=ta,
                               seed=base, cfg=cfg)
        ok = st["step_size"] == ost.step_size and _same(st["inv_mass_diag"], ost.inv_mass[:spec.d]) and \
            all(_same(st["raw"][k][0], t[k]) for k in KEYS)
        info = "eps %.4g -> %.4g" % (pe, ost.step_size)
    elif mode == "dense":
        lanes = int(dense_lanes[int(rng.integers(len(dense_lanes)))])
        cfg = O.Cfg(1, lanes)
        tuning = sampler.warmup(comp, init, dict(opts, lanes_per_chain=lanes, dense_mass=True))
        ost, cov, chol = O.warmup_dense(om, q0, num_warmup=nw, max_tree_depth=depth, target_accept=ta, seed=base, cfg=cfg)
        ok = tuning["epsilon"] == ost.step_size and _same(tuning["cov"], cov) and _same(tuning["chol_cov"], chol)
        info = "dense G=%d eps %.4g" % (lanes, ost.step_size)
        comp.check(comp.L.exmc_hip_model_clear_dense_mass(comp.h))
    else:
        nc = int(rng.integers(2, 7 if spec.d <= 21 else 4))
        lanes = comp.default_lanes                      # the one-launch kernel runs in the kind's default layout
        cfg = O.Cfg(1, lanes)
        opts = dict(opts, lanes_per_chain=lanes)
        _, stats = sampler.sample_chains_independent_compiled(
            comp, nc, dict(opts, vectorized=False, init_values=init or {}))
        raw = stats[0]["extra"]["raw"]
        ok = True
        for c in range(nc):
            t, ost = O.sample(om, init_q=q0, num_warmup=nw, num_samples=ns, max_tree_depth=depth, target_accept=ta,
                              seed=base + 7919 * c, cfg=cfg)
            ok = ok and stats[c]["step_size"] == ost.step_size and _same(stats[c]["inv_mass_diag"], ost.inv_mass[:spec.d]) \
                and stats[c]["divergences"] == ost.divergences and all(_same(raw[k][c], t[k]) for k in KEYS)
        info = "%d chains" % nc
    if not ok:
        raise AssertionError("differs: " + tag)
    return tag + ": " + info


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
