#!/usr/bin/env python3
"""GPU soak of the fused-chain hook (B2', lib/exmc/nuts/tree.ex:613-653: exmc_hip_leapfrog_chain_normal_host through
the Python mirror) with RANDOM settings: d 1..256, K 0..80, the batch size (1..400 chains, so both the scratch path and
the allocating path above 8 MB), the signed step size over eight decades, mu, sigma over twelve decades and below the
guard of normal.ex:18, the inverse mass, and in a third of the runs a few entries of q / p / inv_mass replaced by NaN /
+-inf / 1e300 / denormals -- against the checker's statement of the hook (deterministic log, 64-lane sums), every row of
every sampled chain bit for bit (NaN for NaN).

    gpurun -- 'python tools/fused_chain_soak_gpu.py <first seed> <last seed>'
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle as O  # noqa: E402
from exmc_amd import fused_chain  # noqa: E402

POISON = [np.nan, np.inf, -np.inf, 1e300, -1e300, 1e200, -1e-200, 0.0, -0.0, 5e-324]
CFG = O.Cfg(1, 64)


def one(seed):
    rng = np.random.default_rng(seed)
    d = int(rng.choice([1, 2, int(rng.integers(1, 257)), 63, 64, 65, 128, 256]))
    k = int(rng.choice([0, 1, 2, 32, int(rng.integers(0, 81))]))
    Cn = int(rng.choice([1, 1, 2, int(rng.integers(1, 40)), int(rng.integers(40, 401))]))
    eps = float(rng.choice([-1.0, 1.0]) * 10.0 ** rng.uniform(-6, 2))
    mu = float(rng.choice([0.0, rng.normal() * 10.0]))
    sigma = float(rng.choice([1.0, 10.0 ** rng.uniform(-6, 6), 1e-35, 0.0]))
    im = np.ascontiguousarray(10.0 ** rng.uniform(-1.5, 1.5, size=d))
    scale = float(rng.choice([0.1, 1.0, 30.0]))
    q = rng.normal(size=(Cn, d)) * scale * max(sigma, 1e-3) + mu
    p = rng.normal(size=(Cn, d)) / np.sqrt(im)
    poisoned = bool(rng.integers(3) == 0)
    if poisoned:
        for _ in range(int(rng.integers(1, 6))):
            arr = (q, p)[int(rng.integers(2))]
            arr[int(rng.integers(Cn)), int(rng.integers(d))] = POISON[int(rng.integers(len(POISON)))]
        if rng.integers(4) == 0:
            im[int(rng.integers(d))] = POISON[int(rng.integers(len(POISON)))]
    got = fused_chain.leapfrog_chain_normal(q, p, im, k, eps, mu, sigma)
    rows = 0
    for c in sorted(set([0, Cn - 1] + [int(x) for x in rng.integers(0, Cn, size=3)])):
        exp = O.leapfrog_chain_normal(q[c], p[c], im, k, eps, mu, sigma, CFG)
        for g, e, what in zip(got, exp, ("q", "p", "logp", "grad")):
            if not (g[c].shape == e.shape and np.array_equal(g[c], e, equal_nan=True)):
                raise SystemExit("seed %d chain %d: %s differs (d %d k %d C %d eps %g mu %g sigma %g poisoned %s)"
                                 % (seed, c, what, d, k, Cn, eps, mu, sigma, poisoned))
        rows += k
    return rows, poisoned, (3 * Cn * k * d + Cn * k + 2 * Cn * d + d) * 8 > (8 << 20)


def main():
    lo, hi = int(sys.argv[1]), int(sys.argv[2])
    rows = pois = big = 0
    for s in range(lo, hi):
        r, p, b = one(s)
        rows += r; pois += p; big += b
        if (s - lo + 1) % 500 == 0:
            print("seed %d: %d leapfrog rows compared so far" % (s, rows), flush=True)
    print("fused_chain_soak: seeds %d..%d, %d settings (%d poisoned, %d above the scratch size), %d leapfrog rows "
          "compared with the checker, none differing" % (lo, hi, hi - lo, pois, big, rows))


if __name__ == "__main__":
    main()
